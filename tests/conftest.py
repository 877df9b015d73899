import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
GOLD = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    """A HIP device is visible (asked of the HIP runtime itself: importing torch for this costs a minute on a fresh box)."""
    import ctypes
    import os
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def gpu_count():
    """HIP devices visible (0 without a GPU); asked of the HIP runtime itself, not of torch"""
    import ctypes
    import os
    if not os.path.exists("/dev/kfd"):
        return 0
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def second_device():
    """the ordinal multi-device tests put their second block of chains on: device 1 where the box has one (the driver's 8-GPU node),
    else device 0 again (the code path is then exercised with both blocks on one GPU)"""
    return 1 if gpu_count() >= 2 else 0


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def data_2016():
    from us_potus_model_amd import dataprep
    return dataprep.load_npz(GOLD / "data_2016.npz")["data"]


@pytest.fixture(scope="session")
def cases(data_2016):
    from us_potus_model_amd import dataprep, synthetic
    return {
        "2016": (data_2016, "full"),
        "2012": (dataprep.load_npz(GOLD / "data_2012.npz")["data"], "no_mode_adjustment"),
        "2008": (dataprep.load_npz(GOLD / "data_2008.npz")["data"], "no_mode_adjustment"),
        "small_full": (synthetic.small("full"), "full"),
        "small_nomode": (synthetic.small("no_mode_adjustment"), "no_mode_adjustment"),
    }


def readme_golden(year):
    """tests/golden/readme_<year>.csv (scripts/make_readme_golden.py): the published figures (`# key = number` lines: Brier scores,
    states called correctly, RMSE over the 50 states) and the 52 table rows (with the certified result `actual` per state)."""
    import csv
    import re
    lines = open(GOLD / f"readme_{year}.csv").read().splitlines()
    pub = {m.group(1): float(m.group(2)) for m in (re.match(r"^# (\w+) = ([0-9.eE+-]+)$", ln) for ln in lines) if m}
    rows = list(csv.DictReader(ln for ln in lines if not ln.startswith("#")))
    return pub, rows


def rmse_ex_dc(states, mean_by_state, rows):
    """README.Rmd:392-401: sqrt(mean((posterior mean - actual)^2)) over the 50 states, DC left out."""
    import numpy as np
    act = {r["state"]: float(r["actual"]) for r in rows if r["state"] != "--"}
    d = [mean_by_state[i] - act[s] for i, s in enumerate(states) if s != "DC"]
    return float(np.sqrt(np.mean(np.square(d))))


def election_day_draws(h, n_saved):
    """mu_b[:, T] and predicted_score[T, :] of a handle's saved draws as [chains, draws, S] each (columns of the CmdStan output row, potus_write_array):
    mu_b is S x T column-major (stan:73), predicted_score T x S column-major (stan:135)."""
    import numpy as np
    S, T = int(h.data["S"]), int(h.data["T"])
    a_mu = h.layout["mu_b"][0]
    mu = h.write_array(a_mu + S * (T - 1), a_mu + S * T, n_saved).transpose(1, 0, 2)
    a_ps = h.layout["predicted_score"][0]
    ps = h.write_array(a_ps + (T - 1), a_ps + (T - 1) + T * (S - 1) + 1, n_saved)[:, :, ::T].transpose(1, 0, 2)
    return np.ascontiguousarray(mu), np.ascontiguousarray(ps)


def assert_posterior_within_mcse(mu_b_T, ps_T, golden, nsig=5.0, rhat_max=1.05):
    """Statistical parity with a committed run of the CPU oracle (tests/golden/posterior_*.npz, scripts/make_golden.py posterior): the pooled means of
    mu_b[:, T] and predicted_score[T, :] within `nsig` combined Monte-Carlo standard errors, the 2.5 % / 97.5 % quantiles within 6 MCSE + 0.004
    (the bar of test_posterior_2016_against_golden_and_readme since round 1).  Arrays are [chains, draws, S].  Returns the worst z-score."""
    import numpy as np
    from us_potus_model_amd import diagnostics as dg
    worst = 0.0
    for name, x in (("mu_b_T", mu_b_T), ("predicted_score_T", ps_T)):
        sm = dg.summarise(x)
        se = np.hypot(sm["mcse"], golden[f"{name}__mcse"])
        z = np.abs(sm["mean"] - golden[f"{name}__mean"]) / se
        assert z.max() < nsig, (name, float(z.max()))
        assert sm["rhat"].max() < rhat_max, (name, float(sm["rhat"].max()))
        pooled = x.reshape(-1, x.shape[-1])
        for q, key in ((0.025, "q025"), (0.975, "q975")):
            d = np.abs(np.quantile(pooled, q, axis=0) - golden[f"{name}__{key}"]).max()
            assert d < 6 * se.max() + 0.004, (name, key, float(d))
        worst = max(worst, float(z.max()))
    return worst


def build_call_wrapper(tmp_dir):
    """R/src/potus_call.c (the .Call() entry points of the R shim) compiled against the stub R headers under tests/r_stub/ -- R is not installed here -- and
    linked to the product library: returns the ctypes handle of the result, with the harness functions of the stub typed."""
    import ctypes as C
    import subprocess
    from pathlib import Path
    out = Path(tmp_dir) / "libpotus_call_stub.so"
    pkg = ROOT / "us_potus_model_amd"
    cmd = ["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I", str(ROOT / "tests" / "r_stub"), "-I", str(ROOT / "include"),
           str(ROOT / "R" / "src" / "potus_call.c"), str(ROOT / "tests" / "r_stub" / "r_stub.c"), "-L", str(pkg), "-lpotus_hmc", f"-Wl,-rpath,{pkg}", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    W = C.CDLL(str(out))
    P = C.c_void_p
    W.stub_int_vector.argtypes, W.stub_int_vector.restype = [C.POINTER(C.c_int), C.c_int], P
    W.stub_call0.argtypes, W.stub_call0.restype = [P], P
    W.stub_call1.argtypes, W.stub_call1.restype = [P, P], P
    W.stub_call3.argtypes, W.stub_call3.restype = [P, P, P, P], P
    W.stub_last_error.restype = C.c_char_p
    W.stub_string.argtypes, W.stub_string.restype = [P], C.c_char_p
    W.stub_live_bytes.restype = C.c_longlong
    W.REAL.argtypes, W.REAL.restype = [P], C.POINTER(C.c_double)
    W.XLENGTH.argtypes, W.XLENGTH.restype = [P], C.c_longlong
    return W


def call_wrapper_int(W, values):
    import ctypes as C
    a = (C.c_int * len(values))(*values)
    return W.stub_int_vector(a, len(values))
