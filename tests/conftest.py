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
