"""Dense-metric path (potus_dense.hpp) on the GPU box: rates of its pieces and of the sampler.

  python scripts/micro/dense_probe.py pieces             matrix pass (1-3 right-hand sides), covariance, Cholesky, solve
  python scripts/micro/dense_probe.py sampler C N [cus]  2016 posterior, C chains, N warm-up iterations, dense metric
  python scripts/micro/dense_probe.py stress C N         the same on the configs[4] shape (D = 41 610)
  python scripts/micro/dense_probe.py active             matrix pass with 1..16 of 16 resident chains active, by tile split
  python scripts/micro/dense_probe.py pooled [D]         the POOLED pass (potus_opts.pooled_metric: one full matrix, all right-hand sides on the matrix cores)
                                                         for 1 .. 64 chains x 2 right-hand sides, by row split

Matrices of `pieces` are generated on the device; a matrix pass loads the upper-triangle tiles, about 4 D^2 bytes per chain.
"""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, _abi, dataprep, sampler, synthetic  # noqa: E402

L = sampler.load_library()
DP = C.POINTER(C.c_double)
L.potus_dense_matvec_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, DP, C.POINTER(C.c_longlong)]
L.potus_dense_factor_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, DP]


def err():
    buf = C.create_string_buffer(512)
    L.potus_last_error(buf, 512)
    return buf.value.decode()


ACTIVE_CASES = ("1", "2", "3", "4", "6", "8", "12", "16")      # "n": the first n chains of the handle take part; "a,b,c": those chains
SPLIT_CASES = (0, 1, 2, 3, 4, 6, 8)


def pooled_sweep(D=41610):
    """k_dn_pool_mm + k_dn_pool_finish on a generated matrix: ms per pass, TB/s of the 8 D^2 bytes it streams, TFLOP/s of its 2 D^2 R flops.
    POTUS_PROBE_F32=1: the same with the matrix kept rounded to fp32 (metric_storage = f32: 4 D^2 bytes per pass)."""
    L.potus_dense_pool_matvec_probe.argtypes = L.potus_dense_matvec_probe.argtypes
    nrhs = 2
    LD = (D + 7) // 8 * 8
    for chains in (1, 4, 8, 16, 24, 32, 64):
        x = np.random.default_rng(1).standard_normal((chains, nrhs, D))
        y, ms, nb = np.zeros((chains, nrhs, D)), C.c_double(), C.c_longlong()
        line = []
        for s in (0, 1, 2, 3, 4, 6, 8):
            if s:
                os.environ["POTUS_POOL_SPLIT"] = str(s)
            else:
                os.environ.pop("POTUS_POOL_SPLIT", None)
            rc = L.potus_dense_pool_matvec_probe(0, chains, D, nrhs, None, x.ctypes.data, y.ctypes.data, None, 5, C.byref(ms), C.byref(nb))
            if rc:
                line.append(f"{s}: error {err()}")
                continue
            tbs, tf = nb.value / 1e12 / (ms.value * 1e-3), 2.0 * D * D * chains * nrhs / 1e12 / (ms.value * 1e-3)
            line.append(f"{'auto' if not s else s}: {ms.value:.3f} ms {tbs:.2f} TB/s {tf:.1f} TF")
        print(f"pooled pass D={D} chains={chains:2d} x {nrhs} rhs ({nb.value / 1e9:.2f} GB per call) by row split  " + " | ".join(line), flush=True)
    os.environ.pop("POTUS_POOL_SPLIT", None)


def active_sweep():
    """16 chains of the configs[4] shape resident, a of them active, tile split s: time of one pass."""
    chains, D, nrhs = 16, 41610, 2
    x = np.random.default_rng(1).standard_normal((chains, nrhs, D))
    y, ms, nb = np.zeros((chains, nrhs, D)), C.c_double(), C.c_longlong()
    for a in ACTIVE_CASES:
        os.environ["POTUS_PROBE_ACTIVE"] = a
        n_a = len(a.split(",")) if "," in a else int(a)
        line = []
        for s in SPLIT_CASES:
            if s:
                os.environ["POTUS_DENSE_SPLIT"] = str(s)
            else:
                os.environ.pop("POTUS_DENSE_SPLIT", None)
            rc = L.potus_dense_matvec_probe(0, chains, D, nrhs, None, x.ctypes.data, y.ctypes.data, None, 3, C.byref(ms), C.byref(nb))
            line.append(f"{'auto' if not s else s}: {n_a * nb.value / 1e9 / (ms.value * 1e-3) / 8000:.2f}" if rc == 0 else f"{s}: error {err()}")
        print(f"D={D} resident=16 active={a:>8s} frac of 8 TB/s by split  " + "  ".join(line), flush=True)
    os.environ.pop("POTUS_PROBE_ACTIVE", None); os.environ.pop("POTUS_DENSE_SPLIT", None)


def pieces():
    f32 = bool(int(os.environ.get("POTUS_PROBE_F32", "0")))     # the pass over fp32 storage (generated matrices)
    cases = ((1, 15098, 1), (1, 15098, 2), (8, 15098, 2), (8, 15098, 3), (1, 41610, 2), (4, 41610, 2))
    if os.environ.get("POTUS_PROBE_CASES") == "short":
        cases = ((8, 15098, 2), (4, 41610, 2), (16, 41610, 2))
    for chains, D, nrhs in cases:
        x = np.random.default_rng(1).standard_normal((chains, nrhs, D))
        y, ms, nb = np.zeros((chains, nrhs, D)), C.c_double(), C.c_longlong()
        rc = L.potus_dense_matvec_probe(0, chains, D, nrhs, None, x.ctypes.data, y.ctypes.data, None, 5, C.byref(ms), C.byref(nb))
        if rc:
            print(f"chains={chains} D={D}: error {rc}: {err()}")
            continue
        ok = True                                        # spot check of a few rows against the generator's formula
        for c in (0, chains - 1):
            for i in (0, D // 3, D - 1):
                j = np.arange(D)
                row = np.exp(-np.abs(i - j) / 50.0) * (1.0 + 0.1 * c) + (j == i)
                if f32:
                    row = np.where(j == i, row, row.astype(np.float32).astype(np.float64))      # off-diagonal elements are stored rounded
                ok &= abs(row @ x[c, nrhs - 1] - y[c, nrhs - 1, i]) <= 1e-10 * max(1.0, abs(y[c, nrhs - 1, i]))
        gb = chains * nb.value / 1e9                     # bytes the pass loads: the upper-triangle tiles (about 4 D^2 per chain)
        print(f"matrix pass{' (fp32 storage)' if f32 else ''} chains={chains} D={D} nrhs={nrhs}: {ms.value:.3f} ms, {gb:.2f} GB loaded -> {gb / (ms.value * 1e-3):.0f} GB/s = "
              f"{gb / (ms.value * 1e-3) / 8000:.2f} of 8 TB/s ({chains * D * D * 8 / 1e9 / (ms.value * 1e-3):.0f} GB/s in full-matrix terms); "
              f"rows check {'ok' if ok else 'MISMATCH'}", flush=True)
    if os.environ.get("POTUS_PROBE_CASES") == "short":
        return
    for chains, D, n in ((1, 15098, 100), (4, 15098, 500), (1, 41610, 100)):
        rng = np.random.default_rng(2)
        draws = rng.standard_normal((chains, n, D))
        u = rng.standard_normal((chains, D))
        p, ms = np.zeros((chains, D)), (C.c_double * 3)()
        t = time.time()
        rc = L.potus_dense_factor_probe(0, chains, D, n, draws.ctypes.data, u.ctypes.data, None, None, p.ctypes.data, ms)
        if rc:
            print(f"factor chains={chains} D={D}: error {rc}: {err()}")
            continue
        fl_cov, fl_chol = 2.0 * chains * D * D / 2 * n, chains * D ** 3 / 3.0
        print(f"window end chains={chains} D={D} n={n}: covariance {ms[0]:.1f} ms ({fl_cov / ms[0] / 1e9:.1f} TFLOP/s), Cholesky {ms[1]:.1f} ms "
              f"({fl_chol / ms[1] / 1e9:.1f} TFLOP/s), solve {ms[2]:.2f} ms; wall {time.time() - t:.1f} s", flush=True)


def run_sampler(data, variant, chains, iters, cus, storage=0):
    h = Handle(data, variant, chains=chains, num_warmup=iters, num_samples=0, seed=1843, metric=_abi.METRIC_DENSE, cus_per_chain=cus, metric_storage=storage)
    h.init()
    t0 = time.perf_counter()
    h.run(iters)
    wall = time.perf_counter() - t0
    ms, passes, nbytes, rounds = h.dense_timing()
    lf = h.total_leapfrogs()
    out = dict(metric_storage="f32" if storage else "f64", window_end=h.dense_adapt_timing(), chains=chains, iterations=iters, D=h.D, cus_per_chain=h.cus_per_chain, leapfrogs=lf, wall_s=wall, leapfrogs_per_s=lf / wall,
               matvec_ms=ms, matvec_passes=passes, matvec_gb=nbytes / 1e9, matvec_tb_per_s=nbytes / ms / 1e9, matvec_frac_of_8tbs=nbytes / ms / 1e9 / 8.0,
               rounds=rounds, ms_per_round=1e3 * wall / max(rounds, 1), matvec_share_of_wall=ms * 1e-3 / wall)
    h.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "pieces"
    if what == "pieces":
        pieces()
    elif what == "active":
        if len(sys.argv) > 2:
            ACTIVE_CASES = tuple(sys.argv[2].split(":"))
        if len(sys.argv) > 3:
            SPLIT_CASES = tuple(int(v) for v in sys.argv[3].split(","))
        active_sweep()
    elif what == "pooled":
        pooled_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 41610)
    else:
        chains, iters = int(sys.argv[2]), int(sys.argv[3])
        cus = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        storage = 1 if len(sys.argv) > 5 and sys.argv[5] == "f32" else 0
        if what == "sampler":
            run_sampler(dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"], "full", chains, iters, cus, storage)
        else:
            run_sampler(synthetic.stress(), "full", chains, iters, cus, storage)
