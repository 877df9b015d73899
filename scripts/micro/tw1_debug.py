#!/usr/bin/env python3
"""Development probe: one workgroup per chain against two, row by row around the first window end, next to the oracle."""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from us_potus_model_amd import Handle, dataprep  # noqa: E402
mode = sys.argv[1]
np.set_printoptions(linewidth=200, precision=10)
if mode == "rows":
    data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
    nw, n = 40, 40
    for cus, twin in ((1, 0), (1, 1), (16, 0)):
        if cus == 1 and twin == 1 and os.environ.get("POTUS_LIB"):
            continue
        h = Handle(data, "full", twin=twin, chains=3, num_warmup=nw, num_samples=5, save_warmup=1, seed=99, cus_per_chain=cus); h.init(); h.run(n)
        d = h.draws()[:, :n]
        for it in range(34, 39):
            print(cus, twin, it, d[0, it, :7])
        h.close()
elif mode == "state":
    import ctypes
    data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
    names = "QC GC PC QA0 QA1 QB0 QB1 PH0 PH1 PF0 PF1 MINV RHOTOP PNEAR WMEAN WM2 SCR0 SCR1".split()
    dumps = {}
    for twin in (0, 1):
        h = Handle(data, "full", twin=twin, chains=3, num_warmup=40, num_samples=5, save_warmup=1, seed=99, cus_per_chain=1); h.init()
        lib = h.L
        lib.potus_debug_state.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.potus_debug_state.restype = ctypes.c_int
        for n in (35, 1):
            h.run(n)
            sz = np.zeros(3)
            lib.potus_debug_state(h.h, 0, sz.ctypes.data_as(ctypes.c_void_p), None)
            st = np.zeros((3 * (1 + twin), int(sz[0]), int(sz[1])))
            lib.potus_debug_state(h.h, 1, st.ctypes.data_as(ctypes.c_void_p), None)
            if twin:
                for k, nm in enumerate(names):
                    if not np.array_equal(st[0, k], st[3, k]):
                        print("  two workgroups, after", n, ": sides differ in", nm, np.abs(st[0, k] - st[3, k]).max())
            eps, minv = h.adaptation()
            dumps[(twin, n)] = (st, np.array(eps), np.array(minv))
            if n == 1 and not twin:
                lpv, g = h.log_prob_grad(st[0, 0, :h.D])
                sc = np.zeros(int(sz[2]), dtype=np.uint8)
                lib.potus_debug_state(h.h, 1, st.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p))
                print("  lp_cur of chain 0", sc.view(np.float64)[6], "lp(QC)", lpv[0])
                S, T, P, M, Pop, Nn, Ns = (int(data[k]) for k in ("S", "T", "P", "M", "Pop", "N_national_polls", "N_state_polls"))
                off = 0
                for nm, ln in (("zT", S), ("Z", S * T), ("c", P), ("m", M), ("pop", Pop), ("mue", 1), ("rho", 1), ("ze", T), ("nn", Nn), ("ns", Ns), ("zb", S)):
                    dd = np.abs(g[0, off:off + ln] - st[0, 1, off:off + ln])
                    print("   block", nm, ln, "max diff", dd.max(), "elements differing", int((dd > 0).sum()))
                    off += ln
                dz = (g[0, S:S + S * T] - st[0, 1, S:S + S * T]).reshape(T, S)
                print("   Z block: days with differences", np.nonzero(np.abs(dz).max(axis=1) > 0)[0][:40], "states", np.nonzero(np.abs(dz).max(axis=0) > 0)[0][:60])
            for cch in range(3):
                lpv, g = h.log_prob_grad(st[cch, 0, :h.D])
                print("  twin", twin, "after", n, "chain", cch, ": |GC - grad(QC)| max", np.abs(g[0] - st[cch, 1, :h.D]).max(), "lp", lpv[0], "|g| max", np.abs(g[0]).max(), "|GC| max", np.abs(st[cch, 1, :h.D]).max())
        h.close()
    for n in (35, 1):
        a, b = dumps[(0, n)], dumps[(1, n)]
        print("after", "35" if n == 35 else "36", "eps", a[1], b[1], "minv equal", np.array_equal(a[2], b[2]), np.abs(a[2] / b[2] - 1).max())
        for k, nm in enumerate(names):
            eq = np.array_equal(a[0][0, k], b[0][0, k], equal_nan=True)
            print("   ", nm, "equal" if eq else ("DIFF max abs %g, nan %d/%d" % (np.nanmax(np.abs(a[0][0, k] - b[0][0, k])), np.isnan(a[0][0, k]).sum(), np.isnan(b[0][0, k]).sum())))
elif mode == "seeds0":
    from us_potus_model_amd import synthetic  # noqa: E402
    data = synthetic.small("full")
    for seed in range(2, 9):
        kw = dict(chains=4, num_warmup=40, num_samples=0, save_warmup=1, seed=seed, cus_per_chain=1, max_depth=(3 if seed == 8 else 10))
        print("seed", seed, flush=True)
        h = Handle(data, "full", twin=0, **kw)
        print(" created", flush=True)
        h.init()
        print(" inited", flush=True)
        for k in range(14):
            h.run(1)
            print("  it", k, h.draws()[:, k, 3].astype(int), h.draws()[:, k, 5].astype(int), flush=True)
        h.close()
else:
    from us_potus_model_amd import synthetic  # noqa: E402
    data = synthetic.small("full")
    for seed in range(1, 9):
        kw = dict(chains=4, num_warmup=40, num_samples=0, save_warmup=1, seed=seed, cus_per_chain=1, max_depth=(3 if seed == 8 else 10))
        for twin in (0, 1):
            print("seed", seed, "twin", twin, flush=True)
            h = Handle(data, "full", twin=twin, **kw)
            h.init()
            for k in range(14):
                h.run(1)
                print("  it", k, h.draws()[:, k, 3].astype(int), h.draws()[:, k, 5].astype(int), flush=True)
            h.close()
