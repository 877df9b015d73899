#!/usr/bin/env python3
"""Development check (GPU box) of the cluster path: log-density/gradient of potus_cluster.hpp
against the oracle for several cluster sizes, then short NUTS runs against the one-workgroup path."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from us_potus_model_amd import Handle, dataprep, synthetic  # noqa: E402
from oracle_lib import OracleModel  # noqa: E402

Ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 8]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d16 = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
cases = [("2016/full", d16, "full"), ("2016/nomode", d16, "no_mode_adjustment"), ("small/full", synthetic.small(), "full")]
rng = np.random.default_rng(5)
for name, data, variant in cases:
    om = OracleModel(data, variant)
    for K in Ks:
        h = Handle(data, variant, chains=1, cus_per_chain=K, num_warmup=10, num_samples=0)
        q = rng.uniform(-2, 2, size=(3, h.D))
        q[2] *= 0.1
        lp, g = h.log_prob_grad(q)
        worst = 0.0
        for b in range(3):
            lo, go = om.log_prob_grad(q[b])
            el = abs(lp[b] - lo) / max(1.0, abs(lo))
            eg = np.max(np.abs(g[b] - go) / np.maximum(1.0, np.abs(go)))
            worst = max(worst, el, eg)
            if eg > 1e-9:
                bad = np.argsort(-np.abs(g[b] - go))[:5]
                print("   worst grad idx", bad, g[b][bad], go[bad])
        print(f"{name} K={K}: lp/grad max rel err {worst:.2e}", flush=True)
        h.close()

print("NUTS, 2016/full, 2 chains")
ref = None
for K in [1] + Ks:
    h = Handle(d16, "full", chains=2, cus_per_chain=K, num_warmup=iters, num_samples=10, seed=1843, save_warmup=1)
    t0 = time.time()
    h.init()
    h.run(iters + 10)
    ms, lf = h.last_run_timing()
    dr = h.draws()
    print(f" K={K}: {lf} leapfrogs, {ms:.1f} ms -> {ms*1e3*2/max(lf,1):.2f} us/leapfrog/chain; wall {time.time()-t0:.2f}s; "
          f"stepsize {h.adaptation()[0]}", flush=True)
    print("   n_leapfrog chain0:", dr[0, :12, 4].astype(int), " lp:", np.round(dr[0, :4, 0], 6))
    if ref is None:
        ref = dr
    else:
        n = min(6, dr.shape[1])
        print("   first-6 max |diff| vs K=1:", float(np.max(np.abs(dr[:, :n, :] - ref[:, :n, :]))))
    h.close()
