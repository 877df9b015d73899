#!/usr/bin/env python3
"""Development probe (GPU box): quick timings of the model pass and of NUTS at several chain counts."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep  # noqa: E402

data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
h = Handle(data, "full", chains=1)
rng = np.random.default_rng(0)
for n in (1, 256, 1024):
    q = rng.uniform(-2, 2, (n, h.D))
    h.log_prob_grad(q)
    t = time.perf_counter()
    for _ in range(3):
        h.log_prob_grad(q)
    dt = (time.perf_counter() - t) / 3
    print(f"log_prob_grad n={n}: {dt*1e3:.2f} ms incl. PCIe copies", flush=True)
h.close()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for chains in (1, 8, 64, 256, 512):
    h = Handle(data, "full", chains=chains, num_warmup=iters, num_samples=0, seed=1843)
    h.init()
    ms_tot, lf_tot = 0.0, 0
    for _ in range(3):
        h.run(iters // 3)
        ms, lf = h.last_run_timing()
        ms_tot += ms
        lf_tot += lf
    print(f"NUTS chains={chains}: {lf_tot} leapfrogs in {ms_tot:.1f} ms -> {lf_tot/ms_tot*1e3:.0f} leapfrogs/s, "
          f"{ms_tot*1e3*chains/lf_tot:.2f} us/leapfrog/chain (if all chains were concurrent)", flush=True)
    h.close()
