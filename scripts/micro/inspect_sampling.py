import sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep
data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
for K in (16, 1):
    h = Handle(data, "full", chains=8, num_warmup=300, num_samples=60, seed=1843, cus_per_chain=K)
    h.init(); h.run(360)
    d = h.draws()
    print("K", K, "stepsize", h.adaptation()[0])
    print(" treedepth", np.unique(d[:, :, 3], return_counts=True))
    print(" n_leapfrog", np.unique(d[:, :, 4], return_counts=True))
    print(" accept", d[:, :, 1].mean(), "divergent", d[:, :, 5].sum())
    ms, lf = h.last_run_timing()
    h.close()
