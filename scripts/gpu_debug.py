import sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep, synthetic
which, step, chains, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
data = synthetic.small("full") if which == "small" else dataprep.load_npz(ROOT / "tests/golden/data_2016.npz")["data"]
h = Handle(data, "full", chains=chains, num_warmup=n, num_samples=0)
print("created", flush=True)
if step == "lp":
    lp, g = h.log_prob_grad(np.zeros((n, h.D)))
    print("lp", lp[:2], flush=True)
elif step == "run":
    h.init(); print("init ok", flush=True)
    for i in range(n):
        h.run(1); print("run ok", i, h.total_leapfrogs(), flush=True)
