import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, synthetic
data = synthetic.small("full")
h = Handle(data, "full", chains=1, num_warmup=150, num_samples=0, seed=11, save_warmup=1, cus_per_chain=16)
h.init(); h.run(100)
eps, minv = h.adaptation()
np.save(sys.argv[1], np.concatenate([eps, minv[0], h.draws()[0, 99, :]]))
