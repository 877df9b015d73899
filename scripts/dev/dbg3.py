import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from oracle_lib import OracleModel
from us_potus_model_amd import Handle, synthetic, dataprep
np.set_printoptions(linewidth=200, precision=6)
name = sys.argv[1]; K = int(sys.argv[2]); twin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if name == "stress": data, variant, seed, nw = synthetic.stress(), "full", 5, 4
else: data, variant, seed, nw = dataprep.load_npz(ROOT / "tests" / "golden" / f"data_{name}.npz")["data"], ("full" if name == "2016" else "no_mode_adjustment"), 1843, 30
h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=seed, cus_per_chain=K, twin=twin)
print("K", h.cus_per_chain)
h.init(); h.run(3)
d = h.draws()[:, :3]
m = OracleModel(data, variant)
o = m.default_opts(num_warmup=nw, num_samples=0, save_warmup=1, seed=seed, fast_grad=1)
for c in (0, 1):
    ref = m.sample_chain(c + 1, o)[0][:3]
    print("chain", c, "device\n", d[c][:, :7], "\noracle\n", ref[:, :7])
