import sys, os, struct
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from oracle_lib import OracleModel
from us_potus_model_amd import Handle, synthetic, dataprep
data, variant = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"], "full"
K = int(os.environ.get("POTUS_K", "16"))
h = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=5, cus_per_chain=K)
m = OracleModel(data, variant)
rng = np.random.default_rng(5)
q = rng.uniform(-2, 2, (24, h.D))
S, T, P = data["S"], data["T"], data["P"]
names = [("zT", S), ("Z", S * T), ("c", P), ("m", data["M"]), ("pop", data["Pop"]), ("mue", 1), ("rho", 1), ("ze", T), ("nn", data["N_national_polls"]), ("ns", data["N_state_polls"]), ("zb", S)]
blk, o = [], 0
for n, k in names:
    blk.append((n, o, o + k)); o += k
def where(i):
    for n, a, b in blk:
        if a <= i < b: return f"{n}[{i-a}]"
ref = [m.log_prob_grad(q[i]) for i in range(q.shape[0])]
bad = 0
for rep in range(6):
    lp, g = h.log_prob_grad(q)
    for i in range(q.shape[0]):
        err = np.abs(g[i] - ref[i][1])
        idx = np.nonzero(err > 1e-9 * np.abs(ref[i][1]).max())[0]
        for j in idx[:6]:
            bad += 1
            a, b = g[i][j], ref[i][1][j]
            print(f"rep {rep} point {i} {where(j)}: device {a!r} oracle {b!r} diff {a-b:.3e}  bits {struct.pack('>d', a).hex()} {struct.pack('>d', b).hex()}")
print("bad entries", bad)
