import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from oracle_lib import OracleModel
from us_potus_model_amd import Handle, synthetic, dataprep

def blocks(h, data, variant):
    S, T, P = data["S"], data["T"], data["P"]
    full = variant == "full"
    names = [("zT", S), ("Z", S * T), ("c", P)] + ([("m", data["M"]), ("pop", data["Pop"]), ("mue", 1), ("rho", 1), ("ze", T)] if full else []) + \
            [("nn", data["N_national_polls"]), ("ns", data["N_state_polls"]), ("zb", S)]
    out, o = [], 0
    for n, k in names:
        out.append((n, o, o + k)); o += k
    return out

which = sys.argv[1] if len(sys.argv) > 1 else "stress"
if which == "stress":
    data, variant = synthetic.stress(), "full"
else:
    data, variant = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"], "full"
K = int(os.environ.get("POTUS_K", "0"))
h = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=5, cus_per_chain=K)
print("K", h.cus_per_chain, "D", h.D)
m = OracleModel(data, variant)
rng = np.random.default_rng(5)
q = np.vstack([np.zeros((1, h.D)), rng.uniform(-2, 2, (2, h.D))])
lp, g = h.log_prob_grad(q)
lp2, g2 = h.log_prob_grad(q)
print("deterministic lp/grad:", np.array_equal(lp, lp2), np.array_equal(g, g2), np.abs(g - g2).max())
for i in range(q.shape[0]):
    lpo, go = m.log_prob_grad(q[i])
    err = np.abs(g[i] - go)
    print(f"point {i}: lp rel {abs(lp[i]-lpo)/abs(lpo):.2e} grad max err {err.max():.3e} of {np.abs(go).max():.3e}")
    for n, a, b in blocks(h, data, variant):
        e = err[a:b]
        print(f"   {n:4s} max err {e.max():.3e} at {int(e.argmax())} (|g| max {np.abs(go[a:b]).max():.3e})")
# NUTS determinism
runs = []
for rep in range(2):
    hh = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=5, cus_per_chain=K)
    hh.init(); hh.run(12)
    runs.append(hh.draws()[:, :12].copy()); hh.close()
print("NUTS same bytes:", np.array_equal(runs[0], runs[1]), "first differing iteration:",
      next((int(i) for i in range(12) if not np.array_equal(runs[0][:, i], runs[1][:, i])), None))
print(runs[0][0, :, 3:5].T, runs[1][0, :, 3:5].T)
