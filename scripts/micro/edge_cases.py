import sys, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2])); sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2] / 'tests'))
from us_potus_model_amd import Handle, synthetic
from oracle_lib import OracleModel
cases = {
  "no_national": dict(S=6, T=24, N_state=70, N_national=0, P=9),
  "no_state": dict(S=6, T=24, N_state=0, N_national=25, P=9),
  "one_state": dict(S=1, T=24, N_state=30, N_national=10, P=4),
  "two_days": dict(S=6, T=2, N_state=20, N_national=5, P=3),
  "one_pollster": dict(S=6, T=24, N_state=70, N_national=25, P=1),
  "sparse_days": dict(S=6, T=200, N_state=12, N_national=3, P=3),
  "many_states": dict(S=63, T=30, N_state=200, N_national=20, P=9),
}
for name, kw in cases.items():
    for variant in ("full", "no_mode_adjustment"):
        for cus in (1, 8):
            try:
                data = synthetic.make(seed=3, variant=variant, **kw)
                h = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, save_warmup=1, seed=3, cus_per_chain=cus)
                m = OracleModel(data, variant)
                q = np.random.default_rng(1).uniform(-2, 2, (3, h.D))
                lp, g = h.log_prob_grad(q)
                err = 0.0
                for i in range(3):
                    lpo, go = m.log_prob_grad(q[i])
                    err = max(err, abs(lp[i] - lpo) / max(1.0, abs(lpo)), np.abs(g[i] - go).max() / max(1e-300, np.abs(go).max()))
                h.init(); h.run(4)
                d = h.draws()
                ref = m.sample_chain(1, m.default_opts(num_warmup=10, num_samples=0, save_warmup=1, seed=3, fast_grad=1))[0][:4]
                ok = np.array_equal(d[0][:4, 3:6], ref[:, 3:6]) and np.allclose(d[0][:4, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
                print(f"{name:14s} {variant:20s} cus={cus} D={h.D:5d} lp/grad err {err:.1e} nuts {'ok' if ok else 'MISMATCH'}")
                h.close()
            except Exception as e:
                print(f"{name:14s} {variant:20s} cus={cus} ERROR {type(e).__name__}: {str(e)[:150]}")
