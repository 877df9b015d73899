#!/usr/bin/env python3
"""Which of the reference's published numbers would notice a wrong model?  (GPU box.)

The README tables pin election-day predicted_score, a function of mu_b[:, T] alone.  This script runs the
2016 backtest on the device with ONE ingredient of the model changed at a time (through the data list: the prior scales of the
mode / population / pollster effects, of the partisan non-response bias and of the measurement noise) and reports how far every
pin moves: worst |delta| of mean / interval ends / P(win) over the 52 rows, the Brier scores, states called correctly and the RMSE
against the certified results -- next to the tolerance the tests use.  A pin that stays inside its tolerance under a 1.5-2 x
change of a scale does not constrain that part of the model; those parts then rest on the transcription test and finite
differences alone (DESIGN.md section 2 quotes the table this prints).

  python scripts/pin_sensitivity.py [chains] [iter] > profiles/r03_pin_sensitivity.txt
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import readme_golden, rmse_ex_dc  # noqa: E402
from us_potus_model_amd import Handle, backtest_scores, dataprep  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
it = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TOL = dict(mean=0.005, low=0.010, high=0.010, prob=0.045, ev_wtd_brier=0.003, unwtd_brier=0.003, rmse_ex_dc=4e-4)
npz = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")
base, meta = npz["data"], npz["meta"]
pub, rows = readme_golden("2016")
states, ev = list(meta["states"]), np.asarray(meta["ev_state"], dtype=np.float64)
won = np.array([int(next(r for r in rows if r["state"] == s)["won_readme"]) for s in states])
cases = [("as published", {}), ("sigma_m x 1.5", {"sigma_m": 1.5}), ("sigma_pop x 1.5", {"sigma_pop": 1.5}), ("sigma_c x 1.5", {"sigma_c": 1.5}),
         ("sigma_e_bias x 2", {"sigma_e_bias": 2.0}), ("sigma_measure_noise_state x 1.5", {"sigma_measure_noise_state": 1.5}),
         ("sigma_measure_noise_national x 1.5", {"sigma_measure_noise_national": 1.5}), ("polling_bias_scale x 1.5", {"polling_bias_scale": 1.5}),
         ("random_walk_scale x 1.5", {"random_walk_scale": 1.5}), ("mu_b_T_scale x 1.5", {"mu_b_T_scale": 1.5})]
print(f"2016 backtest, {chains} chains x ({it} + {it}), seed 1843; tolerances of tests/test_gpu_boundary.py: {TOL}; states correct must equal {int(pub['states_correct'])}")
print(f"{'model':36s} {'mean':>7s} {'low':>7s} {'high':>7s} {'prob':>7s} {'evBrier':>8s} {'Brier':>8s} {'states':>6s} {'rmse':>9s}  pins that notice")
for name, scale in cases:
    d = dict(base)
    for k, f in scale.items():
        d[k] = float(base[k]) * f
    h = Handle(d, "full", chains=chains, num_warmup=it, num_samples=it, seed=1843)
    h.init()
    h.run(2 * it)
    sm = h.posterior_summary(ev)
    T = int(d["T"])
    worst = dict(mean=0.0, low=0.0, high=0.0, prob=0.0)
    for r in rows:
        got = sm["national"][T - 1] if r["state"] == "--" else sm["state"][T - 1, states.index(r["state"])]
        for k, j in (("low", 0), ("high", 1), ("mean", 2), ("prob", 3)):
            worst[k] = max(worst[k], abs(got[j] - float(r[k])))
    sc = backtest_scores(sm, ev, won)
    sc["rmse_ex_dc"] = rmse_ex_dc(states, sm["state"][T - 1, :, 2], rows)
    h.close()
    delta = dict(worst, ev_wtd_brier=abs(sc["ev_wtd_brier"] - pub["ev_wtd_brier"]), unwtd_brier=abs(sc["unwtd_brier"] - pub["unwtd_brier"]),
                 rmse_ex_dc=abs(sc["rmse_ex_dc"] - pub["rmse_ex_dc"]))
    noticed = [k for k in TOL if delta[k] > TOL[k]] + (["states_correct"] if sc["states_correct"] != int(pub["states_correct"]) else [])
    print(f"{name:36s} {delta['mean']:7.4f} {delta['low']:7.4f} {delta['high']:7.4f} {delta['prob']:7.4f} {delta['ev_wtd_brier']:8.5f} {delta['unwtd_brier']:8.5f} "
          f"{sc['states_correct']:6d} {delta['rmse_ex_dc']:9.6f}  {', '.join(noticed) if noticed else '-- none'}", flush=True)
