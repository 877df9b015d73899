#!/usr/bin/env python3
"""fp64 against fp32 storage of the dense inverse metric on the 2016 posterior (GPU box): leapfrogs/s and ESS/s side by side, and
the two posteriors against each other.  (SURVEY 8f rank 4 / VERDICT r02 item 6.)

    python scripts/dense_storage_study.py [chains] [warmup] [sampling] [max_depth] > profiles/r03_dense_storage_study.txt

The full configuration (8 x (1000 + 1000), depth 10) takes a quarter of an hour per storage with a dense metric at D = 15 098; the
defaults here (4 chains, 150 + 100, trees cut at depth 8 for both) keep the comparison to a few minutes.  Same seed, same chains:
up to the first metric update the two runs are the same chains bit for bit (unit metric); after it the fp32 run samples under the
rounded matrix."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, _abi, dataprep, diagnostics as dg  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 100
md = int(sys.argv[4]) if len(sys.argv) > 4 else 8
data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
S, T = int(data["S"]), int(data["T"])
res = {}
print(f"2016 posterior (D = 15 098), dense_e, {chains} chains x ({nw} + {ns}), seed 1843, max_treedepth {md}")
for name, st in (("f64", _abi.STORAGE_F64), ("f32", _abi.STORAGE_F32)):
    h = Handle(data, "full", chains=chains, num_warmup=nw, num_samples=ns, seed=1843, metric=_abi.METRIC_DENSE, metric_storage=st, max_depth=md)
    h.init()
    t0 = time.perf_counter()
    h.run(nw)
    t1 = time.perf_counter()
    lf_w = h.total_leapfrogs()
    h.run(ns)
    t2 = time.perf_counter()
    lf = h.total_leapfrogs()
    a = h.layout["mu_b"][0]
    mu = np.transpose(h.write_array(a + S * (T - 1), a + S * T, ns), (1, 0, 2))          # [chain, draw, S]: mu_b[:, T]
    lp = np.transpose(h.write_array(0, 1, ns), (1, 0, 2))
    cols = np.concatenate([lp, mu, 1.0 / (1.0 + np.exp(-mu))], axis=2)
    ess = min(dg.ess_bulk(cols[:, :, j]) for j in range(cols.shape[2]))
    ms, passes, nbytes, rounds = h.dense_timing()
    acc = np.transpose(h.write_array(1, 2, ns), (1, 0, 2)).mean()
    res[name] = cols
    print(f"  {name}: warm-up {t1 - t0:6.1f} s ({lf_w} leapfrogs), sampling {t2 - t1:6.1f} s ({lf - lf_w} leapfrogs): {lf / (t2 - t0):7.0f} leapfrogs/s overall, "
          f"{(lf - lf_w) / (t2 - t1):7.0f} in the sampling phase; ESS_bulk,min {ess:6.1f} -> {ess / (t2 - t1):6.2f} ESS/s; accept_stat {acc:.3f}; "
          f"matrix pass {nbytes / ms / 1e9:.2f} TB/s, {ms / passes:.3f} ms per pass", flush=True)
    h.close()
x, y = res["f64"], res["f32"]
worst = 0.0
for j in range(1, x.shape[2]):
    a, b = x[:, :, j], y[:, :, j]
    se = np.hypot(a.std() / np.sqrt(max(dg.ess_mean(a), 1.0)), b.std() / np.sqrt(max(dg.ess_mean(b), 1.0)))
    worst = max(worst, abs(a.mean() - b.mean()) / se)
print(f"  posterior means of mu_b[:, T] and predicted_score[T, :] (102 columns): worst |difference| = {worst:.2f} combined MCSE")
