"""Throughput at the BASELINE configs[4] shape (51 x 600 days x 10 000 polls, diagonal metric): development probe."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, synthetic
data = synthetic.stress()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for chains in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "8"])]:
    h = Handle(data, "full", chains=chains, num_warmup=iters, num_samples=0, seed=1843, cus_per_chain=int(os.environ.get("POTUS_K", "0")), twin=0)
    h.init()
    ms_tot, lf_tot = 0.0, 0
    for _ in range(3):
        h.run(iters // 3)
        ms, lf = h.last_run_timing()
        ms_tot += ms; lf_tot += lf
    print(f"stress shape, chains={chains}, cus_per_chain={h.cus_per_chain}: {lf_tot} leapfrogs in {ms_tot:.1f} ms -> {lf_tot/ms_tot*1e3:.0f} leapfrogs/s, "
          f"{ms_tot*1e3*chains/lf_tot:.2f} us/leapfrog/chain", flush=True)
    h.close()
