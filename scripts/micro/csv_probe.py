#!/usr/bin/env python3
"""Time of potus_write_stan_csv (GPU box): the 2016 fit, 8 chains, N saved draws -> one CmdStan CSV per chain (43 360 columns).
usage: csv_probe.py [draws per chain]"""
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
h = Handle(data, "full", chains=8, num_warmup=60, num_samples=n, seed=1843)
h.init(); h.run(60 + n)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    t = time.time()
    h.write_stan_csv(d, "probe")
    dt = time.time() - t
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
print(f"potus_write_stan_csv: 8 chains x {n} draws x {h.n_cols} columns = {8 * n * h.n_cols / 1e6:.1f} M numbers, {size / 1e9:.2f} GB of text in {dt:.2f} s "
      f"({8 * n * h.n_cols / dt / 1e6:.1f} M numbers/s; the 1000-draw fit: {dt * 1000 / n:.1f} s)")
h.close()
