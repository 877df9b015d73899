#!/usr/bin/env python3
"""Development (GPU box): sha256 of the draws and the leapfrog count of a short cluster run with the library POTUS_LIB selects --
two builds that claim the same arithmetic must print the same line.   usage: lib_digest.py [year] [chains] [warmup] [samples]"""
import hashlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep  # noqa: E402

year = sys.argv[1] if len(sys.argv) > 1 else "2016"
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 40
variant = "full" if year == "2016" else "no_mode_adjustment"
data = dataprep.load_npz(ROOT / "tests" / "golden" / f"data_{year}.npz")["data"]
for k, twin in ((16, 1), (16, 0), (8, 0), (1, 0)):
    h = Handle(data, variant, chains=chains, num_warmup=nw, num_samples=ns, seed=1843, cus_per_chain=k, twin=twin)
    h.init()
    h.run(nw + ns)
    d = h.draws()
    print(f"{year} K={k} twin={twin}: leapfrogs {h.total_leapfrogs()} sha256 {hashlib.sha256(d.tobytes()).hexdigest()[:16]}", flush=True)
    h.close()
