#!/usr/bin/env python3
"""WRITE_SIZE per kernel of scripts/micro/write_calib.hip against the bytes each kernel stored:
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/wcal -o r -- scripts/micro/write_calib > gpurun_out/wcal/bytes.txt
    python scripts/micro/pmc_calib.py gpurun_out/wcal WRITE_SIZE"""
import sqlite3
import sys
from pathlib import Path

root = Path(sys.argv[1])
want = {ln.split()[1]: int(ln.split()[2]) for ln in (root / "bytes.txt").read_text().splitlines() if ln.startswith("bytes ")}
db = next(root.rglob("*_results.db"))
con = sqlite3.connect(db)
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tables if t.startswith("rocpd_pmc_event")][0]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
info = [t for t in tables if t.startswith("rocpd_info_pmc")][0]
q = (f"select s.kernel_name, sum(p.value) from {pmc} p join {disp} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id "
     f"join {info} i on p.pmc_id = i.id where i.name = 'WRITE_SIZE' group by s.kernel_name")
print("WRITE_SIZE (rocprofv3 --pmc, taken as KiB) against the bytes each kernel stored (MI355X, gfx950):")
for kn, val in cur.execute(q).fetchall():
    key = next((k for k in sorted(want, key=len, reverse=True) if k in kn), None)
    if key:
        print(f"  {key:20s} stored {want[key] / 2**20:10.1f} MiB   WRITE_SIZE {val:14.1f} -> {val * 1024 / want[key]:6.3f} counted bytes per stored byte")
