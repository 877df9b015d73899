#!/usr/bin/env python3
"""A rocprofv3 --pmc counter per kernel against the bytes each kernel of a calibration program moved:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/fcal -o r -- scripts/micro/fetch_calib > gpurun_out/fcal/bytes.txt
    python scripts/micro/pmc_calib.py gpurun_out/fcal FETCH_SIZE > profiles/r04_fetch_size_calibration.txt
(the program prints `bytes <kernel-name substring> <count>` lines; scripts/micro/write_calib.hip / fetch_calib.hip)."""
import sqlite3
import sys
from pathlib import Path

root, counter = Path(sys.argv[1]), sys.argv[2]
want = {ln.split()[1]: int(ln.split()[2]) for ln in (root / "bytes.txt").read_text().splitlines() if ln.startswith("bytes ")}
db = next(root.rglob("*_results.db"))
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tab = lambda prefix: [t for t in tables if t.startswith(prefix)][0]
pmc, disp, sym, info = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
q = (f"select s.kernel_name, sum(p.value), sum(d.end - d.start) from {pmc} p join {disp} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id "
     f"join {info} i on p.pmc_id = i.id where i.name = '{counter}' group by s.kernel_name")
print(f"{counter} (rocprofv3 --pmc, taken as KiB) against the bytes each kernel moved (MI355X, gfx950):")
for kn, val, ns in cur.execute(q).fetchall():
    key = next((k for k in sorted(want, key=len, reverse=True) if k in kn.replace(" ", "")), None)
    if key:
        print(f"  {key:26s} moved {want[key] / 2**20:10.1f} MiB in {ns / 1e6:8.2f} ms ({want[key] / ns:7.1f} GB/s)   {counter} {val:14.1f} -> "
              f"{val * 1024 / want[key]:6.3f} counted bytes per byte")
