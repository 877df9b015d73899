#!/usr/bin/env python3
"""HBM traffic per leapfrog from the two counter passes of scripts/profile_round.sh.

usage: pmc_traffic.py gpurun_out/prof_<tag>  > profiles/rNN_cl_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are summed over the k_cl_run (or k_run) launches of each pass (rocprofv3 --pmc, one
counter per pass) and divided by the leapfrogs of the same launches: the timed launches report their leapfrogs in
the bench line; the untimed warm-up launch is apportioned by launch time.  FETCH_SIZE is doubled, as the MI355X
guide prescribes for gfx950 (128-byte requests are tallied as 64).
"""
import json
import sqlite3
import sys
from pathlib import Path

root = Path(sys.argv[1])


def counter(db, name):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tables if t.startswith("rocpd_pmc_event")][0]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    info = [t for t in tables if t.startswith("rocpd_info_pmc")][0]
    q = (f"select s.kernel_name, count(*), sum(p.value), sum(d.end - d.start) * 1e-6 from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id join {info} i on p.pmc_id = i.id where i.name = ? group by s.kernel_name")
    rows = cur.execute(q, (name,)).fetchall()
    for kn, n, val, ms in rows:
        if "k_cl_run" in kn or kn.startswith("k_run"):
            return ("k_cl_run" if "k_cl_run" in kn else "k_run"), n, val, ms
    raise SystemExit(f"no sampler kernel in {db}")


out = {}
res = {}
for tag, cname, fname in (("fetch", "FETCH_SIZE", "bench_fetch.json"), ("write", "WRITE_SIZE", "bench_write.json")):
    kn, n, val, ms = counter(next((root / f"pmc_{tag}").rglob("*_results.db")), cname)
    line = json.loads((root / fname).read_text().strip().splitlines()[-1])
    lf_timed, ms_timed = line["roofline"]["leapfrogs_in_launches"], line["roofline"]["launch_ms_total"]
    lf_all = lf_timed * ms / ms_timed            # + the warm-up launch, by time
    res[tag] = dict(kernel=kn, launches=n, kb=val, ms=ms, leapfrogs=lf_all, command=f"python bench.py --steps {line['steps']} --no-cpu-baseline",
                    cfg=line["config"].get("parallelism", ""))
f, w = res["fetch"], res["write"]
fb = 2.0 * f["kb"] * 1024.0 / f["leapfrogs"]
wb = w["kb"] * 1024.0 / w["leapfrogs"]
print(json.dumps({
    "kernel": f["kernel"], "command": f["command"] + " (" + f["cfg"] + ")",
    "fetch_size_kb_sum": f["kb"], "write_size_kb_sum": w["kb"], "launches": f["launches"],
    "launch_ms_fetch_pass": f["ms"], "launch_ms_write_pass": w["ms"],
    "leapfrogs_fetch_pass": f["leapfrogs"], "leapfrogs_write_pass": w["leapfrogs"],
    "fetch_bytes_per_leapfrog": fb, "write_bytes_per_leapfrog": wb, "hbm_bytes_per_leapfrog": fb + wb,
    "note": "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes; calibrated for the sampler's load forms at 0.500 counted bytes per streamed byte: profiles/r04_fetch_size_calibration.txt); WRITE_SIZE taken as KiB (1.000 counted bytes per stored byte: profiles/r03_write_size_calibration.txt); "
            "leapfrogs of the untimed warm-up launch estimated from launch time"}, indent=1))
