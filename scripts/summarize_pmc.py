#!/usr/bin/env python3
"""Per-kernel sums of every counter of the --pmc passes under a directory (rocprofv3 SQLite output), with the bench line of
each pass beside it.  usage: summarize_pmc.py gpurun_out/icache_<tag>"""
import json
import sqlite3
import sys
from pathlib import Path

root = Path(sys.argv[1])
for db in sorted(root.rglob("*_results.db")):
    p = db.relative_to(root).parts[0]
    line = None
    try:
        line = json.loads((root / f"bench_{p}.json").read_text().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"== {p}: no bench line ({e})")
    cur = sqlite3.connect(db).cursor()
    try:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration), max(workgroup_size), max(grid_size) "
                                "from counters_collection group by kernel_name, counter_name"))
    except sqlite3.Error as e:
        print(f"== {p}: no counters ({e})")
        continue
    print(f"== {p}")
    if line:
        rf = line["roofline"]
        print(f"   bench: {line['value']:.0f} leapfrogs/s, timed launches {rf['launch_ms_total']:.1f} ms, leapfrogs in them {rf['leapfrogs_in_launches']}")
    for kn, cn, n, val, dur, wg, grid in rows:
        if "k_cl_run" not in kn and not kn.startswith("k_run"):
            continue
        print(f"   {kn[:40]:40s} {cn:28s} dispatches {n:3d} sum {val:18.1f} launch_ms {dur/1e6:10.2f} wg {wg} grid {grid}")
