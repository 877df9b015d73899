#!/usr/bin/env python3
"""Turn rocprofv3's SQLite output (ROCm 7.2 writes <name>_results.db) into the small text
summaries committed under profiles/.  Usage: summarize_rocprof.py <results.db> [<results.db> ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"== {path}")
    try:
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'pct':>7s}")
        for n, c, tot, avg, pct in rows:
            print(f"{n[:60]:60s} {c:6d} {tot/1e3:12.3f} {avg/1e3:12.3f} {pct:7.2f}")
    except sqlite3.Error as e:
        print("  (no kernel stats)", e)
    try:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration), max(vgpr_count), max(sgpr_count), "
                                "max(lds_block_size), max(scratch_size), max(workgroup_size), max(grid_size) from counters_collection "
                                "group by kernel_name, counter_name"))
        if rows:
            print(f"{'kernel':44s} {'counter':12s} {'disp':>5s} {'sum_value':>16s} {'sum_ms':>10s} vgpr sgpr lds scratch wg grid")
        for r in rows:
            print(f"{r[0][:44]:44s} {r[1]:12s} {r[2]:5d} {r[3]:16.1f} {r[4]/1e6:10.3f} {r[5]} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]}")
    except sqlite3.Error as e:
        print("  (no counters)", e)
