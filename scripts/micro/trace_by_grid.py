#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) by kernel and launch grid: calls, mean and total duration.
The dense product is launched with a grid that follows the number of active chains, so this is its time by launch shape.
Usage: trace_by_grid.py <kernel_trace.csv> [name-substring]"""
import csv
import sys
from collections import defaultdict

path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: [0, 0.0])
with open(path, newline="") as f:
    rd = csv.DictReader(f)
    cols = {c.lower(): c for c in rd.fieldnames}
    name_c = cols.get("kernel_name")
    s_c, e_c = cols.get("start_timestamp"), cols.get("end_timestamp")
    gx, gy, wx = cols.get("grid_size_x") or cols.get("grid_size"), cols.get("grid_size_y"), cols.get("workgroup_size_x") or cols.get("workgroup_size")
    for row in rd:
        n = row[name_c]
        if want not in n:
            continue
        key = (n.split("(")[0][:48], int(row[gx]) // max(int(row[wx]), 1) if gx and wx else -1, int(row[gy]) if gy else -1)
        a = acc[key]
        a[0] += 1
        a[1] += (int(row[e_c]) - int(row[s_c])) * 1e-6
print(f"{'kernel':48s} {'wg_x':>6s} {'wg_y':>5s} {'calls':>7s} {'mean_ms':>10s} {'total_ms':>11s}")
for (n, x, y), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:48s} {x:6d} {y:5d} {c:7d} {t / c:10.4f} {t:11.2f}")
