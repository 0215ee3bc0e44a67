#!/usr/bin/env python3
"""Aggregates a rocprofv3 --kernel-trace CSV by kernel and grid shape: for the intra pass this separates the launches of I pictures
(few workgroups, long sub-level chains) from those of B pictures (many workgroups, short chains).
usage: trace_by_grid.py <dir or *_kernel_trace.csv> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "intra_ctu_kernel"
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for fn in files:
    with open(fn) as f:
        for r in csv.DictReader(f):
            if want in r["Kernel_Name"]:
                rows.append(r)
agg = defaultdict(lambda: [0, 0.0, 0])
for r in rows:
    gx, gy, wg = int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"])
    n_wg = (gx // wg) * gy
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    b = 0
    while (1 << (b + 1)) <= n_wg:
        b += 1
    key = (wg // 64, 1 << b)
    agg[key][0] += 1; agg[key][1] += dur; agg[key][2] += n_wg
print(f"{want}: {len(rows)} launches")
print("waves  workgroups>=   launches   total_ms   avg_us   ns_per_workgroup")
tot = sum(v[1] for v in agg.values())
for (w, b), (n, d, wgs) in sorted(agg.items()):
    print(f"{w:5d} {b:13d} {n:10d} {d / 1e3:10.2f} {d / n:8.1f} {1e3 * d / max(wgs, 1):10.1f}   {100 * d / tot:5.1f}%")
