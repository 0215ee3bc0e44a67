#!/usr/bin/env python3
"""from a rocprofv3 --kernel-trace CSV of ONE timed mode: the interval between the first and last pass kernel of the steady part,
the time the GPU had at least one kernel running (union over all queues), the time >= 2 / >= 3 queues were running kernels,
and the summed kernel time per family.  usage: trace_active.py <dir> [skip_fraction]"""
import csv, glob, os, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for fn in files:
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0], r["Queue_Id"]) for r in csv.DictReader(open(fn))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + int((t1 - t0) * skip)                       # drop start-up / warm-up
rows = [r for r in rows if r[0] >= lo]
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, at = 0, ev[0][0], defaultdict(float)
for t, d in ev:
    at[min(depth, 4)] += t - last
    last = t
    depth += d
span = ev[-1][0] - ev[0][0]
print(f"interval {span / 1e6:.1f} ms; kernels running: none {100 * at[0] / span:.1f} %, one {100 * at[1] / span:.1f} %, two {100 * at[2] / span:.1f} %, three {100 * at[3] / span:.1f} %, four or more {100 * at[4] / span:.1f} %")
fam = defaultdict(float)
for s, e, n, q in rows:
    fam["preparation" if n.startswith("prep_") else n] += e - s
for n, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {n:28s} {v / 1e6:9.1f} ms summed = {v / span:5.2f} x the interval")
print(f"  all kernels summed = {sum(fam.values()) / span:.2f} x the interval")
if len(sys.argv) > 3:                                  # timeline: per window of N ms the share of time with >= 1 kernel and the summed kernel time
    win = float(sys.argv[3]) * 1e6
    allrows = []
    for fn in files:
        allrows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fn))]
    allrows.sort()
    b0 = allrows[0][0]
    nwin = int((max(r[1] for r in allrows) - b0) / win) + 1
    summed = [0.0] * nwin; prep = [0.0] * nwin; cnt = [0] * nwin
    marks = []
    for s, e, n in allrows:
        w = int((s - b0) / win)
        summed[w] += e - s
        cnt[w] += 1
        if "prep_" in n: prep[w] += e - s
        marks.append((s, 1)); marks.append((e, -1))
    marks.sort()
    busy = [0.0] * nwin
    depth, last = 0, b0
    for t, d in marks:
        if depth > 0:
            a = last
            while a < t:
                w = int((a - b0) / win)
                nxt = min(t, b0 + (w + 1) * win)
                busy[w] += nxt - a
                a = nxt
        last = t
        depth += d
    for w in range(nwin):
        print(f"  window {w:3d}: busy {100 * busy[w] / win:5.1f} %  summed kernel time {summed[w] / win:4.2f} x  preparation {prep[w] / win:4.2f} x  dispatches {cnt[w]}")
