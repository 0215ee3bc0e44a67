#!/usr/bin/env python3
"""from a rocprofv3 --kernel-trace --memory-copy-trace CSV directory: per hardware queue, inside a time window given as fractions of the
trace: share of the time a kernel of that queue was running, which kernel families ran there, and the idle gaps in front of every
mc_kernel launch (= the start of a batch's passes on that stream): how long the stream waited and what finished last before it.
usage: trace_queues.py <dir> <from_fraction> <to_fraction>"""
import csv, glob, os, sys
from collections import defaultdict
d, f0, f1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0], r["Queue_Id"]) for r in csv.DictReader(open(fn))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * f0, t0 + (t1 - t0) * f1
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = hi - lo
byq = defaultdict(list)
for r in rows: byq[r[3]].append(r)
print(f"window {span/1e6:.1f} ms, {len(rows)} dispatches, {len(byq)} queues")
for q, rs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    busy = sum(e - s for s, e, _, _ in rs)
    fam = defaultdict(float)
    for s, e, n, _ in rs: fam["prep" if n.startswith("prep_") else n] += e - s
    top = ", ".join(f"{n} {100*v/span:.0f}%" for n, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6])
    gaps = []
    prev_end = None
    for s, e, n, _ in rs:
        if n == "mc_kernel" and prev_end is not None and prev_name != "mc_kernel": gaps.append(s - prev_end)
        prev_end, prev_name = e, n
    g = sorted(gaps)
    gs = f"gaps before a batch's first mc_kernel: n {len(g)}, median {g[len(g)//2]/1e3:.0f} us, mean {sum(g)/len(g)/1e3:.0f} us, max {g[-1]/1e3:.0f} us, summed {100*sum(g)/span:.1f}% of the window" if g else ""
    print(f"queue {q}: busy {100*busy/span:.1f}%  [{top}]  {gs}")
