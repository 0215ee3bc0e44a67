#!/usr/bin/env python3
"""from a rocprofv3 --kernel-trace --memory-copy-trace CSV directory: per window of W ms — share of the time with 0 / 1 / 2 / 3 / >= 4
kernels running, summed kernel time, preparation launches, host-to-device bytes and the share of the time at least one copy was running.
Tells the decode region's bound apart: GPU busy all the time (the passes), copy engine busy all the time (PCIe), or neither (the host).
usage: trace_decode.py <dir> [W=50]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
W = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 50e6
k = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    k += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]) for r in csv.DictReader(open(fn))]
c = []
for fn in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        c.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), int(r.get("Size", r.get("Bytes", 0)) or 0)))
k.sort(); c.sort()
b0 = min(k[0][0], c[0][0] if c else k[0][0])
n = int((max(r[1] for r in k) - b0) / W) + 1
def cover(iv, n):
    """per window: time with depth 0,1,2,3,4+ and summed time"""
    ev = []
    for s, e in iv: ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    out = [[0.0] * 5 for _ in range(n)]
    depth, last = 0, b0
    for t, dd in ev:
        a = last
        while a < t:
            w = int((a - b0) / W); end = min(t, b0 + (w + 1) * W)
            out[w][min(depth, 4)] += end - a; a = end
        last = t; depth += dd
    return out
kc = cover([(s, e) for s, e, _ in k], n)
cc = cover([(s, e) for s, e, dr, _ in c if "HOST_TO_DEVICE" in dr.upper() or "H2D" in dr.upper() or dr == ""], n)
prep = defaultdict(int); h2d = defaultdict(float); summed = defaultdict(float)
for s, e, nm in k:
    w = int((s - b0) / W); summed[w] += e - s
    if nm.startswith("prep_finish"): prep[w] += 1
for s, e, dr, b in c:
    if "DEVICE_TO_HOST" in dr.upper(): continue
    h2d[int((s - b0) / W)] += b
print("window   none   one   two  three  >=4  | summed  | prep sets | H2D MB   GB/s  copy busy %   >=2 copies %")
for w in range(n):
    t = kc[w]; cw = cc[w]
    print(f"{w:5d}  {100*t[0]/W:5.1f} {100*t[1]/W:5.1f} {100*t[2]/W:5.1f} {100*t[3]/W:5.1f} {100*t[4]/W:5.1f}  | {summed[w]/W:5.2f}x  | {prep[w]:6d}    | {h2d[w]/1e6:7.1f} {h2d[w]/W:6.1f}   {100*(W-cw[0])/W:5.1f}        {100*sum(cw[2:])/W:5.1f}")
