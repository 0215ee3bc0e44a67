#!/usr/bin/env python3
"""from a rocprofv3 --kernel-trace CSV: time of the batched (32 lists per launch) preparation kernels against the passes they precede.
usage: trace_prep.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for fn in files:
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        gy, gz = int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"])
        gx = int(r["Grid_Size_X"]) // wg
        batched = gy >= 16 or gz >= 16 or (name in ("prep_finish", "prep_intra_relevel") and gx >= 16)
        key = (name, "batch" if batched else "single")
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot_prep = sum(v[1] for (n, b), v in agg.items() if n.startswith("prep_") and b == "batch")
tot_pass = sum(v[1] for (n, b), v in agg.items() if not n.startswith("prep_") and b == "batch")
for (n, b), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:28s} {b:6s} calls {c:6d} total {us / 1e3:9.2f} ms avg {us / c:8.1f} us")
print(f"batched preparation {tot_prep / 1e3:.1f} ms vs batched passes {tot_pass / 1e3:.1f} ms = {100 * tot_prep / max(tot_pass, 1):.1f} %")
