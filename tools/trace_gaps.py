#!/usr/bin/env python3
"""Diagnostic: read a rocprofv3 --kernel-trace CSV and print, per kernel family, the busy time and, per
queue, how much of the traced interval the queue had a kernel running (the rest is dispatch gaps).
usage: trace_gaps.py <..._kernel_trace.csv>"""
import csv
import collections
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = collections.defaultdict(lambda: [0, 0.0])
    qs = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("<")[0].split("(")[0]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ks[name][0] += 1
        ks[name][1] += (e - s) / 1e3
        qs[r["Queue_Id"]].append((s, e))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    print(f"traced interval {(t1 - t0) / 1e6:.2f} ms, {len(rows)} dispatches, {len(qs)} queues")
    for k, (n, us) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:40s} n={n:7d} total {us / 1e3:9.2f} ms  avg {us / n:8.1f} us")
    # the intra pass by launch size
    big = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if "intra_ctu_kernel" in r["Kernel_Name"]:
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)
            b = 1
            while b < wg:
                b *= 4
            key = (b, int(r["Workgroup_Size_X"]), int(r.get("LDS_Block_Size", 0) or 0) // 8192 * 8)
            big[key][0] += 1
            big[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, (n, us) in sorted(big.items()):
        print(f"    intra launches with <= {k[0]:6d} workgroups of {k[1]:3d} threads, ~{k[2]:3d} KiB LDS: n={n:6d} total {us / 1e3:9.2f} ms avg {us / n:8.1f} us")
    tot_busy = 0.0
    for q, iv in sorted(qs.items()):
        iv.sort()
        busy = sum(e - s for s, e in iv) / 1e3
        gaps = [(iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1)]
        span = (iv[-1][1] - iv[0][0]) / 1e3
        gaps_s = sorted(gaps)
        med = gaps_s[len(gaps_s) // 2] if gaps_s else 0
        print(f"  queue {q}: {len(iv)} dispatches, span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms ({100 * busy / span:.0f}%), "
              f"median gap {med:.1f} us, mean gap {sum(gaps) / max(1, len(gaps)):.1f} us")
        tot_busy += busy
    # concurrency: average number of kernels in flight
    ev = []
    for iv in qs.values():
        for s, e in iv:
            ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur, last, area = 0, t0, 0
    for t, d in ev:
        area += cur * (t - last); last = t; cur += d
    print(f"  average kernels in flight: {area / (t1 - t0):.2f}")


if __name__ == "__main__":
    main()
