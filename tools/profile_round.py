#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.
usage: profile_round.py STATS_DIR FETCH_DIR WRITE_DIR OUT_DIR TAG [WORKLOAD]
  STATS_DIR  rocprofv3 --kernel-trace --stats --output-format csv
  FETCH_DIR  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv      (own pass)
  WRITE_DIR  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv      (own pass)
Writes OUT_DIR/{TAG_kernel_stats_WORKLOAD.csv, TAG_pmc_fetch_summary_WORKLOAD.csv, TAG_pmc_write_summary_WORKLOAD.csv,
TAG_pmc_traffic_WORKLOAD.json, TAG_profile_meta_WORKLOAD.json} — the names bench.py looks for under profiles/; the meta file carries the
sha of the device sources the profile was taken with (bench.py cites a profile only for the kernels it describes).
FETCH_DIR / WRITE_DIR "-": only the kernel stats are written.
HBM bytes per launch = FETCH_SIZE[KiB] x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE[KiB] x 1024."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return max(hits, key=os.path.getsize)


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def pmc_summary(d, counter, out_path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(find(d, "counter_collection.csv")) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    with open(out_path, "w") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel_Name", "Dispatches", f"{counter}_sum_KiB", f"{counter}_avg_KiB"])
        for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, round(s, 1), round(s / n, 1)])
    return agg


def main():
    stats_dir, fetch_dir, write_dir, out_dir, tag = sys.argv[1:6]
    wl = "_" + sys.argv[6] if len(sys.argv) > 6 else ""
    os.makedirs(out_dir, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import time
    with open(os.path.join(out_dir, f"{tag}_profile_meta{wl}.json"), "w") as fh:
        json.dump(dict(kernels_sha=bench.kernels_sha(), taken=time.strftime("%Y-%m-%d %H:%M:%S"), tool="rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)"), fh)
    rows = list(csv.DictReader(open(find(stats_dir, "kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats{wl}.csv"), "w") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]),
                        int(r["MinNs"]), int(r["MaxNs"])])
    if fetch_dir == "-" or write_dir == "-":                    # kernel stats only (no counter passes in this run)
        return
    fe = pmc_summary(fetch_dir, "FETCH_SIZE", os.path.join(out_dir, f"{tag}_pmc_fetch_summary{wl}.csv"))
    wr = pmc_summary(write_dir, "WRITE_SIZE", os.path.join(out_dir, f"{tag}_pmc_write_summary{wl}.csv"))
    traffic = {}
    for k in sorted(set(fe) | set(wr)):
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0], 1)
        f_kib = fe.get(k, [0, 0.0])[1] / n
        w_kib = wr.get(k, [0, 0.0])[1] / n
        traffic[k] = dict(launches=n, fetch_KiB_raw_per_launch=round(f_kib, 1), write_KiB_per_launch=round(w_kib, 1),
                          hbm_bytes_per_launch_corrected=int(f_kib * 1024 * 2 + w_kib * 1024))
    with open(os.path.join(out_dir, f"{tag}_pmc_traffic{wl}.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)


if __name__ == "__main__":
    main()
