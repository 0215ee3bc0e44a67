#!/bin/bash
# Runs on the GPU box: instruction-mix counters per kernel of one bench step (own PMC pass, kernels serialised by the profiler).
set -e -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/p_insts -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-end-to-end > $O/bench_insts.json 2> $O/rocprof_insts.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_busy -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-end-to-end > $O/bench_busy.json 2> $O/rocprof_busy.log
python3 - <<PY
import csv, glob, collections, os
for d in ("/tmp/p_insts", "/tmp/p_busy"):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getsize)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    with open("$O/" + os.path.basename(d) + "_summary.csv", "w") as fh:
        names = sorted({c for v in agg.values() for c in v})
        fh.write("kernel," + ",".join(names) + "\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            fh.write('"' + k + '",' + ",".join(f"{v.get(c, 0):.0f}" for c in names) + "\n")
    print(open("$O/" + os.path.basename(d) + "_summary.csv").read())
PY
