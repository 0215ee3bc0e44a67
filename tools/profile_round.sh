#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel stats of the same command and the two PMC passes.
# (the profiled runs skip the end-to-end child processes — their kernels would be counted in — and the counter passes the CPU baseline)
# Results land in gpurun_out/prof/ ; the summaries worth judging are then copied into profiles/.
set -e -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp
if [ -z "$SKIP_PLAIN_BENCH" ]; then python3 $R/bench.py > $O/bench.json; echo "bench done"; fi
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --no-end-to-end > $O/bench_under_rocprof.json 2> $O/rocprof_stats.log
python3 $R/tools/profile_round.py /tmp/p_stats - - $O ${TAG:-r03} ${WORKLOAD:-2160p_main10}      # the kernel stats are safe even if a counter pass below runs out of time
echo "stats done"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --steps 1 --no-cpu-baseline --no-profile --no-end-to-end > $O/bench_under_pmc_fetch.json 2> $O/rocprof_fetch.log
echo "fetch done"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --steps 1 --no-cpu-baseline --no-profile --no-end-to-end > $O/bench_under_pmc_write.json 2> $O/rocprof_write.log
echo "write done"
python3 $R/tools/profile_round.py /tmp/p_stats /tmp/p_fetch /tmp/p_write $O ${TAG:-r03} ${WORKLOAD:-2160p_main10}
ls -la $O
