#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace of the bench command -> gpurun_out/<tag>_trace/  (per-kernel durations)
#   (4. the same kernel trace for the 32 x 32 workload -> gpurun_out/<tag>_s32_trace/)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) -> gpurun_out/<tag>_pmc_{fetch,write}/
# and reduce them to profiles/<tag>_kernel_stats.csv and profiles/<tag>_traffic.json (tools/rocpd_summary.py).
# The profiled program is python3 itself (no shell / env hop between rocprofv3 and the process that touches the GPU).
set -o pipefail
tag=${1:-r2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$root/gpurun_out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-train --no-graph"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o trace -- python3 $root/bench.py $args > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err || { echo "trace failed"; tail -5 $out/${tag}_trace.err; exit 1; }
echo "trace done"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o fetch -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-train --no-graph --profile-steps 0 > $out/${tag}_pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $out/${tag}_pmc_fetch.log; exit 1; }
echo "pmc fetch done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o write -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-train --no-graph --profile-steps 0 > $out/${tag}_pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $out/${tag}_pmc_write.log; exit 1; }
echo "pmc write done"
# 4. kernel trace of the 32 x 32 workload (BASELINE config 2)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_s32_trace -o trace -- python3 $root/bench.py --workload s32 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-train --no-graph > $out/${tag}_s32_trace_bench.json 2> $out/${tag}_s32_trace.err || { echo "s32 trace failed"; tail -5 $out/${tag}_s32_trace.err; exit 1; }
echo "s32 trace done"
cd $root
find $out/${tag}_trace $out/${tag}_pmc_fetch $out/${tag}_pmc_write -name "*.db" -o -name "*.csv" | head -20
