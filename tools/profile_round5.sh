#!/bin/bash
# Round-5 rocprofv3 evidence, collected on the GPU box through gpurun from the repo root:
#   kernel traces (--kernel-trace --stats) of  bench.py (S128), bench.py --workload s32 / repaint128 / ref128 and of the
#   training step (tools/train_step_run.py);  two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) of the S128 bench for
#   roofline.traffic;  one VALU / memory-unit busy pass for the output conv.
# Reduced on the CPU side by tools/rocpd_summary.py / tools/pmc_traffic.py into profiles/r4_*.  The profiled program is
# python3 itself (no shell / env hop between rocprofv3 and the process that touches the GPU).  Which parts: $2 (default all).
set -o pipefail
tag=${1:-r5}
what=${2:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$root/gpurun_out
common="--no-cpu-baseline --no-secondary --no-train --no-graph"
cd /tmp
trace() {   # name, seconds, program args...
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs rocprofv3 --kernel-trace --stats -d $out/${tag}_${name}_trace -o trace -- python3 "$@" > $out/${tag}_${name}_trace_bench.json 2> $out/${tag}_${name}_trace.err \
    || { echo "$name trace failed"; tail -5 $out/${tag}_${name}_trace.err; exit 1; }
  echo "$name trace done"
}
pmc() {     # name, counters, seconds, program args...
  local name=$1 ctr=$2 secs=$3; shift 3
  timeout -k 10 $secs rocprofv3 --pmc $ctr -d $out/${tag}_pmc_${name} -o pmc -- python3 "$@" > $out/${tag}_pmc_${name}.log 2>&1 \
    || { echo "pmc $name failed"; tail -5 $out/${tag}_pmc_${name}.log; return 1; }
  echo "pmc $name done"
}
if [ $what = all ] || [ $what = traces ]; then
  trace s128 400 $root/bench.py --steps 2 --warmup 1 $common
  trace s32 300 $root/bench.py --workload s32 --steps 3 --warmup 1 $common
  trace ref128 300 $root/bench.py --workload ref128 --steps 2 --warmup 1 $common
  trace train 300 $root/tools/train_step_run.py 3
fi
if [ $what = all ] || [ $what = repaint ]; then
  trace repaint 500 $root/bench.py --workload repaint128 --batch 32 --steps 1 --warmup 0 --profile-steps 0 $common
fi
if [ $what = all ] || [ $what = pmc ]; then
  pmc fetch FETCH_SIZE 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || exit 1
  pmc write WRITE_SIZE 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || exit 1
  pmc valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || echo "(VALU pass skipped)"
  pmc mfma MfmaUtil 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || echo "(MfmaUtil pass skipped)"
  pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || echo "(LDS pass skipped)"
  pmc wait "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" 500 $root/bench.py --steps 1 --warmup 0 $common --profile-steps 0 || echo "(wait pass skipped)"
fi
if [ $what = all ] || [ $what = pmctrain ]; then     # round 5: the training step's kernels (wgrad_wino_kernel, conv1x1_reg_kernel, gn_bwd_kernel)
  pmc train_mfma MfmaUtil 400 $root/tools/train_step_run.py 2 || echo "(train MfmaUtil pass skipped)"
  pmc train_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 400 $root/tools/train_step_run.py 2 || echo "(train LDS pass skipped)"
fi
if [ $what = all ] || [ $what = pmc2 ]; then     # the ch = 64 networks (promote these to profiles/) and config 2
  for wl in ref128 s32; do
    pmc ${wl}_mfma MfmaUtil 400 $root/bench.py --workload $wl --steps 1 --warmup 0 $common --profile-steps 0 || echo "($wl MfmaUtil pass skipped)"
    pmc ${wl}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 400 $root/bench.py --workload $wl --steps 1 --warmup 0 $common --profile-steps 0 || echo "($wl LDS pass skipped)"
  done
fi
cd $root
# reduce on the box (gpurun copies at most 64 MiB back; the rocpd databases are far larger) and drop the raw output
red=$out/${tag}_reduced
mkdir -p $red
for w in s128 s32 ref128 train repaint; do
  db=$(find $out/${tag}_${w}_trace -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py stats $db $red/${tag}_${w}_kernel_stats.csv > $red/${tag}_${w}_kernel_stats.txt && cp $out/${tag}_${w}_trace_bench.json $red/
done
fdb=$(find $out/${tag}_pmc_fetch -name "*.db" 2>/dev/null | head -1); wdb=$(find $out/${tag}_pmc_write -name "*.db" 2>/dev/null | head -1)
[ -n "$fdb" ] && [ -n "$wdb" ] && python3 tools/rocpd_summary.py traffic $fdb $wdb $red/${tag}_traffic.json > $red/${tag}_traffic.txt
dbs=""
for c in valu mfma lds wait; do d=$(find $out/${tag}_pmc_$c -name "*.db" 2>/dev/null | head -1); [ -n "$d" ] && dbs="$dbs $d"; done
[ -n "$dbs" ] && python3 tools/rocpd_summary.py kernel_counters $red/${tag}_s128_counters.json $dbs > $red/${tag}_s128_counters.txt
for wl in ref128 s32 train; do
  dbs=""
  for c in mfma lds; do d=$(find $out/${tag}_pmc_${wl}_$c -name "*.db" 2>/dev/null | head -1); [ -n "$d" ] && dbs="$dbs $d"; done
  [ -n "$dbs" ] && python3 tools/rocpd_summary.py kernel_counters $red/${tag}_${wl}_mfma_lds_counters.json $dbs > $red/${tag}_${wl}_mfma_lds_counters.txt
done
rm -rf $out/${tag}_*_trace $out/${tag}_pmc_*
ls -la $red
