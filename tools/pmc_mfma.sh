#!/bin/bash
# MFMA utilisation (rocprofv3 derived counter MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs)) and LDS bank
# conflicts per kernel, for the 128 x 128 and the 32 x 32 workloads (run through gpurun from the repo root; PMC passes
# only, no trace domains).  Reduced by: python tools/rocpd_summary.py counters <db> <counter> ... -> JSON on stdout.
set -o pipefail
tag=${1:-r2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$root/gpurun_out
cd /tmp
for wl in s128 s32; do
  for pmc in MfmaUtil "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    name=${tag}_pmc_${wl}_$(echo $pmc | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $pmc -d $out/$name -o pmc -- python3 $root/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-train --no-graph --profile-steps 0 > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
    echo "$name done"
  done
done
