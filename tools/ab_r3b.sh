#!/bin/bash
# A/B runs of round 3 (second session): one gpurun call, same box.  Usage: bash tools/ab_r3b.sh
set -o pipefail
O=gpurun_out/ab1
mkdir -p $O
B="python bench.py --workload s32 --no-cpu-baseline --no-train --no-secondary --steps 5 --warmup 1"
MCEDM_RES_1X1_32=0 $B > $O/s32_base.log 2>&1
$B > $O/s32_p32.log 2>&1
MCEDM_WINO_MIN_HW=256 $B > $O/s32_p32_w256.log 2>&1
MCEDM_RES_1X1_32=0 MCEDM_WINO_MIN_HW=256 $B > $O/s32_w256.log 2>&1
for f in s32_base s32_p32 s32_p32_w256 s32_w256; do echo $f; grep '^{' $O/$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unet_fwd_ms'])"; done
MCEDM_GNBWD_BIG=0 python tools/train_step_run.py 5 > $O/train_base.log 2>&1; tail -1 $O/train_base.log
python tools/train_step_run.py 5 > $O/train_big.log 2>&1; tail -1 $O/train_big.log
MCEDM_GNBWD_BIG=8192 python tools/train_step_run.py 5 > $O/train_big8k.log 2>&1; tail -1 $O/train_big8k.log
python -m pytest tests/test_hip_backward.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
