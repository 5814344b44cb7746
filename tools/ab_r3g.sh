#!/bin/bash
O=gpurun_out/ab6; mkdir -p $O
BASE=/root/repo/m-cedm_amd/_ab/base.so
echo "== 1x1 micro: base / new"; MCEDM_LIB=$BASE python tools/conv1x1_ab.py 2>&1 | grep -v amdgpu; python tools/conv1x1_ab.py 2>&1 | grep -v amdgpu
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > $O/s128_$v.log 2>&1
  echo "s128 $v"; grep '^{' $O/s128_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  echo "train $v"; MCEDM_LIB=$L python tools/train_step_run.py 5 2>&1 | tail -1
done
for v in base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --workload s32 --no-cpu-baseline --no-train --no-secondary --steps 5 --warmup 1 --profile-steps 0 > $O/s32_$v.log 2>&1
  echo "s32 $v"; grep '^{' $O/s32_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
