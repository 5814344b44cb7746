#!/bin/bash
BASE=/root/repo/m-cedm_amd/_ab/base.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for wl in s32 s128; do
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --workload $wl --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > gpurun_out/dpp2_$v.log 2>&1
  echo "$wl $v"; grep '^{' gpurun_out/dpp2_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
done
for v in base new; do L=""; [ $v == base ] && L=$BASE; echo "train $v"; MCEDM_LIB=$L python tools/train_step_run.py 5 2>&1 | tail -1 | cut -c1-50; done
