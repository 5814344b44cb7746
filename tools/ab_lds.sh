#!/bin/bash
BASE=/root/repo/m-cedm_amd/_ab/base.so
python -m pytest tests/test_hip_wino.py -m gpu -x -q 2>&1 | tail -2
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  echo "wino micro $v"; MCEDM_LIB=$L python tools/wino_spread.py 32 128 128 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150
done
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > gpurun_out/lds_$v.log 2>&1
  echo "s128 $v"; grep '^{' gpurun_out/lds_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
