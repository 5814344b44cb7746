#!/bin/bash
BASE=/root/repo/m-cedm_amd/_ab/base.so
python -m pytest tests/test_hip_wino.py tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -1
for v in base new base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > gpurun_out/dpp_$v.log 2>&1
  echo "s128 $v"; grep '^{' gpurun_out/dpp_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
for v in base new; do
  L=""; [ $v == base ] && L=$BASE
  MCEDM_LIB=$L python bench.py --workload ref128 --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > gpurun_out/dppr_$v.log 2>&1
  echo "ref128 $v"; grep '^{' gpurun_out/dppr_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
