#!/bin/bash
O=gpurun_out/ab5; mkdir -p $O
for m in 0 16 0 16; do
  MCEDM_WINO_MODE=$m python bench.py --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 --profile-steps 0 > $O/s128_$m.log 2>&1
  echo "mode=$m"; grep '^{' $O/s128_$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
done
