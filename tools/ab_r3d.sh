#!/bin/bash
set -o pipefail
O=gpurun_out/ab3; mkdir -p $O
for w in ref128 repaint128; do
  for v in base new base new; do
    L=""; [ $v == base ] && L=/root/repo/m-cedm_amd/_ab/base.so
    extra=""; [ $w == repaint128 ] && extra="--steps 1 --warmup 1"
    [ $w == ref128 ] && extra="--steps 3 --warmup 1"
    MCEDM_LIB=$L python bench.py --workload $w --no-cpu-baseline --no-train --no-secondary $extra > $O/${w}_$v.log 2>&1
    echo "$w $v"; grep '^{' $O/${w}_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms'))"
  done
done
