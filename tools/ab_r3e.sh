#!/bin/bash
set -o pipefail
O=gpurun_out/ab4; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_module.py tests/test_hip_ddpm.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in 0 1 0 1; do
  MCEDM_SMALL_COUT_V4=$v python bench.py --workload ref128 --no-cpu-baseline --no-train --no-secondary --steps 3 --warmup 1 > $O/ref_$v.log 2>&1
  echo "v4=$v"; grep '^{' $O/ref_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('unet_fwd_ms')); print([ (k['name'][:40], k['launches'], round(k['total_ms']/k['launches']*1e3,1)) for k in d['kernels'] if 'small' in k['name']])"
done
