#!/bin/bash
set -o pipefail
O=gpurun_out/ab2; mkdir -p $O
B="python bench.py --workload s32 --no-cpu-baseline --no-train --no-secondary --steps 5 --warmup 1"
for v in 1 2 1 2; do MCEDM_RES_1X1_32=$v $B > $O/s32_p$v.log 2>&1; echo "p32=$v"; grep '^{' $O/s32_p$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unet_fwd_ms'])"; done
MCEDM_RES_1X1_32=2 python -m pytest tests/test_hip_parity.py tests/test_hip_module.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
