#!/bin/bash
# ablation of the Winograd kernel's side work (timeline build): MCEDM_LIB=m-cedm_amd/_ab/tl.so
export MCEDM_LIB=/root/repo/m-cedm_amd/_ab/tl.so
for m in 1 3 5 9 13 15; do echo "mode $m"; MCEDM_WINO_MODE=$m python tools/wino_timeline.py 32 128 128 2>&1 | grep -v amdgpu.ids; done
