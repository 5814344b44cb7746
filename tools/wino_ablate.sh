#!/bin/bash
# In-kernel timeline of the Winograd kernels (timeline build: tools/build_ab.sh tl "-DMCEDM_WINO_TIMELINE" conv_wino.hip conv_wino1.hip):
#   mode 1   wave 0's cycles per chunk: top / MFMA stream / barrier / epilogue
#   mode 3   the same without the weight reloads (wrong results; the one ablation left since the side work runs unconditionally)
#   mode 16  cycles per SLOT of a stage (8 MFMAs + one slice of side work each)
#   mode 32  cycles per phase of the epilogue (nu-transform + requests, exchange rounds, stores, statistics, accumulator init)
#   MCEDM_WINO1=1: the one-wave-per-SIMD kernel (K loop per chunk, epilogue per tile)
export MCEDM_LIB=${MCEDM_LIB:-/root/repo/m-cedm_amd/_ab/tl.so}
for m in 1 3 16 32; do echo "mode $m"; MCEDM_WINO1=0 MCEDM_WINO_MODE=$m python tools/wino_timeline.py 32 128 128 2>&1 | grep -v amdgpu.ids; done
echo "one wave per SIMD"; MCEDM_WINO1=1 python tools/wino_timeline.py 32 128 128 2>&1 | grep -v amdgpu.ids
