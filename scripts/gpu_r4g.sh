#!/bin/bash
# round 4, session g: conv_patch4w after the TRANS -> VALU hazard fix (fix-up slices over taps 2-8)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv3x3_patch" 2>&1 | tail -8
PATCH_VARIANTS=0,10,11 timeout 900 python tools/patch_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g_patch_bench.log
