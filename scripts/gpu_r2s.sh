#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread -k "conv3x3_patch" 2>&1 | tail -6
PATCH_VARIANTS=3,4,6,7,8,9 PATCH_ONLY="unet 320,unet 640->320,unet 960,vae 128,vae 256->128" timeout 400 python tools/patch_bench.py 2>&1 | grep -v amdgpu
