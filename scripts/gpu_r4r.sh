#!/bin/bash
# round 4, session r: cycle counts of the hand-placed flash key loop (s_memtime per wave) + parity of the build
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "hand_placed" 2>&1 | tail -4 | tee gpurun_out/r4r_tests.log
FLASH_DBG=1 FLASH_VARIANTS=25,26,27 FLASH_ROUNDS=5 timeout 300 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4r_flash.log
