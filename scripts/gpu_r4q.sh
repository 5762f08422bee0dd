#!/bin/bash
# round 4, session q: the hand-placed flash attention (variant 26, flash4w.hip): parity, then TFLOP/s against variant 25
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "hand_placed" 2>&1 | tail -15 | tee gpurun_out/r4q_tests.log
FLASH_VARIANTS=25,26 FLASH_ROUNDS=5 timeout 300 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4q_flash.log
