#!/bin/bash
# round 3, session d: 128 x 128 wave-tile implicit GEMM (igemm2_big.hip) - parity, then the tile sweep on the long-K / wide-N shapes
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "igemm_conv3x3 or linear_geglu or layernorm_fold or dominant" --timeout=300 --timeout-method=thread > gpurun_out/r3d_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3d_t.log | tail -8
SWEEP_NO_FLASH=1 SWEEP_ROUNDS=3 SWEEP_VARIANTS=0,36,51,62,70,71 timeout 900 python tools/sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3d_sweep.log
echo "sweep rc=$?"
cat gpurun_out/r3d_sweep.log
