#!/bin/bash
# round 3, session g: flash generation 2.5 (high-occupancy loop + VALU diet)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "flash" --timeout=300 --timeout-method=thread > gpurun_out/r3g_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3g_t.log | tail -8
FLASH_VARIANTS=6,14,17,18,19,20,21 timeout 600 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3g_flash_bench.log
cat gpurun_out/r3g_flash_bench.log
