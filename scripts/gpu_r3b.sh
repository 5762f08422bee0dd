#!/bin/bash
# round 3, session b: flash-attention generation 3 variants (fragment read order / hand-placed stream)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "flash" --timeout=300 --timeout-method=thread > gpurun_out/r3b_t_flash.log 2>&1
echo "flash tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3b_t_flash.log | tail -8
FLASH_VARIANTS=${FLASH_VARIANTS:-6,9,10,11,12,13,14} timeout 600 python tools/flash_bench.py > gpurun_out/r3b_flash_bench.log 2>&1
echo "flash bench rc=$?"
cat gpurun_out/r3b_flash_bench.log
