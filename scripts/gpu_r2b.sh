#!/bin/bash
# Round 2, session B: the patch-resident conv3x3 kernel - parity, then timing against the implicit GEMM.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --timeout=120 --timeout-method=thread -x -k "patch or two_sources or subpixel" > gpurun_out/t_patch.log 2>&1
echo "patch tests rc=$?" | tee -a gpurun_out/status.log
grep -n "parity\|passed\|failed\|Error\|assert" gpurun_out/t_patch.log | tail -40
PATCH_VARIANTS=${PATCH_VARIANTS:-1,2,3,4} timeout 300 python tools/patch_bench.py > gpurun_out/patch_bench.log 2>&1
echo "patch bench rc=$?" | tee -a gpurun_out/status.log
cat gpurun_out/patch_bench.log | tail -40
