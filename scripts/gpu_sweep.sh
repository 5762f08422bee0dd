#!/bin/bash
# Short GPU session: kernel parity tests + tile/attention sweep.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --timeout=150 --timeout-method=thread > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/status.log
timeout 400 python tools/sweep.py > gpurun_out/sweep.log 2>&1
echo "sweep rc=$?" >> gpurun_out/status.log
cat gpurun_out/status.log; tail -3 gpurun_out/t_kernels.log
