#!/bin/bash
# Short GPU-box session: kernel + pipeline parity tests and smoke() of the current library state.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=150 --timeout-method=thread > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/status.log
timeout 500 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s --timeout=300 --timeout-method=thread > gpurun_out/t_pipe.log 2>&1
echo "pipeline rc=$?" >> gpurun_out/status.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/status.log
cat gpurun_out/status.log; tail -5 gpurun_out/t_kernels.log; grep -n "parity\|passed\|failed\|Error" gpurun_out/t_pipe.log | tail -30; tail -3 gpurun_out/smoke.log
