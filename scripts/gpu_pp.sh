#!/bin/bash
# Short GPU session for the ping-pong GEMM tile (variant 60): its parity tests, then a sweep against the
# current tiles on the shapes where N is a multiple of 256.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "pingpong or linear_geglu" --timeout=120 --timeout-method=thread > gpurun_out/t_pp.log 2>&1
echo "pp tests rc=$?" | tee gpurun_out/status.log
grep -n "pingpong\|v60\|passed\|failed\|Error\|error" gpurun_out/t_pp.log | tail -25
SWEEP_NO_FLASH=1 SWEEP_ONLY=${SWEEP_ONLY:-vae.conv} SWEEP_ROUNDS=${SWEEP_ROUNDS:-3} SWEEP_VARIANTS=${SWEEP_VARIANTS:-34,60,61,62,63} timeout 200 python tools/sweep.py > gpurun_out/sweep_pp.log 2>&1
echo "sweep rc=$?" | tee -a gpurun_out/status.log
cat gpurun_out/sweep_pp.log | tail -30
