#!/bin/bash
# round 4, session a: the hand-placed four-wave K loop (tile variant 72): parity tests, then GEMM / conv throughput vs the
# ping-pong tile and hipBLASLt on the same box
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "k4w or two_sources or linear or pingpong or ln_ or geglu" 2>&1 | tail -15
timeout 300 python tools/gemm_bench.py 2>&1 | tail -5 | tee gpurun_out/r4a_gemm.log
SWEEP_VARIANTS=0,62,72 SWEEP_NO_FLASH=1 SWEEP_ROUNDS=3 timeout 900 python tools/sweep.py 2>&1 | tail -30 | tee gpurun_out/r4a_sweep.log
