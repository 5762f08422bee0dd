#!/bin/bash
# round 4, session f: the four-wave hand-placed patch convolution (tile variants 10 / 11): parity, then throughput against the
# 8- / 12-wave tiles on the benchmark shapes
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv3x3_patch" 2>&1 | tail -25
PATCH_VARIANTS=0,10,11 timeout 900 python tools/patch_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f_patch_bench.log
