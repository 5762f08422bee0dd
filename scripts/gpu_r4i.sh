#!/bin/bash
# round 4, session i: flash attention generation 2.5 with the row sums on the matrix pipe (variants 22 / 23) against 19 / 20
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FLASH_VARIANTS=19,22,20,23 FLASH_ROUNDS=5 timeout 600 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i_flash.log
