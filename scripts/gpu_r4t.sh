#!/bin/bash
# round 4, session t: key-split launches of the hand-placed flash attention: per-piece loop cycles
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FLASH_DBG=1 FLASH_VARIANTS=26,27 FLASH_ROUNDS=3 timeout 300 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t_flash.log
