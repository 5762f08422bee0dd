#!/bin/bash
# round 4, session o: flash_attn64 occupancy steps (time against the number of 4-wave workgroups), cost of the ln_out hand-off
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FLASH_OCC=1 FLASH_OCC_BH=1,2,3,4,5,6,7 FLASH_VARIANTS=25,20,21 FLASH_ROUNDS=3 timeout 300 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4o_flash_occ2.log
