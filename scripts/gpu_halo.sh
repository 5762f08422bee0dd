#!/bin/bash
# First GPU session of round 2: the experimental halo-shared 3x3 convolution tile (csrc/igemm3.hip).
#   1. gated parity tests of variants 70-73; 2. interleaved timing against the current tiles;
#   3. end-to-end A/B of the whole map with MARIGOLD_HALO_CONV=0/1/2 (bench without CPU baseline / profile).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
MG_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_gpu_experimental.py -m gpu -q -s --timeout=120 --timeout-method=thread > gpurun_out/t_halo.log 2>&1
echo "halo tests rc=$?" | tee gpurun_out/status.log
grep -n "parity\|passed\|failed\|Error" gpurun_out/t_halo.log | tail -40
HALO_BUDGET_S=60 timeout 120 python tools/halo_check.py > gpurun_out/halo_check.log 2>&1
echo "halo check rc=$?" | tee -a gpurun_out/status.log
tail -12 gpurun_out/halo_check.log
for mode in 0 1 2; do
  MARIGOLD_HALO_CONV=$mode timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/bench_halo$mode.json 2> gpurun_out/bench_halo$mode.log
  echo "bench halo=$mode rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_halo$mode.json'));print(d['value'],d['ms_per_step'])" 2>/dev/null)" | tee -a gpurun_out/status.log
done
# 4. tile variant 49 (128x320, BK = 32, two workgroups per CU) on the K = C linears of the 320-channel level
SWEEP_NO_FLASH=1 SWEEP_ONLY="unet.linear" SWEEP_ROUNDS=3 SWEEP_VARIANTS=0,46,49,53 timeout 120 python tools/sweep.py > gpurun_out/sweep_v49.log 2>&1
echo "sweep v49 rc=$?" | tee -a gpurun_out/status.log
tail -8 gpurun_out/sweep_v49.log
