#!/bin/bash
# round 4, session c: variant 72 after the MFMA -> accvgpr_read hazard fix: parity tests, schedule variants, sweep with split-K
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "igemm or linear" 2>&1 | tail -8
for lib in "" _k4w1 _k4w2 _k4w3 _k4w4 _k4w5; do
  echo "== lib$lib"
  MARIGOLD_HIP_LIB=$PWD/marigold_amd/libmarigold_hip$lib.so GEMM_VARIANTS=62,72 GEMM_SIZES=4096 GEMM_ROUNDS=5 timeout 200 python tools/gemm_bench.py 2>&1 | grep "^gemm"
  MARIGOLD_HIP_LIB=$PWD/marigold_amd/libmarigold_hip$lib.so SWEEP_ONLY="512->512 @96" SWEEP_VARIANTS=62,72 SWEEP_NO_FLASH=1 SWEEP_ROUNDS=3 timeout 200 python tools/sweep.py 2>&1 | grep "vae.conv"
done 2>&1 | tee gpurun_out/r4c_sched.log
MARIGOLD_IGEMM_SPLITK_ANY=1 SWEEP_VARIANTS=0,62,72 SWEEP_NO_FLASH=1 SWEEP_ROUNDS=3 timeout 900 python tools/sweep.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r4c_sweep.log
