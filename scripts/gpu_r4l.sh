#!/bin/bash
# round 4, session l: same-box A/B of the plain 640-channel 3x3 convolutions on the 192 x 320 GEMM tile (MARIGOLD_IGEMM73_CONV)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for round in 1 2; do
  for k in 0 1; do
    MARIGOLD_IGEMM73_CONV=$k timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('IGEMM73_CONV=$k', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'calib', j['calibration'].get('gemm4096_bf16_tflops'), j['calibration'].get('gemm4096_bf16_tflops_hipblaslt'), j['calibration'].get('shader_mhz_under_mfma_load'))
"
  done
done 2>&1 | tee gpurun_out/r4l_ab_igemm73_conv.log
