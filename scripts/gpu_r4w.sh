#!/bin/bash
# round 4, session w: the 320-channel level's GroupNorm + SiLU fused into the patch convolution (default) vs one apply pass + the plain
# convolution on the 192 x 320 GEMM tile, from Cin 640 / from Cin 320 (interleaved twice)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
one() {
  env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.0))
"
}
for r in 1 2; do one MARIGOLD_UNFUSE_320=0; one MARIGOLD_UNFUSE_320=640; one MARIGOLD_UNFUSE_320=320; done 2>&1 | tee gpurun_out/r4w_ab.log
