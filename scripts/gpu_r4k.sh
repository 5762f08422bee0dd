#!/bin/bash
# round 4, session k: whole map A/B of the 192 x 320 hand-placed GEMM tile in the automatic choice (MARIGOLD_K4WB=0/1)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for round in 1 2; do
  for k in 0 1; do
    MARIGOLD_K4WB=$k timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('K4WB=$k', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'stages', {a:round(b['ms'],1) for a,b in j['stages'].items() if isinstance(b,dict) and 'ms' in b})
"
  done
done 2>&1 | tee gpurun_out/r4k_ab_k4wb.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -4
