#!/bin/bash
# round 4, session ad: GroupNorm by-product also from the 16 x 16 x 256 tile (the VAE's sub-pixel up-sampling); final checks
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r4ad_tests.log
for r in 1 2; do for v in 0 1; do
  MARIGOLD_GN_BYPRODUCT=$v timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('GN_BYPRODUCT=$v ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if n in ('groupnorm','conv3x3_patch')), {a:round(b['ms'],2) for a,b in j['stages'].items()})
"
done; done 2>&1 | tee gpurun_out/r4ad_ab.log
