#!/bin/bash
# round 4, session h: whole map with the four-wave patch convolution / the hand-placed GEMM tile in the automatic choices
# (A/B by MARIGOLD_CP4W / MARIGOLD_IGEMM72_VAE), then the parity suites
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for round in 1 2; do
  for k in 0 1; do
    MARIGOLD_CP4W=$k MARIGOLD_IGEMM72_VAE=$k timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('CP4W=$k', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'ens', j['stages'].get('ensemble',{}).get('ms'), 'stages', {a:round(b['ms'],1) for a,b in j['stages'].items() if isinstance(b,dict) and 'ms' in b})
"
  done
done 2>&1 | tee gpurun_out/r4h_ab_cp4w.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
