#!/bin/bash
# Round-5 session 2: program-driven tile / split-K sweep at the per-GPU ensemble sizes, GroupNorm one-launch threshold A/B.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/sweep_program.py --ensembles 1,2,3,5,10 > gpurun_out/sweep_program.log 2>&1
echo "sweep rc=$?"
grep "per UNet forward" gpurun_out/sweep_program.log
for e in 1 2 5; do for wg in 64 32 16 8; do
  MARIGOLD_TUNING=1 MARIGOLD_GN_SLAB_MIN_WG=$wg timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e gn_slab_min_wg=$wg', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('groupnorm','igemm_mfma')})
"
done; done
