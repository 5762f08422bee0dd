#!/bin/bash
# round 3, session k: deep-level conv tiles (A/B), full GPU suite
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for d in 0 62 0 62; do
  MARIGOLD_DEEP_TILE=$d timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('DEEP_TILE=$d', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 --timeout-method=thread > gpurun_out/r3k_t.log 2>&1
echo "gpu tests rc=$?"
tail -4 gpurun_out/r3k_t.log
