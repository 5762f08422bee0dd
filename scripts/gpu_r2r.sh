#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do for m in coarse fine; do
  MARIGOLD_GN_CHUNKS=$m timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$m', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5))
"
done; done

