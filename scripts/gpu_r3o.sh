#!/bin/bash
# round 3, session o: GroupNorm fusion policy with the 12-wave convolution tiles (auto | all | none)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in auto all none auto all none; do
  MARIGOLD_FUSE_GN=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('FUSE_GN=$v', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
