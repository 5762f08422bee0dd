#!/bin/bash
# small-ensemble (multi-GPU shard sizes) step time with and without hipGraph replay of the denoising loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in 1 2 3 5; do for g in "" "--graph"; do
  timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline --no-profile $g 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e graph=\"$g\"', 'ms', j['ms_per_step'], 'stages', {k: round(v['ms'],1) for k,v in j.get('stages',{}).items()})
"
done; done
