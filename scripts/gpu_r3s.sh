#!/bin/bash
# round 3, session s: ensembling evaluation - polling the stream vs hipStreamSynchronize (stage time / evaluations)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 0 1 0 1; do
  MARIGOLD_ENS_SPIN=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); e=j['stages']['ensemble']; print('ENS_SPIN=$v', 'ms', j['ms_per_step'], e, 'us/eval', round(e['ms']*1000/e['cost_evaluations'],1))
"
done
