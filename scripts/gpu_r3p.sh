#!/bin/bash
# round 3, session p: small ensembles (the shard sizes of a member-parallel run) on the final build
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in 1 2 3 5 8; do
  timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e', 'ms', j['ms_per_step'], 'maps/s', j['value'])
"
done
