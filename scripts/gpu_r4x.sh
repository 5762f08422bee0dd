#!/bin/bash
# round 4, session x: small ensembles per GPU (E = 1, 2, 3, 5, 8) on the final build
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for e in 1 2 3 5 8; do
  timeout 300 python bench.py --ensemble $e --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e ms', j['ms_per_step'], 'maps/s', j['value'])
"
done 2>&1 | tee gpurun_out/r4x_small_ensembles.log
