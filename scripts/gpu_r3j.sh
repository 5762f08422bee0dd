#!/bin/bash
# round 3, session j: rowgemm at small ensembles (wave count / column split per M) - kernel tests, then whole maps per E with
# the kernel off / on
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "rowgemm" --timeout=300 --timeout-method=thread 2>&1 | tail -3
for e in 1 2 3 5 10; do for rg in 0 1; do
  MARIGOLD_ROWGEMM=$rg timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e ROWGEMM=$rg', 'ms', j['ms_per_step'], 'stages', {k: round(v['ms'],1) for k,v in j.get('stages',{}).items()})
"
done; done
