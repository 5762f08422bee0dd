#!/bin/bash
# the skip concat's two GroupNorm statistics launches as one: parity, then a same-box A/B at E = 1, 2, 10
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/gn2.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --timeout-method=thread -k "groupnorm or two_sources or gn or replay" 2>&1 | tail -5 >> gpurun_out/gn2.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x --timeout=300 --timeout-method=thread 2>&1 | tail -4 >> gpurun_out/gn2.log
for e in 1 2 10; do for round in 1 2; do for f in 0 1; do
  MARIGOLD_TUNING=1 MARIGOLD_GN_STATS_ONE_LAUNCH=$f timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ensemble $e 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('E=$e one_launch=$f', 'ms', j['ms_per_step'], 'groupnorm', round(k['groupnorm']['ms'],2), k['groupnorm']['launches'], 'launches', sum(v['launches'] for v in k.values()))
" >> gpurun_out/gn2.log
done; done; done
cat gpurun_out/gn2.log
