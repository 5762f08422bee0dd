#!/bin/bash
# round 3, session r: the alignment optimiser natively - parity with the scipy-driven form, ensembling stage A/B
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -m gpu -q -s -k "ensemble" --timeout=300 --timeout-method=thread 2>&1 | grep -E "parity\] ensemble alignment|passed|failed|Error" | tail -12
timeout 200 python tools/ens_eval_bench.py 2>&1 | tail -1
for v in 0 1 0 1; do
  MARIGOLD_ENS_NATIVE_BFGS=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ENS_NATIVE_BFGS=$v', 'ms', j['ms_per_step'], j['stages'].get('ensemble'))
"
done
