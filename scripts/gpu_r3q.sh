#!/bin/bash
# round 3, session q: GroupNorm statistics chunking (A/B: MARIGOLD_GN_CHUNKS_R2=1 = the round-2 rule), parity
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 1 0 1 0; do
  MARIGOLD_GN_CHUNKS_R2=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('GN_CHUNKS_R2=$v', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=900 --timeout-method=thread 2>&1 | tail -3
