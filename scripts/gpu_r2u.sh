#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
for round in 1 2; do for f in 0 1; do
  MARIGOLD_XATTN_FUSED=$f timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('fused=$f', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()}, 'launches', sum(v['launches'] for v in k.values()))
"
done; done
