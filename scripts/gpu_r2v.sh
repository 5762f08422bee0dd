#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_r2y.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()})
"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
