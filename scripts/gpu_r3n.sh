#!/bin/bash
# round 3, session n: K-split cross-attention on the deep levels - kernel parity, standalone timing, pipeline A/B, parity
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "rowgemm" --timeout=300 --timeout-method=thread 2>&1 | tail -2
timeout 300 python tools/xattn_deep_bench.py 2>&1 | grep xattn
for v in 0 1 0 1; do
  MARIGOLD_XATTN_KSPLIT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('XATTN_KSPLIT=$v', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5), 'roofline', j['roofline']['kernel'], j['roofline']['frac'])
"
done
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=900 --timeout-method=thread 2>&1 | tail -3
