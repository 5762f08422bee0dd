#!/bin/bash
# round 3, session l: 12-wave conv_patch tiles (variants 6, 8, 9) - kernel parity, pipeline A/B (MARIGOLD_PATCH_N320=3: the round-2 tiles), pipeline parity
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv3x3_patch" --timeout=300 --timeout-method=thread 2>&1 | tail -3
for v in 3 0 3 0; do
  MARIGOLD_PATCH_N320=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('PATCH_N320=$v', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=900 --timeout-method=thread 2>&1 | tail -3
