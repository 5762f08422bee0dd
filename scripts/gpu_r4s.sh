#!/bin/bash
# round 4, session s: whole-map A/B of the hand-placed flash attention (MARIGOLD_FLASH4W=0/1, interleaved twice) + pipeline tests
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
one() {
  MARIGOLD_FLASH4W=$1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('FLASH4W=$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.0))
"
}
for r in 1 2; do one 0; one 1; done 2>&1 | tee gpurun_out/r4s_ab.log
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 | tee gpurun_out/r4s_tests.log
