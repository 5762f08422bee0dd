#!/bin/bash
# round 4, session u: whole-map A/B of the key split (MARIGOLD_FLASH4W_SPLIT=0/1) and of the hand-placed flash kernel (MARIGOLD_FLASH4W=0),
# then the whole GPU test suite on this build
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
one() {
  env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.0))
"
}
for r in 1 2; do one MARIGOLD_FLASH4W=0; one MARIGOLD_FLASH4W_SPLIT=0; one MARIGOLD_FLASH4W_SPLIT=1; done 2>&1 | tee gpurun_out/r4u_ab.log
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r4u_tests.log
