#!/bin/bash
# round 4, session ac: split-K combined in the launch (the tile's last split sums the slabs and runs the epilogue) vs the splitk_reduce launch
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "igemm or conv or linear or split" 2>&1 | tail -4 | tee gpurun_out/r4ac_tests.log
one() {
  env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v.get('launches',0)}\" for n,v in k.items() if v['ms']>1.0))
"
}
for r in 1 2; do one MARIGOLD_SPLITK_FUSED=0; one MARIGOLD_SPLITK_FUSED=1; done 2>&1 | tee gpurun_out/r4ac_ab.log
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 | tee -a gpurun_out/r4ac_tests.log
