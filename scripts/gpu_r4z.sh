#!/bin/bash
# round 4, session z: GroupNorm statistics as a by-product of the producing convolution (VAE 12-wave tiles): parity, whole-map A/B
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "output_groupnorm_statistics or conv3x3_patch" 2>&1 | tail -6 | tee gpurun_out/r4z_tests.log
one() {
  env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v.get('launches',0)}\" for n,v in k.items() if v['ms']>1.0), {a:round(b['ms'],1) for a,b in j['stages'].items()})
"
}
for r in 1 2; do one MARIGOLD_GN_BYPRODUCT=0; one MARIGOLD_GN_BYPRODUCT=1; done 2>&1 | tee gpurun_out/r4z_ab.log
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 | tee -a gpurun_out/r4z_tests.log
