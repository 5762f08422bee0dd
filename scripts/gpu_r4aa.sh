#!/bin/bash
# round 4, session aa: MG_OP_FLASH_ATTN512 on two-wave workgroups for small launches; every VAE attention on it
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "flash_attn512" 2>&1 | tail -4 | tee gpurun_out/r4aa_tests.log
timeout 300 python tools/flash512_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4aa_flash512.log
one() {
  env $1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if n in ('softmax','flash_attn512')), {a:round(b['ms'],2) for a,b in j['stages'].items()})
"
}
for r in 1 2; do one MARIGOLD_VAE_FLASH_SMALL=0; one MARIGOLD_VAE_FLASH_SMALL=1; done 2>&1 | tee gpurun_out/r4aa_ab.log
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 | tee -a gpurun_out/r4aa_tests.log
