#!/bin/bash
# round 4, session n: MG_OP_FLASH_ATTN512 (VAE mid-block attention as a flash kernel): parity tests, kernel A/B against the
# three-stage form, whole-map A/B (MARIGOLD_VAE_FLASH=0/1, interleaved twice)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "flash_attn512" 2>&1 | tail -5 | tee gpurun_out/r4n_tests.log
timeout 300 python tools/flash512_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4n_flash512.log
one() {
  MARIGOLD_VAE_FLASH=$1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('VAE_FLASH=$1 ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.0))
"
}
for r in 1 2; do one 0; one 1; done 2>&1 | tee gpurun_out/r4n_ab.log
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -x 2>&1 | tail -3 | tee -a gpurun_out/r4n_tests.log
