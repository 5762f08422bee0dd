#!/bin/bash
# round 4, session j: the 192 x 320 hand-placed GEMM tile (variant 73): parity, sweep on the N = 320 k layers; VAE attention chunk A/B
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "igemm or linear" 2>&1 | tail -6
MARIGOLD_IGEMM_SPLITK_ANY=1 SWEEP_VARIANTS=0,46,72,73 SWEEP_NO_FLASH=1 SWEEP_ROUNDS=3 timeout 900 python tools/sweep.py 2>&1 | grep -v amdgpu.ids | tail -32 | tee gpurun_out/r4j_sweep.log
for mb in 0 192; do
  MARIGOLD_VAE_ATTN_CHUNK_MB=$mb timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('CHUNK_MB=$mb', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'stages', {a:round(b['ms'],1) for a,b in j['stages'].items() if isinstance(b,dict) and 'ms' in b})
"
done 2>&1 | tee gpurun_out/r4j_vae_attn_chunk.log
