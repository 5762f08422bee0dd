#!/bin/bash
# mid-size ensembles after the 192 x 320 tile's threshold change: E = 6, 7, 8 before / after + sweep for the table
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for thr in 200 120; do for e in 5 6 7 8 10; do
  MARIGOLD_TUNING=1 MARIGOLD_IGEMM73_CONV_MIN_TILES=$thr timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('min_tiles=$thr E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j['kernels'].items() if k in ('igemm_mfma','conv3x3_patch')})
"
done; done
timeout 900 python tools/sweep_program.py --ensembles 6,7,8 --variants 23,35,22,32,36,46,51,62,72,73 --rounds 3 --iters 8 --vae --emit-db gpurun_out/gfx950_e678.json > gpurun_out/sweep_program_e678.log 2>&1
grep "per UNet forward\|table entries" gpurun_out/sweep_program_e678.log
