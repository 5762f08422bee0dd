#!/bin/bash
# Round-5 session 3: the tuning table (two interleaved rounds per candidate), then A/B of the table + GroupNorm rule + row-resident GEMM threshold.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tools/sweep_program.py --ensembles 1,2,3,4,5,8,10 --variants 23,35,22,32,36,46,51,62,72,73 --rounds 3 --iters 8 --vae --emit-db gpurun_out/gfx950.json > gpurun_out/sweep_program.log 2>&1
echo "sweep rc=$?"
grep "per UNet forward\|table entries" gpurun_out/sweep_program.log
cp gpurun_out/gfx950.json marigold_amd/tuning/gfx950.json
run() { # name, env...
  local name=$1; shift
  for e in 1 2 3 5 10; do
    env MARIGOLD_TUNING=1 "$@" timeout 300 python bench.py --ensemble $e --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$name E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('groupnorm','igemm_mfma','rowgemm_mfma')})
"
  done
}
run heuristics MARIGOLD_TUNING_DB=0 MARIGOLD_GN_SLAB_SMALL_KB=0
run table MARIGOLD_GN_SLAB_SMALL_KB=0
run table+gn X=1
run table+gn+norowgemm_small MARIGOLD_ROWGEMM_MIN_M=20000
