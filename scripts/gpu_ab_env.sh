#!/bin/bash
# Same-box A/B of engine / library tuning switches: scripts/gpu_ab_env.sh "<name>" "<ENV=.. for A>" "<ENV=.. for B>" [bench args]
# Interleaved twice; prints ms per map and the per-class kernel times of each run.  MARIGOLD_TUNING=1 is set for both sides.
export PYTHONUNBUFFERED=1 MARIGOLD_TUNING=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
NAME=$1; A=$2; B=$3; shift 3
LOG=gpurun_out/ab_$NAME.log
: > $LOG
for round in 1 2; do
  for side in A B; do
    if [ $side == A ]; then E="$A"; else E="$B"; fi
    env $E timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_${NAME}_$side.tsv "$@" 2>gpurun_out/ab_${NAME}_err_$side.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$side [$E] round=$round ms_per_map', j['ms_per_step'], 'alone', j.get('latency_ms_per_map'), {k: round(v['ms'],2) for k,v in j.get('stages',{}).items()}, 'launches', sum(v['launches'] for v in j['kernels'].values()), {k: (round(v['ms'],2), v['launches']) for k,v in j['kernels'].items() if v['ms'] > 1.0}, 'gemm', (j.get('calibration') or {}).get('gemm4096_bf16_tflops'))
" >> $LOG
  done
done
cat $LOG; tail -3 gpurun_out/ab_${NAME}_err_B.log
