#!/bin/bash
# Same-box A/B of one tuning switch (engine._tune / csrc mg_tuning_int names; honoured under MARIGOLD_TUNING=1 only):
#   gpurun -- 'ENS="10 1 5" KTESTS="fold or igemm_conv3x3" bash scripts/gpu_ab_env.sh MARIGOLD_FOLD_SHORTCUT 0 1'
# runs the kernel tests selected by KTESTS (optional), then bench.py at every ensemble size of ENS with the switch at each given
# value, two interleaved rounds; one line per run in gpurun_out/ab_env_$VAR.log (how profiles/r5_conv_shortcut_fold_ab.log and
# r5_gn_stats_one_launch_ab.log were made).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
VAR=$1; shift
: > gpurun_out/ab_env_$VAR.log
if [ -n "$KTESTS" ]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --timeout-method=thread -k "$KTESTS" 2>&1 | tail -5 >> gpurun_out/ab_env_$VAR.log
fi
for e in ${ENS:-10 1}; do for round in 1 2; do for v in "$@"; do
  env MARIGOLD_TUNING=1 $VAR=$v timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ensemble $e 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('E=$e $VAR=$v', 'ms', j['ms_per_step'], ' '.join(f\"{n}={x['ms']:.2f}/{x['launches']}\" for n,x in k.items() if x['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()})
" >> gpurun_out/ab_env_$VAR.log
done; done; done
cat gpurun_out/ab_env_$VAR.log
