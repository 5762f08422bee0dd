#!/bin/bash
# In-program A/B of two tuning tables (_ab/gfx950_A.json = committed, _ab/gfx950_B.json = + the deeper-ring sweep): per-op tables of
# both at E = 1, 2, 3, 5 (two interleaved rounds) - an entry is kept only if its layer got faster INSIDE the program.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ab_tables.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --timeout-method=thread -k "small_ops" 2>&1 | tail -2 >> gpurun_out/ab_tables.log
for round in 1 2; do for t in A B; do
  cp _ab/gfx950_$t.json marigold_amd/tuning/gfx950.json
  for e in 1 2 3 5; do
    timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_${t}${round}_e$e.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$t$round E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],2), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('igemm_mfma','time_embedding')})
" >> gpurun_out/ab_tables.log
  done
done; done
cat gpurun_out/ab_tables.log
