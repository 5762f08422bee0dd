#!/bin/bash
# same-box A/B with per-op tables: gpurun_out/ops_old.tsv (the _ab_old worktree) and gpurun_out/ops_new.tsv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for side in _ab_old .; do
  tag=$([ $side == . ] && echo new || echo old)
  (cd $R/$side && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops $R/gpurun_out/ops_$tag.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$tag', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'stages', {a: round(b['ms'],1) for a,b in j['stages'].items()}, 'ens', j['stages'].get('ensemble'))
")
done
