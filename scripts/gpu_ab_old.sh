#!/bin/bash
# same-box A/B: the tree at HEAD (_ab_old, a git worktree with its own built library) against the working tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
for round in 1 2; do
  for side in _ab_old .; do
    (cd $R/$side && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$side', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'ens', j['stages'].get('ensemble',{}).get('ms'))
")
  done
done
