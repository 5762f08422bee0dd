#!/bin/bash
# same-box A/B of alternative builds of the library (_ab_libs/lib_<name>.so, loaded through MARIGOLD_HIP_LIB) against the in-tree one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do for lib in default "$@"; do
  if [ $lib == default ]; then unset MARIGOLD_HIP_LIB; else export MARIGOLD_HIP_LIB=$PWD/_ab_libs/lib_$lib.so; fi
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$lib', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()})
"
done; done
