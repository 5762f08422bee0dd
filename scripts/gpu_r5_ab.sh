#!/bin/bash
# same-box A/B: _ab_libs/lib_<name>.so builds against the in-tree library, E = 10 with the per-op table of the LAST run of each
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 --timeout-method=thread -k "${KTESTS:-igemm or linear or conv}" 2>&1 | tail -3
for round in 1 2; do for lib in "$@" default; do
  if [ $lib == default ]; then unset MARIGOLD_HIP_LIB; else export MARIGOLD_HIP_LIB=$PWD/_ab_libs/lib_$lib.so; fi
  timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ensemble ${ENS:-10} --dump-ops gpurun_out/ops_ab_$lib.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$lib', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()})
"
done; done
