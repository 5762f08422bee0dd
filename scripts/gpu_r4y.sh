#!/bin/bash
# round 4, session y: where E = 1 / 2 spend their time (per-class table), with and without the hipGraph replay of the denoising loop
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for e in 1 2; do for g in "" "--graph"; do
  timeout 300 python bench.py --ensemble $e --steps 3 --warmup 1 --no-cpu-baseline $g --dump-ops gpurun_out/r4y_ops_e$e.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('E=$e $g ms', j['ms_per_step'], 'sum of kernels', round(sum(v['ms'] for v in k.values()),1), ' '.join(f\"{n}={v['ms']:.1f}/{v.get('launches',0)}\" for n,v in k.items() if v['ms']>0.5), j.get('stages'))
"
done; done 2>&1 | tee gpurun_out/r4y_small.log
