#!/bin/bash
# conv_shortcut folded into conv2: kernel parity, then a same-box A/B (MARIGOLD_FOLD_SHORTCUT 0 / 1) at E = 10, 1, 5
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/fold.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --timeout-method=thread -k "fold or igemm_conv3x3" 2>&1 | tail -15 >> gpurun_out/fold.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x --timeout=300 --timeout-method=thread 2>&1 | tail -8 >> gpurun_out/fold.log
for e in 10 1 5; do for round in 1 2; do for f in 0 1; do
  MARIGOLD_TUNING=1 MARIGOLD_FOLD_SHORTCUT=$f timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ensemble $e --dump-ops gpurun_out/ops_fold${f}_e$e.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('E=$e fold=$f', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.2f}\" for n,v in k.items() if v['ms']>1.5), {a: round(b['ms'],1) for a,b in j['stages'].items()})
" >> gpurun_out/fold.log
done; done; done
cat gpurun_out/fold.log
