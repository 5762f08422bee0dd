#!/bin/bash
# round 3, session e: one-launch GroupNorm (MG_OP_GN_SLAB) - parity, pipeline parity, A/B against the chunked passes
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "groupnorm" --timeout=300 --timeout-method=thread > gpurun_out/r3e_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3e_t.log | tail -8
for gn in 0 1; do
  MARIGOLD_GN_SLAB=$gn timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3e_ops_gn$gn.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('GN_SLAB=$gn', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=600 --timeout-method=thread > gpurun_out/r3e_t_pipe.log 2>&1
echo "pipeline tests rc=$?"
tail -5 gpurun_out/r3e_t_pipe.log
