#!/bin/bash
# round 3, session i: row-resident GEMM (MG_OP_ROWGEMM) - kernel parity, standalone timing, pipeline A/B, pipeline parity
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "rowgemm" --timeout=300 --timeout-method=thread > gpurun_out/r3i_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3i_t.log | tail -8
timeout 300 python tools/rowgemm_bench.py 10 12 8 > gpurun_out/r3i_rowgemm_bench.log 2>&1; cat gpurun_out/r3i_rowgemm_bench.log | grep -v amdgpu.ids
for rg in 0 1 0 1; do
  MARIGOLD_ROWGEMM=$rg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3i_ops_rg$rg.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('ROWGEMM=$rg', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
done
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=900 --timeout-method=thread > gpurun_out/r3i_t_pipe.log 2>&1
echo "pipeline tests rc=$?"
tail -5 gpurun_out/r3i_t_pipe.log
