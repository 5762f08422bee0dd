#!/bin/bash
# round 4, session ab: the whole GPU suite + smoke + a bench line on the final commit
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r4ab_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r4ab_smoke.log
timeout 600 python bench.py 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('ms', j['ms_per_step'], 'value', j['value'], 'frac', j['roofline']['frac'], 'cpu', j['cpu_baseline']['value'], j['calibration']['gemm4096_bf16_tflops'], j['calibration']['gemm4096_bf16_tflops_hipblaslt'])" | tee gpurun_out/r4ab_bench.log
