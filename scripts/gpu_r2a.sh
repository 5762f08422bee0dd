#!/bin/bash
# Round 2, session A: new kernel-level cases (dominant shapes, sub-pixel up-conv), full-size parity vs the committed
# oracle goldens, bench with the per-op table.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --timeout=200 --timeout-method=thread -k "dominant or subpixel or benchmark_shape" > gpurun_out/t_kernels_new.log 2>&1
echo "new kernel tests rc=$?" | tee -a gpurun_out/status.log
grep -n "parity\|passed\|failed\|Error" gpurun_out/t_kernels_new.log | tail -30
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s --timeout=400 --timeout-method=thread > gpurun_out/t_fullsize.log 2>&1
echo "fullsize rc=$?" | tee -a gpurun_out/status.log
grep -n "parity\|property\|passed\|failed\|Error\|assert" gpurun_out/t_fullsize.log | tail -40
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_r2a.tsv > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.log
echo "bench rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_r2a.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'])" 2>/dev/null)" | tee -a gpurun_out/status.log
