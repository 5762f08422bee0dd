#!/bin/bash
# Round 2, session D: everything after the cleanup (generation-1 GEMM removed, scheduler tail, colorize, fp32 boundary)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout=400 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "all gpu tests rc=$?" | tee -a gpurun_out/status.log
tail -15 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/status.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_r2d.tsv > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.log
echo "bench rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_r2d.json'));print(d['value'],d['ms_per_step'],{k:(v['launches'],v['ms']) for k,v in d['kernels'].items()})" 2>/dev/null)" | tee -a gpurun_out/status.log
for e in 1 2; do
timeout 200 python bench.py --steps 5 --warmup 2 --ensemble $e --no-cpu-baseline --no-profile > gpurun_out/bench_e$e.json 2> gpurun_out/bench_e$e.log
echo "bench E=$e rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_e$e.json'));print(d['value'],d['ms_per_step'])" 2>/dev/null)" | tee -a gpurun_out/status.log
done
