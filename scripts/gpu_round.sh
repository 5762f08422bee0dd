#!/bin/bash
# One GPU-box session: kernel parity tests, full-size bench (+per-op table), rocprof kernel stats,
# pipeline parity tests, smoke.  Everything is bounded by timeouts; logs land in gpurun_out/.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/status.log
nproc >> gpurun_out/status.log; lscpu | grep "Model name" >> gpurun_out/status.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --timeout=150 --timeout-method=thread > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/status.log
timeout 300 python tools/sweep.py > gpurun_out/sweep.log 2>&1
echo "sweep rc=$?" >> gpurun_out/status.log
timeout 900 python bench.py --steps 3 --warmup 1 --dump-ops gpurun_out/ops_full.tsv > gpurun_out/bench_full.json 2> gpurun_out/bench_full.log
echo "bench rc=$?" >> gpurun_out/status.log
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_bench.log 2>&1)
echo "rocprof rc=$?" >> gpurun_out/status.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s --timeout=240 --timeout-method=thread > gpurun_out/t_pipe.log 2>&1
echo "pipeline rc=$?" >> gpurun_out/status.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/status.log
ls -la gpurun_out/prof 2>/dev/null | head >> gpurun_out/status.log
# keep only the stats CSVs from the trace (the raw kernel trace can be large)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
cat gpurun_out/status.log
