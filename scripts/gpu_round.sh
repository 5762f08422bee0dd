#!/bin/bash
# One GPU-box session: kernel parity tests, full-size bench (+per-op table), rocprof kernel stats,
# PMC passes (HBM traffic), pipeline parity tests, smoke.  Everything is bounded by timeouts; logs land
# in gpurun_out/.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/status.log
nproc >> gpurun_out/status.log; lscpu | grep "Model name" | head -1 >> gpurun_out/status.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --timeout=200 --timeout-method=thread > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/status.log
timeout 900 python bench.py --steps 4 --warmup 1 --dump-ops gpurun_out/ops_full.tsv > gpurun_out/bench_full.json 2> gpurun_out/bench_full.log
echo "bench rc=$?" >> gpurun_out/status.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -s --timeout=400 --timeout-method=thread > gpurun_out/t_pipe.log 2>&1
echo "pipeline rc=$?" >> gpurun_out/status.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/status.log
rm -rf gpurun_out/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_bench.log 2>&1)
echo "rocprof rc=$?" >> gpurun_out/status.log
if [ "$1" == "pmc" ]; then
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o r2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1)
  echo "pmc fetch rc=$?" >> gpurun_out/status.log
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o r2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write.log 2>&1)
  echo "pmc write rc=$?" >> gpurun_out/status.log
fi
ls -la gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write 2>/dev/null | head -20 >> gpurun_out/status.log
find gpurun_out -name "*kernel_trace*" -size +30M -delete 2>/dev/null
cat gpurun_out/status.log
