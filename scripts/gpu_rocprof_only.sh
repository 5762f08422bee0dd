#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r2 -- python $R/bench.py --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_bench.log 2>&1)
echo "rocprof rc=$?"; find gpurun_out -name "*kernel_trace*" -size +30M -delete 2>/dev/null; ls gpurun_out/prof; tail -1 gpurun_out/rocprof_bench.log | cut -c1-200
