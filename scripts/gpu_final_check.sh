#!/bin/bash
# last check of the tree as it will be judged: pipeline / full-size parity, smoke, one full bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 4 --warmup 1 --dump-ops gpurun_out/ops_final.tsv > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log; tail -1 gpurun_out/bench_final.json | cut -c1-400
