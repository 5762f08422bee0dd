#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.log; tail -1 gpurun_out/bench_final2.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], {a: round(b['ms'],1) for a,b in j['stages'].items()}, j['stages']['ensemble'])"
