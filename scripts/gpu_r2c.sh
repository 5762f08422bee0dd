#!/bin/bash
# Round 2, session C: engine with the patch conv / fused norms: pipeline parity, then whole-map A/B on one box.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -s -x --timeout=400 --timeout-method=thread > gpurun_out/t_pipe.log 2>&1
echo "pipeline+fullsize rc=$?" | tee -a gpurun_out/status.log
grep -n "parity\|property\|passed\|failed\|Error\|assert" gpurun_out/t_pipe.log | tail -60
for cfg in "0 none" "1 none" "1 auto" "1 all"; do
  set -- $cfg
  MARIGOLD_PATCH_CONV=$1 MARIGOLD_FUSE_GN=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/ops_p$1_$2.tsv > gpurun_out/bench_p$1_$2.json 2> gpurun_out/bench_p$1_$2.log
  echo "bench patch=$1 fuse=$2 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_p$1_$2.json'));print(d['value'],d['ms_per_step'],{k:(v['launches'],v['ms']) for k,v in d['kernels'].items()})" 2>/dev/null)" | tee -a gpurun_out/status.log
done
