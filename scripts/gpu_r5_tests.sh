#!/bin/bash
# the whole GPU suite with its [parity] lines + smoke (+ optional bench)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s --timeout=400 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/t_all.log; grep -n "^FAILED\|Error" gpurun_out/t_all.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
if [ "$1" == "bench" ]; then
  timeout 600 python bench.py --steps 6 --warmup 2 --dump-ops gpurun_out/ops_e10.tsv > gpurun_out/bench_e10.json 2> gpurun_out/bench_e10.log
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/bench_e10.json') if l.startswith('{')][-1])
print(j['ms_per_step'], {k:(round(v['ms'],1),v['launches']) for k,v in j['kernels'].items()}, j['stages'], j['calibration'].get('gemm4096_bf16_tflops'))"
fi
