#!/bin/bash
# round 3, session a: flash-attention generation 3 - parity tests, variants side by side, then the baseline bench of this box
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "flash" --timeout=300 --timeout-method=thread > gpurun_out/r3a_t_flash.log 2>&1
echo "flash tests rc=$?"
tail -5 gpurun_out/r3a_t_flash.log
timeout 600 python tools/flash_bench.py > gpurun_out/r3a_flash_bench.log 2>&1
echo "flash bench rc=$?"
cat gpurun_out/r3a_flash_bench.log
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3a_ops.tsv > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.log
echo "bench rc=$?"
tail -3 gpurun_out/r3a_bench.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3a_bench.json').read().strip().splitlines()[-1])
print('ms', j['ms_per_step'], {k:v['ms'] for k,v in j['kernels'].items() if v['ms']>1})
print(j['calibration'])
PY
