#!/bin/bash
# round 3, session c: flash generation 3 with permuted V^T / matrix-pipe row sums; QKV epilogue's permuted section; bench
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "flash or layernorm_fold or linear_geglu" --timeout=300 --timeout-method=thread > gpurun_out/r3c_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3c_t.log | tail -8
FLASH_DBG=1 FLASH_VARIANTS=6,10,12,13,14,15,16,17,18 timeout 600 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c_flash_bench.log
echo "flash bench rc=$?"
cat gpurun_out/r3c_flash_bench.log
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3c_ops.tsv > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.log
echo "bench rc=$?"
tail -3 gpurun_out/r3c_bench.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c_bench.json').read().strip().splitlines()[-1])
print('ms', j['ms_per_step'], {k:(v['ms'], v['launches']) for k,v in j['kernels'].items() if v['ms']>1})
PY
