#!/bin/bash
# round 3, session f: conv_patch with two fragment register sets + counted LDS waits - parity, isolated timings, whole map
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "patch or conv3x3 or dominant" --timeout=300 --timeout-method=thread > gpurun_out/r3f_t.log 2>&1
echo "tests rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r3f_t.log | tail -8
timeout 600 python tools/patch_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3f_patch_bench.log
cat gpurun_out/r3f_patch_bench.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3f_ops.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}/{v['launches']}\" for n,v in k.items() if v['ms']>1.5))
"
