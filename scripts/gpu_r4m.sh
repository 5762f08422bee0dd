#!/bin/bash
# round 4, session m: flash row sums as plain v_add_f32 (variant 25) vs packed (19); bench line with the calibration on variant 72;
# per-op check of the 1280 -> 10240 GEGLU
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FLASH_VARIANTS=19,25 FLASH_ROUNDS=5 timeout 300 python tools/flash_bench.py 2>&1 | grep -v amdgpu.ids | head -3 | tee gpurun_out/r4m_flash.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r4m_ops.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'calib', j['calibration'])
"
grep "up_blocks.1.attentions.1.transformer_blocks.0.ff" gpurun_out/r4m_ops.tsv | awk -F'\t' '{print $2,$4,$7}' | head -4
