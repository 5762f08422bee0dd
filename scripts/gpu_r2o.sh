#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== det stress (no-SLP build)"; timeout 300 python tools/det_stress.py 2>&1 | grep -v amdgpu | grep -v " 0/9" ; echo "(lines above = failures; none = all bit-stable)"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
timeout 200 python tools/ln_probe.py 2>&1 | grep -v amdgpu
for round in 1 2; do for lib in slp default; do
  if [ $lib == slp ]; then export MARIGOLD_HIP_LIB=$PWD/_ab_libs/lib_slp.so; else unset MARIGOLD_HIP_LIB; fi
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$lib', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5))
"
done; done
