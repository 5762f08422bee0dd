#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=400 --timeout-method=thread 2>&1 | tail -3
timeout 300 python tools/det_stress.py 2>&1 | grep -v amdgpu | grep -v " 0/9" ; echo "(det stress: lines above = failures; none = all bit-stable)"
scripts/gpu_ab_old.sh
