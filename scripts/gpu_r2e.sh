#!/bin/bash
# kernel parity + short-K instruction counts (PMC) + short-K timing probes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -5
scripts/gpu_pmc_shortk.sh 2>&1 | grep -v "^SQ_\|rc=" | grep "SQ_WAVES=" 
timeout 200 python tools/store_probe.py 2>&1 | grep -v amdgpu | head -3
timeout 300 python tools/geglu_probe.py 2>&1 | grep -v amdgpu | tail -12
