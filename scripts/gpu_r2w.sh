#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 --timeout-method=thread 2>&1 | tail -3
scripts/gpu_ab_libs.sh before
