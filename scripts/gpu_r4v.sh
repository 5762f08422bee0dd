#!/bin/bash
# round 4, session v: the whole GPU test suite on the build with the key-split flash attention
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r4v_tests.log
