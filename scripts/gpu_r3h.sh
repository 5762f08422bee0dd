#!/bin/bash
# round 3, session h: the whole GPU suite (new: heavy-tailed stress, bit stability at scale, ensembles > 32, odd image sizes)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --timeout=600 --timeout-method=thread > gpurun_out/r3h_t_all.log 2>&1
echo "all gpu tests rc=$?"
grep -E "passed|failed|Error" gpurun_out/r3h_t_all.log | tail -6
grep -E "^\[stress\]|^\[parity\] (pipeline depth at|2-rank)" gpurun_out/r3h_t_all.log
