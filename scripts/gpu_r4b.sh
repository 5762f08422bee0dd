#!/bin/bash
# round 4, session b: where are variant 72's wrong outputs?
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python tools/k4w_debug.py 2>&1 | grep -v amdgpu.ids | tail -120
