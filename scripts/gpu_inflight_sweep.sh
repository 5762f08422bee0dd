#!/bin/bash
# Maps in flight x ensemble size on one box: scripts/gpu_inflight_sweep.sh "<E list>" "<in-flight list>"
# (per-GPU shards of the member-parallel path are small ensembles: how many lanes they want).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/inflight_sweep.log
: > $LOG
for e in $1; do
  for n in $2; do
    timeout 300 python bench.py --ensemble $e --in-flight $n --steps 12 --warmup 2 --no-cpu-baseline --no-profile 2>gpurun_out/inflight_sweep_err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e in_flight=$n ms_per_map', j['ms_per_step'], 'one_at_a_time', j.get('latency_ms_per_map'), 'gemm', (j.get('calibration') or {}).get('gemm4096_bf16_tflops'))
" >> $LOG
  done
done
cat $LOG; tail -3 gpurun_out/inflight_sweep_err.log
