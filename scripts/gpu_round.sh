#!/bin/bash
# One full GPU-box session of a round: the whole GPU suite with its [parity] lines, smoke, the headline bench (+ per-op table),
# the same bench under torchrun with the RCCL path forced on one rank, the small ensembles (per-GPU shards of 2 / 4 / 8 GPUs),
# rocprofv3 kernel stats and - with "pmc" - the two HBM-traffic counter passes.  Everything is bounded by timeouts; logs land in
# gpurun_out/ (copy what is to be judged into profiles/).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/status.log
nproc >> gpurun_out/status.log; lscpu | grep "Model name" | head -1 >> gpurun_out/status.log
timeout 1200 python -m pytest tests -m gpu -q -s --timeout=400 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/status.log; tail -1 gpurun_out/t_all.log >> gpurun_out/status.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/status.log
timeout 900 python bench.py --steps 12 --warmup 2 --dump-ops gpurun_out/ops_full.tsv > gpurun_out/bench_full.json 2> gpurun_out/bench_full.log
echo "bench rc=$?" >> gpurun_out/status.log
MARIGOLD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/bench_nccl_world1.json 2> gpurun_out/bench_nccl_world1.log
echo "bench nccl rc=$?" >> gpurun_out/status.log
: > gpurun_out/small_ensembles.log
for e in 1 2 3 5 8; do
  timeout 300 python bench.py --ensemble $e --steps 6 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_e$e.tsv 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('E=$e', 'ms_per_map', j['ms_per_step'], 'one_at_a_time', j.get('latency_ms_per_map'), 'stages', {k: round(v['ms'],1) for k,v in j.get('stages',{}).items()}, 'launches', sum(v['launches'] for v in j['kernels'].values()), {k: (round(v['ms'],1), v['launches']) for k,v in j['kernels'].items() if v['ms'] > 1.0})
" >> gpurun_out/small_ensembles.log
done
echo "small ensembles done" >> gpurun_out/status.log
# the other configurations of BASELINE.json on one GPU (C2: E = 1 is in the small-ensemble list above)
: > gpurun_out/configs.log
for cfg in "--scheduler lcm --denoise 4 --ensemble 1" "--kind normals --ensemble 4" "--ensemble 8" "--kind iid --ensemble 1 --denoise 4"; do
  timeout 300 python bench.py $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$cfg |', j['metric'], '|', j['value'], j['unit'], j['ms_per_step'], 'ms', 'one_at_a_time', j.get('latency_ms_per_map'))
" >> gpurun_out/configs.log
done
echo "configs done" >> gpurun_out/status.log
rm -rf gpurun_out/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r6 -- python $R/bench.py --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_bench.log 2>&1)
echo "rocprof rc=$?" >> gpurun_out/status.log
if [ "$1" == "pmc" ]; then
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o r6 -- python $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1)
  echo "pmc fetch rc=$?" >> gpurun_out/status.log
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o r6 -- python $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write.log 2>&1)
  echo "pmc write rc=$?" >> gpurun_out/status.log
fi
find gpurun_out -name "*kernel_trace*" -size +30M -delete 2>/dev/null
ls -la gpurun_out/prof/* gpurun_out/pmc_fetch/* gpurun_out/pmc_write/* 2>/dev/null | head -20 >> gpurun_out/status.log
cat gpurun_out/status.log; cat gpurun_out/small_ensembles.log; cat gpurun_out/configs.log
python -c "
import json
for n in ('bench_full','bench_nccl_world1'):
    j=json.loads([l for l in open(f'gpurun_out/{n}.json') if l.startswith('{')][-1])
    print(n, j['value'], j['ms_per_step'], j.get('roofline'), j.get('collective'), (j.get('calibration') or {}).get('gemm4096_bf16_tflops'))"
