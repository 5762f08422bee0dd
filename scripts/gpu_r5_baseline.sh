#!/bin/bash
# Round-5 baseline session: the whole GPU suite (-s: the [parity] lines), smoke, the headline bench with the per-op table, the same
# bench under torchrun with the RCCL path forced on one rank, and the small ensembles (the per-GPU shards of 2 / 4 / 8 GPUs).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/status.log
timeout 900 python -m pytest tests -m gpu -q -s --timeout=400 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/status.log
tail -3 gpurun_out/t_all.log >> gpurun_out/status.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/status.log
timeout 600 python bench.py --steps 6 --warmup 2 --dump-ops gpurun_out/ops_e10.tsv > gpurun_out/bench_e10.json 2> gpurun_out/bench_e10.log
echo "bench rc=$?" >> gpurun_out/status.log
MARIGOLD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_e10_nccl1.json 2> gpurun_out/bench_e10_nccl1.log
echo "bench nccl rc=$?" >> gpurun_out/status.log
for e in 1 2 3 5; do
  timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_e$e.tsv > gpurun_out/bench_e$e.json 2> gpurun_out/bench_e$e.log
  echo "bench e$e rc=$?" >> gpurun_out/status.log
done
python - <<'EOF' >> gpurun_out/status.log
import json
for n in ("e10", "e10_nccl1", "e1", "e2", "e3", "e5"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/bench_{n}.json") if l.startswith("{")][-1])
        print(n, j["ms_per_step"], "ms", {k: round(v["ms"], 1) for k, v in j.get("stages", {}).items()},
              {k: (round(v["ms"], 1), v["launches"]) for k, v in j.get("kernels", {}).items()}, (j.get("calibration") or {}).get("gemm4096_bf16_tflops"), j.get("collective"))
    except Exception as e:
        print(n, "failed", e)
EOF
cat gpurun_out/status.log
