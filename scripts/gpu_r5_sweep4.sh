#!/bin/bash
# Round-5 session: the deeper-ring tiles (24 / 25 / 26) - parity, then the small ensembles re-swept with them; merged into the
# tuning table; benches before / after on the same box.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/sweep4_bench.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --timeout-method=thread -k "igemm or linear or pingpong" 2>&1 | tail -5 >> gpurun_out/sweep4_bench.log
bench() { # name
  for e in 1 2 3 5; do
    MARIGOLD_TUNING=1 timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('igemm_mfma','time_embedding')})
"
  done
}
bench before >> gpurun_out/sweep4_bench.log
timeout 900 python tools/sweep_program.py --ensembles 1,2,3,5 --variants 23,24,25,26,35,22,32,36,46,73 --splits 1,2,3,4,6,8,12,16 --rounds 3 --iters 8 --emit-db gpurun_out/db_ring.json > gpurun_out/sweep4.log 2>&1
echo "sweep rc=$?" >> gpurun_out/sweep4_bench.log
python - <<'PY' >> gpurun_out/sweep4_bench.log
import json
p = "marigold_amd/tuning/gfx950.json"
db = json.load(open(p))
n0 = len(db["igemm"])
new = json.load(open("gpurun_out/db_ring.json"))["igemm"]
db["igemm"].update(new)
json.dump(db, open(p, "w"), indent=0)
json.dump(db, open("gpurun_out/gfx950_merged.json", "w"), indent=0)
print("table", n0, "->", len(db["igemm"]), "new", len(new), "on rings", sum(1 for v in new.values() if v[0] in (24, 25, 26)))
PY
bench after >> gpurun_out/sweep4_bench.log
grep "per UNet forward\|table entries" gpurun_out/sweep4.log >> gpurun_out/sweep4_bench.log
cat gpurun_out/sweep4_bench.log
