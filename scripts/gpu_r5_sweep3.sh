#!/bin/bash
# Round-5 session: (a) the small ensembles re-swept with deeper split-K (12 / 16 / 24 K ranges: the 12^2 / 24^2 levels of 1-3
# members are weight streaming on too few workgroups), (b) the conv2 + conv_shortcut launches at every ensemble size; merged into
# the tuning table; benches before / after on the same box.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bench() { # name
  for e in 1 2 3 5 8 10; do
    MARIGOLD_TUNING=1 timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('igemm_mfma',)})
"
  done
}
bench before > gpurun_out/sweep3_bench.log
timeout 900 python tools/sweep_program.py --ensembles 1,2,3 --variants 23,35,22,32,36,46,51,62,72,73 --splits 1,2,3,4,6,8,12,16,24 --rounds 3 --iters 8 --emit-db gpurun_out/db_small.json > gpurun_out/sweep3_small.log 2>&1
echo "sweep small rc=$?" >> gpurun_out/sweep3_bench.log
timeout 600 python tools/sweep_program.py --ensembles 4,5,6,7,8,10 --only shortcut --variants 23,35,22,32,36,46,51,62,72,73 --splits 1,2,3,4,6,8 --rounds 3 --iters 8 --emit-db gpurun_out/db_fold.json > gpurun_out/sweep3_fold.log 2>&1
echo "sweep fold rc=$?" >> gpurun_out/sweep3_bench.log
python - <<'PY' >> gpurun_out/sweep3_bench.log
import json
p = "marigold_amd/tuning/gfx950.json"
db = json.load(open(p))
n0 = len(db["igemm"])
for f in ("gpurun_out/db_small.json", "gpurun_out/db_fold.json"):
    try:
        new = json.load(open(f))["igemm"]
    except Exception as e:
        print(f, "missing", e); continue
    db["igemm"].update(new)
    print(f, len(new), "entries")
json.dump(db, open(p, "w"), indent=0)
json.dump(db, open("gpurun_out/gfx950_merged.json", "w"), indent=0)
print("table", n0, "->", len(db["igemm"]))
PY
bench after >> gpurun_out/sweep3_bench.log
grep "per UNet forward\|table entries" gpurun_out/sweep3_small.log gpurun_out/sweep3_fold.log >> gpurun_out/sweep3_bench.log
cat gpurun_out/sweep3_bench.log
