#!/bin/bash
# Tuning-table session: bench the current table, sweep the real launches of the programs at the ensemble sizes of ENS
# (tools/sweep_program.py; extra arguments pass through, e.g. --only shortcut --splits 1,2,4,8), merge the winners into
# marigold_amd/tuning/gfx950.json ON THE BOX, bench again; the merged table comes back as gpurun_out/gfx950_merged.json.
#   gpurun -- 'ENS=1,2,3,5 bash scripts/gpu_sweep.sh --variants 23,24,25,26,35,22,32,36,46,73 --splits 1,2,3,4,6,8,12,16'
# A sweep times a launch in isolation: vote its entries inside the program before committing them (scripts/gpu_ab_tables.sh,
# tools/merge_tuning_tables.py).
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ENS=${ENS:-1,2,3,5}
bench() { # name
  for e in ${ENS//,/ }; do
    MARIGOLD_TUNING=1 timeout 300 python bench.py --ensemble $e --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 E=$e', 'ms', j['ms_per_step'], {k: (round(v['ms'],1), v['launches']) for k,v in j.get('kernels',{}).items() if k in ('igemm_mfma',)})
"
  done
}
bench before > gpurun_out/sweep_bench.log
timeout 1200 python tools/sweep_program.py --ensembles $ENS --rounds 3 --iters 8 --emit-db gpurun_out/db_new.json "$@" > gpurun_out/sweep.log 2>&1
echo "sweep rc=$?" >> gpurun_out/sweep_bench.log
python - <<'PY' >> gpurun_out/sweep_bench.log
import json
p = "marigold_amd/tuning/gfx950.json"
db = json.load(open(p))
n0 = len(db["igemm"])
new = json.load(open("gpurun_out/db_new.json"))["igemm"]
db["igemm"].update(new)
json.dump(db, open(p, "w"), indent=0)
json.dump(db, open("gpurun_out/gfx950_merged.json", "w"), indent=0)
print("table", n0, "->", len(db["igemm"]), "new", len(new))
PY
bench after >> gpurun_out/sweep_bench.log
grep "per UNet forward\|table entries" gpurun_out/sweep.log >> gpurun_out/sweep_bench.log
cat gpurun_out/sweep_bench.log
