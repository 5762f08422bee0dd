#!/bin/bash
# SQ wave-state counters per kernel over one map (bench.py --steps 1): where the wave cycles of every kernel class go
# (parked at waitcnt / barrier, issue-stalled, issuing), instruction counts, MFMA busy cycles.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_sq_$tag
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_sq_$tag -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_sq_$tag.log 2>&1)
  echo "$tag rc=$?"
done
python - <<'PY'
import csv,glob,collections,re,json
agg=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_sq_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
        d=agg.setdefault(k,collections.defaultdict(float))
        d[r["Counter_Name"]]+=float(r["Counter_Value"])
        d["_n_"+r["Counter_Name"]]+=1
rows=[]
for k,d in agg.items():
    wc=4*d.get("SQ_WAVE_CYCLES",0)
    if wc<=0: continue
    rows.append((wc,k,d))
rows.sort(reverse=True)
out={}
print(f"{'kernel':70s} {'launches':>8s} {'wave Gcyc':>9s} {'parked':>7s} {'stall':>7s} {'issue':>7s} | {'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'SALU/MFMA':>9s} {'mfma busy/wave cyc':>10s}")
for wc,k,d in rows[:28]:
    n=d.get("_n_SQ_WAVE_CYCLES",0)
    mf=d.get("SQ_INSTS_MFMA",0)
    row={"launches":n,"wave_cycles":wc,"parked":4*d["SQ_WAIT_ANY"]/wc,"issue_stall":4*d["SQ_WAIT_INST_ANY"]/wc,"issuing":4*d["SQ_ACTIVE_INST_ANY"]/wc,
         "valu_per_mfma":(d.get("SQ_INSTS_VALU",0)-mf)/mf if mf else None,"lds_per_mfma":d.get("SQ_INSTS_LDS",0)/mf if mf else None,
         "salu_per_mfma":d.get("SQ_INSTS_SALU",0)/mf if mf else None,"mfma_busy_per_wave_cycle":d.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/wc if mf else None,
         "waves":d.get("SQ_WAVES",0)}
    out[k]=row
    f=lambda x: "   -   " if x is None else f"{x:7.2f}"
    print(f"{k[:70]:70s} {n:8.0f} {wc/1e9:9.2f} {row['parked']:7.2f} {row['issue_stall']:7.2f} {row['issuing']:7.2f} | {f(row['valu_per_mfma'])} {f(row['lds_per_mfma'])} {f(row['salu_per_mfma'])} {f(row['mfma_busy_per_wave_cycle'])}")
json.dump(out,open("gpurun_out/r3_pmc_sq_summary.json","w"),indent=1)
PY
