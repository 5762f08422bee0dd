#!/bin/bash
# round 4, session d: (1) whole-map A/B of the hand-placed K loop in the automatic tile choice (MARIGOLD_K4W=0/1);
# (2) more schedule variants on the 4096^3 GEMM; (3) SQ counters of the 4096^3 GEMM: variant 72 against hipBLASLt
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
mkdir -p gpurun_out
for round in 1 2; do
  for k in 0 1; do
    MARIGOLD_K4W=$k MARIGOLD_DEEP_TILE=$([ $k = 1 ] && echo 72 || echo 62) timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('K4W=$k', 'ms', j['ms_per_step'], ' '.join(f\"{n}={v['ms']:.1f}\" for n,v in k.items() if v['ms']>1.5), 'ens', j['stages'].get('ensemble',{}).get('ms'), j.get('calibration',{}).get('shader_mhz_under_mfma_load'))
"
  done
done 2>&1 | tee gpurun_out/r4d_ab_k4w.log
for lib in "" _k4w1 _k4w2; do
  echo "== lib$lib"
  MARIGOLD_HIP_LIB=$PWD/marigold_amd/libmarigold_hip$lib.so GEMM_VARIANTS=62,72 GEMM_SIZES=4096 GEMM_ROUNDS=5 timeout 200 python tools/gemm_bench.py 2>&1 | grep "^gemm"
done 2>&1 | tee gpurun_out/r4d_sched.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_gemm_$tag
  (cd /tmp && export TMPDIR=/tmp && GEMM_VARIANTS=62,72 GEMM_SIZES=4096 GEMM_ROUNDS=1 timeout 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_gemm_$tag -o sq -- python $R/tools/gemm_bench.py > $R/gpurun_out/pmc_gemm_$tag.log 2>&1)
  echo "$tag rc=$?"
done
python - <<'PY'
import csv,glob,collections,re,json
agg=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_gemm_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0][:60]
        d=agg.setdefault(k,collections.defaultdict(float))
        d[r["Counter_Name"]]+=float(r["Counter_Value"]); d["_n_"+r["Counter_Name"]]+=1
for k,d in agg.items():
    if not d.get("SQ_INSTS_MFMA"): continue
    n=d["_n_SQ_WAVE_CYCLES"]; wc=4*d["SQ_WAVE_CYCLES"]
    print(k, "launches", n)
    print("   wave cycles/launch %.3g  parked %.2f stall %.2f issuing %.2f  wait_inst_lds %.3f" % (wc/n, 4*d["SQ_WAIT_ANY"]/wc, 4*d["SQ_WAIT_INST_ANY"]/wc/ (2 if d["_n_SQ_WAIT_INST_ANY"]>n else 1), 4*d["SQ_ACTIVE_INST_ANY"]/wc, 4*d["SQ_WAIT_INST_LDS"]/wc))
    mf=d["SQ_INSTS_MFMA"]
    print("   per MFMA: VALU %.2f LDS %.2f SALU %.2f VMEM_RD %.3f ; mfma busy cycles / wave cycle %.3f" % ((d["SQ_INSTS_VALU"]-mf)/mf, d["SQ_INSTS_LDS"]/mf, d["SQ_INSTS_SALU"]/mf, d["SQ_INSTS_VMEM_RD"]/mf, d["SQ_VALU_MFMA_BUSY_CYCLES"]/wc))
    print("   LDS bank conflict / idx active %.3f (%.3g / %.3g)  addr conflict %.3g" % (d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1), d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"], d.get("SQ_LDS_ADDR_CONFLICT",0)))
PY
