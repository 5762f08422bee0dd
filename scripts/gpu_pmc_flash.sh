#!/bin/bash
# PMC passes over the flash-attention variants (counters only: no trace domains besides the kernel names the CSV carries)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_fa_$tag
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_fa_$tag -o fa -- python $R/tools/flash_pmc.py > $R/gpurun_out/pmc_fa_$tag.log 2>&1)
  echo "$tag rc=$?"; tail -2 $R/gpurun_out/pmc_fa_$tag.log
done
python - <<'PY'
import csv,glob,collections,re
agg=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_fa_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "flash" not in r["Kernel_Name"]: continue
        k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
        d=agg.setdefault(k,collections.defaultdict(list))
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in d.items(): print(f"   {c:28s} {sum(v)/len(v):14.4g}  (n={len(v)})")
PY
