#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_sk_$tag -o sk -- python $R/tools/shortk_pmc.py > $R/gpurun_out/pmc_sk_$tag.log 2>&1)
  echo "$tag rc=$?"
done
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_sk_*/sk_counter_collection.csv")):
    rows=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "igemm2" not in r["Kernel_Name"]: continue
        k=(r["Dispatch_Id"], r["Kernel_Name"].split("igemm2_kernel")[1][:48])
        rows.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
    for k,v in rows.items(): print(k[0],k[1]," ".join(f"{a}={b:.4g}" for a,b in v.items()))
PY
