#!/usr/bin/env python
"""Aggregates the rocprofv3 --pmc passes of tools/microbench.py (gpurun_out/pmc/set*/..., made by
scripts/gpu_pmc_only.sh) per kernel into profiles/r1_pmc_mfma_util_microbench.json: MFMA busy fraction of the
SIMD time, VALU / LDS / SALU instructions per MFMA, LDS bank-conflict fraction.  Counter units follow
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES quad-cycles."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "set*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
                if not k.startswith(("igemm2_", "flash_attn64")):
                    continue
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
    # wall time of the same launches, from the kernel trace of the pass that collected the MFMA-busy counter
    dur = collections.defaultdict(float)
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "set1", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
                dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    out = {}
    for k, c in sorted(sums.items()):
        d = {}
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and dur.get(k):
            # busy cycles summed over the 1024 SIMDs / (SIMDs x wall time x nominal 2.4 GHz): the fraction of the
            # nominal-clock MFMA roof (the chip runs 1.7-2.1 GHz under this load, so the pipe itself is busier)
            d["mfma_busy_frac_of_2p4GHz_roof"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * dur[k] * 2.4e9)
            d["seconds_profiled"] = dur[k]
        if c.get("SQ_INSTS_MFMA"):
            for name in ("VALU", "LDS", "SALU"):
                if c.get("SQ_INSTS_" + name):
                    d[f"insts_{name.lower()}_per_mfma"] = c["SQ_INSTS_" + name] / c["SQ_INSTS_MFMA"]
            if c.get("SQ_INSTS_VALU"):
                d["insts_valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVE_CYCLES"):
            d["wave_cycles_per_wave"] = 4.0 * c["SQ_WAVE_CYCLES"] / max(c.get("SQ_WAVES", 0.0), 1.0)
        d["raw"] = dict(c)
        out[k] = d
    dst = os.path.join(ROOT, "profiles", "r1_pmc_mfma_util_microbench_final.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "raw"} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
