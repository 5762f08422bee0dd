#!/usr/bin/env python
"""Turns the two rocprofv3 --pmc passes of bench.py (gpurun_out/pmc_fetch, gpurun_out/pmc_write; made by
`scripts/gpu_round.sh pmc`) into profiles/<round>_pmc_hbm_traffic.{csv,json}: HBM bytes per launch per kernel,
FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads - MI355X_MICROARCH.md, HBM).
Usage: python tools/pmc_traffic.py [round tag, default r6]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != name:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            a = agg[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
    base = os.path.join(ROOT, "gpurun_out")
    import glob

    def one(d):
        hits = glob.glob(os.path.join(base, d, "**", "*counter_collection.csv"), recursive=True)
        assert hits, f"no counter_collection.csv under gpurun_out/{d}"
        return max(hits, key=os.path.getmtime)   # (gpurun_out/ keeps the files of earlier rounds)
    f = load(one("pmc_fetch"), "FETCH_SIZE")
    w = load(one("pmc_write"), "WRITE_SIZE")
    classes = {"igemm_mfma": "igemm2_", "rowgemm_mfma": "rowgemm_", "conv3x3_patch": "conv_patch", "flash_attn64": "flash_attn64",
               "groupnorm": "gn_"}
    out = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.csv"), "w") as c:
        c.write("kernel,launches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,hbm_MB_per_launch_corrected\n")
        for k in sorted(f, key=lambda k: -f[k][1]):
            n, fs = f[k]
            ws = w.get(k, [0, 0.0])[1]
            c.write(f"\"{k}\",{n},{fs:.0f},{ws:.0f},{(2 * fs + ws) * 1024 / n / 1e6:.3f}\n")
    for cls, prefix in classes.items():
        n = sum(v[0] for k, v in f.items() if k.startswith(prefix))
        fs = sum(v[1] for k, v in f.items() if k.startswith(prefix))
        ws = sum(v[1] for k, v in w.items() if k.startswith(prefix))
        if n:
            out[cls] = {"launches": n, "bytes_per_launch": (2 * fs + ws) * 1024 / n,
                        "fetch_kb_sum": fs, "write_kb_sum": ws}
    # which build the counters belong to: the content hash bench.py compares against (the GPU box has no .git) and, for the reader,
    # the commit this tree sits on.  Run this tool on the tree that was sent to the GPU box, before editing it further.
    sys.path.insert(0, ROOT)
    from marigold_amd.util.host import build_fingerprint
    import subprocess
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "marigold_amd"], capture_output=True, text=True).stdout.strip())
    except OSError:
        head, dirty = "", False
    out["_build"] = {"fingerprint": build_fingerprint(), "git_head": head, "git_dirty": dirty}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json"), "w") as j:
        json.dump(out, j, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
