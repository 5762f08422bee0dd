#!/bin/bash
# round 4, session e: per-op effect of the hand-placed K loop in the pipeline (op tables for MARIGOLD_K4W=0 / 1)
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in 0 1; do
  MARIGOLD_K4W=$k MARIGOLD_DEEP_TILE=$([ $k = 1 ] && echo 72 || echo 62) timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r4e_ops_k4w$k.tsv 2>/dev/null | tail -1 | cut -c1-200
done
python - <<'PY'
import csv, collections, re
def load(p):
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(p), delimiter='\t'):
        if r['class']!='igemm_mfma': continue
        lab=re.sub(r'\.(\d+)\.', '.#.', r['label'])
        k=(r['stage'],lab,round(float(r['GFLOP']),1))
        a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['ms'])
    return agg
a=load('gpurun_out/r4e_ops_k4w0.tsv'); b=load('gpurun_out/r4e_ops_k4w1.tsv')
for k in a:
    if k in b and abs(a[k][1]-b[k][1])>0.03*a[k][1] and a[k][1]>0.3:
        print(f"{k[1]:55s} GF={k[2]:8.1f} n={a[k][0]:3d}  K4W=0 {a[k][1]:6.2f} ms ({k[2]*a[k][0]/a[k][1]:5.0f} TF/s)   K4W=1 {b[k][1]:6.2f} ms ({k[2]*b[k][0]/b[k][1]:5.0f} TF/s)")
PY
