#!/usr/bin/env python
"""In-program vote between two tuning tables (runs without a GPU; the timings come from the MI355X): _ab/gfx950_A.json (the
committed table) and _ab/gfx950_B.json (A + a new sweep's entries) were both benched with ``bench.py --dump-ops`` at E = 1, 2, 3, 5
(scripts/gpu_ab_tables.sh -> gpurun_out/ops_{A,B}{1,2}_e<E>.tsv).  A sweep times a launch in isolation (same launch back to
back, warm L2); inside the program some of its winners lose.  An entry that differs between the tables is taken from B only if
the layers it applies to ran >= 2 % faster in B's programs; everything else stays as in A.  Writes _ab/gfx950_C.json.
"""
import os
import sys, json, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import marigold_amd as M
from marigold_amd import _lib as L, tuning
from marigold_amd.schedulers import DDIMScheduler
A = json.load(open('_ab/gfx950_A.json')); B = json.load(open('_ab/gfx950_B.json'))
a, b = A['igemm'], B['igemm']
changed = {k for k in set(a) | set(b) if a.get(k, [0, 0])[:2] != b.get(k, [0, 0])[:2]}
print(len(a), len(b), 'changed', len(changed))
tuning.ENABLED = False
pipe = M.build_synthetic_pipeline("depth", default_processing_resolution=0)
pipe.unet.dry(); pipe.vae.dry()
pipe.unet.set_context(pipe.empty_text_embed)
def load(f):
    return [(r['stage'], r['label'], float(r['ms'])) for r in csv.DictReader(open(f), delimiter='\t')]
votes = collections.defaultdict(lambda: [0.0, 0.0, 0])   # key -> [tA, tB, n]
for E in (1, 2, 3, 5):
    prog = pipe.unet.denoise_program(E, 96, 96, DDIMScheduler(), 10, rgb_broadcast=True)
    lab2key = {}
    for op, lab in zip(prog.seq.ops, prog.seq.labels):
        if op.kind == L.OP_IGEMM:
            lab2key[('denoise', lab)] = tuning.key_of(op)
    for kind, st in (("encode", "vae.encode"), ("decode", "vae.decode")):
        seq, _, _ = pipe.vae._program(kind, E if kind == "decode" else 1, 96 if kind == "decode" else 768, 96 if kind == "decode" else 768, L.POST_DEPTH if kind == "decode" else 0)
        for op, lab in zip(seq.ops, seq.labels):
            if op.kind == L.OP_IGEMM:
                lab2key[(st, lab)] = tuning.key_of(op)
    tabs = {t: [load(f'gpurun_out/ops_{t}{r}_e{E}.tsv') for r in (1, 2)] for t in 'AB'}
    n = len(tabs['A'][0])
    for i in range(n):
        st, lab, _ = tabs['A'][0][i]
        k = lab2key.get((st, lab))
        if k is None or k not in changed:
            continue
        ta = (tabs['A'][0][i][2] + tabs['A'][1][i][2]) / 2
        tb = (tabs['B'][0][i][2] + tabs['B'][1][i][2]) / 2
        v = votes[k]; v[0] += ta; v[1] += tb; v[2] += 1
    pipe.unet._programs.clear(); pipe.vae._programs.clear()
c = dict(a)
took = kept = unseen = 0
gain = 0.0
for k in sorted(changed):
    if k not in votes:
        unseen += 1      # a shape of another ensemble size (not in these programs): keep A's choice
        continue
    ta, tb, n = votes[k]
    if tb < 0.98 * ta:
        if k in b: c[k] = b[k]
        else: c.pop(k, None)
        took += 1; gain += ta - tb
    else:
        kept += 1
print('took', took, 'kept', kept, 'unseen', unseen, 'in-program gain (sum over E=1,2,3,5 maps) ms', round(gain, 2))
A['igemm'] = c
json.dump(A, open('_ab/gfx950_C.json', 'w'), indent=0)
print('table C', len(c), 'entries; on rings', sum(1 for v in c.values() if v[0] in (24, 25, 26)))
