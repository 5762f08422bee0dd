"""Measured launch choices for MG_OP_IGEMM ("find-db"): for a layer shape the library's heuristic (csrc/igemm2.hip::
mg_igemm_auto_variant + the automatic split-K rule) is not the fastest tile on, ``gfx950.json`` names the tile variant and
split-K count that measured fastest on an MI355X - written by ``tools/sweep_program.py --emit-db`` from timings of the REAL
launches of the denoising / VAE programs (real buffers and epilogues) at the ensemble sizes one GPU sees (E = 10 on one GPU;
5 / 3 / 2 / 1 members per GPU when a map's ten members are sharded over 2 / 4 / 8 GPUs), committed with its sweep logs under
``profiles/``.  A shape absent from the table runs the heuristic; every variant named here is covered by the parity tests
(tests/test_gpu_kernels.py lists them; the folded launches' tile / split pairs in FOLD_CASES).  The table is gfx950's: the library
itself refuses any other device (mg_init), so no second gate is needed here.  The choice is a pure function of the op - deterministic, the same on every rank.

Key = (M, N, K, taps, stride, epilogue, transposed section?, batch_z, residual?, row statistics out?, folded LayerNorm in?,
second source?, time-embedding row?): what the tile's time depends on; B / H / W enter through M only.
"""
import json
import os

from .. import _lib as L

_DB = None
_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gfx950.json")
ENABLED = not (os.environ.get("MARIGOLD_TUNING") == "1" and os.environ.get("MARIGOLD_TUNING_DB") == "0")   # A/B: heuristics only


def key_of(op):
    i = op.i
    M, K = i[0] * i[4] * i[5], i[7] * i[3]
    key = (f"{M},{i[6]},{K},{i[7]},{i[8]},{i[12]},{int(i[14] >= 0)},{max(1, i[15])},"
           f"{int(bool(op.p[5]))},{int(bool(op.p[8]))},{int(bool(op.p[9]))},{int(bool(op.p[7]))},{int(bool(op.p[4]))}")
    key += f",x{i[32]}" if op.p[12] else ""   # (a folded 1x1 convolution: its channel count)
    # (round 6) what else decides whether a tile is legal or fast - appended only where it differs from the plain form the table
    # was swept on, so that existing entries keep their keys and an odd launch can never collide with them: a virtual up-sampled
    # input, a padding other than the tap window's own, operand row strides wider than the channel count
    if i[10] or i[11]:
        key += f",u{i[10]}x{i[11]}"
    if i[9] != (1 if i[7] in (9, 4) else 0):
        key += f",p{i[9]}"
    c0 = i[24] if op.p[7] else i[3]
    if (i[17] and i[17] != c0) or (op.p[7] and i[25] and i[25] != i[3] - c0) or (i[20] and i[20] != K + (i[32] if op.p[12] else 0)):
        key += f",ld{i[17]}.{i[25]}.{i[20]}"
    return key


def _fits_31bit(op):
    """The hand-placed tiles (72 / 73) address their operands with 31-bit byte offsets (csrc/igemm2.hip::dispatch_tile); the
    library's own choice falls back to 62 / 46 beyond that - a table entry must not take that fallback away."""
    i = op.i
    c0 = i[24] if op.p[7] else i[3]
    lda = i[17] or c0
    lda1 = (i[25] or i[3] - c0) if op.p[7] else 0
    ldx0 = (i[34] or (i[33] if op.p[13] else i[32])) if op.p[12] else 0
    ldx1 = (i[35] or i[32] - i[33]) if op.p[13] else 0
    ldw = i[20] or i[7] * i[3] + (i[32] if op.p[12] else 0)
    return i[0] * i[1] * i[2] * max(lda, lda1, ldx0, ldx1) < (1 << 30) and i[6] * ldw < (1 << 30)


def load():
    global _DB
    if _DB is None:
        try:
            with open(_PATH) as f:
                _DB = json.load(f)["igemm"]
        except FileNotFoundError:
            _DB = {}
    return _DB


def apply(op):
    """Set the measured (tile variant, split-K count) on an MG_OP_IGEMM op that leaves both to the library (i[19] == 0 and
    i[31] == 0).  Returns the op."""
    if not ENABLED or op.kind != L.OP_IGEMM or op.i[19] != 0 or op.i[31] != 0:
        return op
    hit = load().get(key_of(op))
    if hit is not None and not (int(hit[0]) in (72, 73) and not _fits_31bit(op)):
        op.i[19], op.i[31] = int(hit[0]), int(hit[1])
    return op
