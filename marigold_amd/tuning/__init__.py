"""Measured launch choices for MG_OP_IGEMM ("find-db"): for a layer shape the library's heuristic (csrc/igemm2.hip::
mg_igemm_auto_variant + the automatic split-K rule) is not the fastest tile on, ``gfx950.json`` names the tile variant and
split-K count that measured fastest on an MI355X - written by ``tools/sweep_program.py --emit-db`` from timings of the REAL
launches of the denoising / VAE programs (real buffers and epilogues) at the ensemble sizes one GPU sees (E = 10 on one GPU;
5 / 3 / 2 / 1 members per GPU when a map's ten members are sharded over 2 / 4 / 8 GPUs), committed with its sweep logs under
``profiles/``.  A shape absent from the table runs the heuristic; every variant named here is covered by the parity tests
(tests/test_gpu_kernels.py lists them).  The choice is a pure function of the op - deterministic, the same on every rank.

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
    return key + (f",x{i[32]}" if op.p[12] else "")   # (a folded 1x1 convolution: its channel count)


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
    if hit is not None:
        op.i[19], op.i[31] = int(hit[0]), int(hit[1])
    return op
