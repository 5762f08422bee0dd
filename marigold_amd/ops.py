"""Op descriptors for libmarigold_hip (one ``mg_op`` per kernel launch) and ``OpSeq``, the
host-side container that turns a list of them into a native program (``mg_program_*``).

torch is used only as the owner of device memory and the source of the HIP stream handle.
Field layout of every op is documented in include/marigold_hip.h.
"""
import ctypes

import torch

from . import _lib as L
from ._lib import MgOp


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return int(x)


def make_op(kind, i=(), f=(), p=(), l=()):
    op = MgOp()
    op.kind = kind
    for k, v in enumerate(i):
        op.i[k] = int(v)
    for k, v in enumerate(f):
        op.f[k] = float(v)
    for k, v in enumerate(p):
        op.p[k] = _ptr(v)
    for k, v in enumerate(l):
        op.l[k] = int(v)
    return op


def current_stream_handle():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------- builders

def igemm(a, w, out, *, B, H, W, Cin, Ho, Wo, N, taps=1, stride=1, pad=0, up=None, bias=None,
          rowvec=None, residual=None, epi=L.EPI_BF16, ldo=None, out2=None, trans_from=-1, ldt=0,
          batch_z=1, ldr=0, lda=0, ldw=0, zstrides=(0, 0, 0, 0), scale=1.0, variant=0,
          rowvec_bcast=False, n_alg=0, k_alg=0, a1=None, C0=0, lda1=0, ln_out=None, ln_in=None, ln_g=None, ln_c=None,
          ln_eps=1e-5, sm_scale=0.0, sm_cols=0, c2=0, trans_perm=False, ln_counters=None, splits=0, fold=None):
    """``fold`` = (x0, x1 | None, Cx0, Cx[, ldx0, ldx1]): a 1x1 convolution of a second tensor (the ResNet block's conv_shortcut)
    as extra K of this 3x3 convolution; ``w`` rows are then [conv weights | shortcut weights]."""
    hu, wu = up if up else (0, 0)
    cp = _ptr(ln_counters) or 0   # the program's own row-block tickets (two int32 halves of the device address)
    c_lo, c_hi = cp & 0xffffffff, (cp >> 32) & 0xffffffff
    c_lo, c_hi = (c_lo - (1 << 32) if c_lo >= 1 << 31 else c_lo), (c_hi - (1 << 32) if c_hi >= 1 << 31 else c_hi)
    if ldo is None:
        ldo = N // 2 if epi == L.EPI_GEGLU else N
    return make_op(L.OP_IGEMM,
                   i=[B, H, W, Cin, Ho, Wo, N, taps, stride, pad, hu, wu, epi, ldo, trans_from,
                      batch_z, ldr, lda, ldt, variant, ldw, int(rowvec_bcast), n_alg, k_alg, C0, lda1,
                      int(trans_perm), sm_cols, c2, c_lo, c_hi, splits] + ([fold[3], fold[2], fold[4] if len(fold) > 4 else 0, fold[5] if len(fold) > 5 else 0] if fold else []),
                   f=[scale, ln_eps, sm_scale], p=[a, w, out, bias, rowvec, residual, out2, a1, ln_out, ln_in, ln_g, ln_c] + ([fold[0], fold[1]] if fold else []),
                   l=list(zstrides))


def conv3x3(a0, w, out, *, B, H, W, C0, N, a1=None, C1=0, subpix=False, ss=None, silu=False, bias=None, rowvec=None,
            residual=None, lda0=0, lda1=0, ldo=0, ldr=0, ldw=0, rowvec_bcast=False, variant=0, wz=0, gn_part=None, gn_cpg=0,
            gn_slots=0):
    """Patch-resident conv3x3 / pad 1 (MG_OP_CONV3X3): fused GroupNorm scale/shift (+SiLU) on the input, second
    channel source, sub-pixel 2x up-sampling; ``gn_part``: the output's GroupNorm partial sums as a by-product
    (``conv3x3_gn_slots`` tells whether / how many slots per image)."""
    return make_op(L.OP_CONV3X3,
                   i=[B, H, W, C0, C1, N, int(subpix), int(silu), lda0, lda1, ldo, ldr, ldw, int(rowvec_bcast), variant, gn_cpg, gn_slots],
                   p=[a0, w, out, bias, rowvec, residual, a1, ss, gn_part], l=[wz])


def conv3x3_gn_slots(op, f16=False):
    """Partial-table slots per image this MG_OP_CONV3X3 fills with its output's GroupNorm statistics (0: its tile does not);
    ``f16``: asked of the fp16-operand build (its tile choice differs where a kernel is bf16-only)."""
    return int(L.load(f16).mg_conv3x3_gn_slots(ctypes.byref(op)))


def rowgemm(x, wp, out, *, M, K, N, form=L.RG_BF16, ldx=0, ldo=0, ldr=0, residual=None, ln_in=None, ln_out=None, vt=None,
            gn_ss=None, tokens=0, ldt=0, trans_from=0, waves=0, ln_eps=1e-5, sm_cols=0, sm_scale=0.0, dbg=None, nsplit=0,
            xattn=None, xout=None):
    """Row-resident GEMM (MG_OP_ROWGEMM): ``wp`` from weights.pack_rowgemm (form RG_XATTN: pack_rowgemm_xattn).  ``xattn`` (GEGLU
    form, K = 320, no column split): a pack_rowgemm_xattn image - the collapsed cross-attention runs on the rows in registers before
    the projection (``ln_in`` = the statistics of the rows as loaded, ``xout`` = where the updated rows go, ``sm_cols`` / ``sm_scale``)."""
    return make_op(L.OP_ROWGEMM, i=[M, K, N, ldx, ldo, ldr, form, tokens, ldt, trans_from, waves, sm_cols, nsplit], f=[ln_eps, sm_scale],
                   p=[x, wp, out, residual, ln_in, ln_out, vt, gn_ss, dbg, xattn, xout if xattn is not None else None])


def linear(x, w, out, *, M, K, N, **kw):
    """out[M][N] = x[M][K] @ w[N][K]^T (+ fused epilogue)."""
    return igemm(x, w, out, B=1, H=M, W=1, Cin=K, Ho=M, Wo=1, N=N, taps=1, **kw)


def gn_stats(x, partials, *, B, HW, C, chunks, groups, Ctot=0, coff=0, slot0=0, slots=0, gamma=None, beta=None, ss=None,
             counters=None, eps=0.0, x1=None, C1=0):
    """Partials [B][slots][groups][2]; with ``ss`` the image's last-arriving block also finalizes (MG_OP_GN_STATS).  ``x1``
    ([B][HW][C1]): the concat's second source in the same launch (slots slot0 + chunks ...)."""
    return make_op(L.OP_GN_STATS, i=[B, HW, C, chunks, Ctot, coff, groups, slot0, slots, C1], f=[eps],
                   p=[x, partials, gamma, beta, ss, counters, x1])


def gn_finalize(partials, gamma, beta, ss, *, B, C, groups, slots, HW, eps):
    return make_op(L.OP_GN_FINALIZE, i=[B, C, groups, slots, HW], f=[eps], p=[partials, gamma, beta, ss])


def gn_apply(x, ss, out, *, B, HW, C, silu, x1=None, C0=0):
    return make_op(L.OP_GN_APPLY, i=[B, HW, C, int(silu), C0], p=[x, ss, out, x1])


def gn_slab(x0, out, ss, *, B, HW, C, groups, gamma, beta, eps, silu=False, x1=None, C0=0):
    """GroupNorm in one launch (MG_OP_GN_SLAB): scale / shift into ``ss`` and, with ``out``, the normalised tensor."""
    return make_op(L.OP_GN_SLAB, i=[B, HW, C, C0, groups, int(silu)], f=[eps], p=[x0, x1, out, gamma, beta, ss])


SPLITK_WS_BYTES = 64 << 20   # MG_SPLITK_WS_BYTES (csrc/common.h): what MG_OP_IGEMM p[14] must hold
FLASH_WS_BYTES = 4096 + 255 * 4 * 4 * (16384 + 1024)   # tickets + four partial results for up to 255 split blocks of queries (tests: split = 1)
# What the engine allocates per program: the automatic rule (split = 0) only splits a left-over of at most CUs / 8 blocks (32 on
# MI355X; 40 leaves room for a larger part) - 11 MB instead of 71 MB zeroed per Builder.  A smaller workspace than a launch could
# use is safe: the plan then does not split (flash4w.hip::mg_flash4w_plan).
FLASH_WS_BYTES_AUTO = 4096 + 40 * 4 * 4 * (16384 + 1024)


def flash_attn64(q, k, vt, o, *, B, heads, Ntok, ldq, ldo, ldvt, sq, sk, svt, so, scale, variant=0, vt_perm=False, dbg=None,
                 redo_thr=0.0, ws=None, ws_bytes=0, split=0):
    """``vt_perm``: V^T holds its keys in the order [0-3, 8-11, 4-7, 12-15] inside every group of 16 (what MG_OP_IGEMM's
    transposed section writes with ``trans_perm``) - generation 3 consumes that order without a lane exchange.
    ``redo_thr`` (tests only; 0 = 2^100): the row-sum bound above which the hand-placed kernel (variant 26) redoes a block of
    queries with the running-maximum loop.  ``ws`` (optional, ZEROED once, then owned by the launches of one stream): workspace of
    the hand-placed kernel's key-split blocks - the blocks of 256 queries beyond the last multiple of the CU count are split
    along the keys over the chip (``FLASH_WS_BYTES`` covers every case); ``split``: 0 = when it pays, 1 = always (tests), 2 = never."""
    return make_op(L.OP_FLASH_ATTN64, i=[B, heads, Ntok, ldq, ldo, ldvt, variant, int(vt_perm), ws_bytes // 1024, split], f=[scale, redo_thr],
                   p=[q, k, vt, o, dbg, ws], l=[sq, sk, svt, so])


def flash_attn512(q, k, vt, o, *, B, Ntok, ldq, ldo, ldvt, sq, sk, svt, so, scale):
    """One head of width 512 (the VAE mid-block attention), flash form: no score matrix in memory (MG_OP_FLASH_ATTN512)."""
    return make_op(L.OP_FLASH_ATTN512, i=[B, Ntok, ldq, ldo, ldvt], f=[scale], p=[q, k, vt, o], l=[sq, sk, svt, so])


VT_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def permute_vt_keys(vt):
    """Natural V^T [..., keys] (keys a multiple of 16) -> the ``vt_perm`` order (host helper for tests / tools)."""
    sh = vt.shape
    return vt.reshape(*sh[:-1], sh[-1] // 16, 16)[..., list(VT_PERM16)].reshape(sh).contiguous()


def softmax_rows(s, p, *, R, ncols, lds, ldp):
    return make_op(L.OP_SOFTMAX_ROWS, i=[R, ncols, lds, ldp], p=[s, p])


def sched_step(x, model_out, noise, out, *, n, cx, cm, cn=0.0):
    return make_op(L.OP_SCHED_STEP, f=[cx, cm, cn], p=[x, model_out, noise, out], l=[n])


def linear_small_m(x, w, b, out, *, M, N, K, act_in=0, act_out=0, ldo=0):
    return make_op(L.OP_LINEAR_SMALL_M, i=[M, N, K, act_in, act_out, ldo], p=[x, w, b, out])


def latent_1x1(x, w, b, out, *, B, Ci, Co, HW, scale=1.0):
    return make_op(L.OP_LATENT_1X1, i=[B, Ci, Co, HW], f=[scale], p=[x, w, b, out])


def im2col_small(src0, src1, out, *, B, H, W, C0, C1, Kp, bcast0=False):
    return make_op(L.OP_IM2COL_SMALL, i=[B, H, W, C0, C1, Kp, int(bcast0)], p=[src0, src1, out])


def conv3x3_head(x, ss, w, bias, out, *, B, H, W, C, Cout, ldo=0, silu=True):
    """GroupNorm apply (``ss`` = scale / shift [B][2][C], or None) [+ SiLU] + conv3x3 pad 1 to <= 4 fp32 channels, one launch
    (MG_OP_CONV3X3_HEAD); ``w`` bf16 [>= Cout][9 C] as weights.pack_conv3x3 lays it out."""
    return make_op(L.OP_CONV3X3_HEAD, i=[B, H, W, C, Cout, ldo, int(silu)], p=[x, ss, w, bias, out])


def post_nchw(x, out, *, B, HW, Cout, ldi, post=L.POST_NONE, scale=1.0, noise=None, cx=0.0, cm=0.0, cn=0.0):
    return make_op(L.OP_POST_NCHW, i=[B, HW, Cout, ldi, post], f=[scale, cx, cm, cn], p=[x, out, noise])


def ens_depth_stats(d, scratch, out, *, E, HW):
    return make_op(L.OP_ENS_DEPTH_STATS, i=[E], p=[d, scratch, out], l=[HW])


def ens_depth_median(d, st, med, mad, minmax, scratch, *, E, HW, reduction=0, has_shift=True):
    return make_op(L.OP_ENS_DEPTH_MEDIAN, i=[E, reduction, int(has_shift)],
                   p=[d, st, med, mad, minmax, scratch], l=[HW])


def ens_depth_norm(med, unc, minmax, *, HW, shift_invariant=True):
    return make_op(L.OP_ENS_DEPTH_NORM, i=[int(shift_invariant)], p=[med, unc, minmax], l=[HW])


def ens_normals(n, out, unc, *, E, HW, reduction=0):
    return make_op(L.OP_ENS_NORMALS, i=[E, reduction], p=[n, out, unc], l=[HW])


def resize(src, dst, tmp, *, planes, Hin, Win, Hout, Wout, mode, u8):
    return make_op(L.OP_RESIZE, i=[planes, Hin, Win, Hout, Wout, mode, int(u8)], p=[src, dst, tmp])


def colorize(depth, lut, out, *, n, lo=0.0, hi=1.0):
    return make_op(L.OP_COLORIZE, f=[lo, hi], p=[depth, lut, out], l=[n])


def memset(dst, nbytes, value=0):
    return make_op(L.OP_MEMSET, i=[value], p=[dst], l=[nbytes])


def copy(src, dst, nbytes):
    return make_op(L.OP_COPY, p=[src, dst], l=[nbytes])


# --------------------------------------------------------------------------- containers

def launch(op, stream=None, lib=None):
    """Launch one op on torch's current stream (or the given raw handle); ``lib``: the library build (default: bf16 operands)."""
    lib = lib or L.load()
    L.check(lib.mg_launch(ctypes.byref(op), stream if stream is not None else current_stream_handle()),
            f"mg_launch({L.OP_NAMES.get(op.kind, op.kind)})", lib)


class OpSeq:
    """An ordered list of ops + the tensors they reference (kept alive), compiled on demand to
    a native ``mg_program`` so that a whole UNet forward / denoising loop / VAE pass is ONE
    C call (and optionally one hipGraph launch)."""

    def __init__(self, name="", f16=False):
        self.name = name
        self.f16 = bool(f16)   # the library build this program belongs to: fp16 operands (libmarigold_hip_f16.so) or bf16
        self.ops = []
        self.labels = []
        self.keep = []
        self.zero_state = set()   # data_ptr()s of the held tensors that are zero-initialised kernel state (tickets, workspaces, pad columns)
        self._prog = None
        self._captured = False

    def add(self, op, label=""):
        self.ops.append(op)
        self.labels.append(label)
        self._prog = None
        return op

    def hold(self, *tensors):
        self.keep.extend(tensors)
        return tensors[0] if len(tensors) == 1 else tensors

    def extend(self, other):
        self.ops.extend(other.ops)
        self.labels.extend(other.labels)
        self.keep.extend(other.keep)
        self.zero_state |= other.zero_state
        self._prog = None

    def __len__(self):
        return len(self.ops)

    def compile(self):
        if self._prog is None:
            lib = L.load(self.f16)
            arr = (MgOp * len(self.ops))(*self.ops)
            prog = lib.mg_program_create(arr, len(self.ops))
            if not prog:
                L.check(1, "mg_program_create", lib)
            self._prog = prog
            self._captured = False
        return self._prog

    def run(self, stream=None):
        lib = L.load(self.f16)
        prog = self.compile()
        L.check(lib.mg_program_run(prog, stream if stream is not None else current_stream_handle()),
                f"mg_program_run({self.name})", lib)

    def run_range(self, first, count, stream=None):
        """Replay ops [first, first+count) only (step-wise inspection of a denoising program in the parity tests)."""
        lib = L.load(self.f16)
        L.check(lib.mg_program_run_range(self.compile(), int(first), int(count),
                                         stream if stream is not None else current_stream_handle()),
                f"mg_program_run_range({self.name})", lib)

    def validate(self):
        """Dry-run every op through its launcher's contract checks (works without a GPU)."""
        lib = L.load(self.f16)
        L.check(lib.mg_program_validate(self.compile()), f"mg_program_validate({self.name})", lib)

    def run_eager(self, stream=None):
        for op in self.ops:
            launch(op, stream, L.load(self.f16))

    def capture(self, stream=None):
        """Capture into a hipGraph (the stream must not be the legacy default stream)."""
        lib = L.load(self.f16)
        prog = self.compile()
        L.check(lib.mg_program_capture(prog, stream if stream is not None else current_stream_handle()),
                f"mg_program_capture({self.name})", lib)
        self._captured = True

    def profile(self, stream=None):
        """Per-op milliseconds (HIP events on the launch stream)."""
        lib = L.load(self.f16)
        prog = self.compile()
        ms = (ctypes.c_float * len(self.ops))()
        L.check(lib.mg_program_profile(prog, stream if stream is not None else current_stream_handle(), ms),
                f"mg_program_profile({self.name})", lib)
        return list(ms)

    def __del__(self):
        try:
            if self._prog is not None:
                L.load(self.f16).mg_program_destroy(self._prog)
        except Exception:
            pass
