"""Program builders: walk the SD-v2 UNet / AutoencoderKL structure (``arch.py``) and emit the
flat op programs that libmarigold_hip replays.  This is the host-side mirror of the module
seams the reference calls (marigold_depth_pipeline.py:461-463 ``unet(...)``, :491-492
``vae.encoder``/``quant_conv``, :512-513 ``post_quant_conv``/``vae.decoder``); all arithmetic
happens in the HIP kernels, torch only owns the device buffers.

Data layout: activations bf16 NHWC; latents / decoded maps fp32 NCHW (the reference's layout at
the pipeline boundary).  Buffers come from a size-keyed pool that is recycled along the static
program order, so the working set of a forward stays small and cache-friendly.
"""
import math
import os
import threading
from collections import defaultdict

import torch

from . import _lib as L
from . import ops as O
from . import tuning
from . import weights as Wm
from .arch import UNetConfig, VAEConfig, unet_up_resnet_channels

LATENT_SCALE = 0.18215  # marigold_depth_pipeline.py:118


def _tune(name, default):
    """Tuning switch ``name`` (an environment variable): honoured ONLY under MARIGOLD_TUNING=1 (same-box A/B runs, sweeps);
    without it the engine builds the product configuration whatever else the environment holds.  The library's switches sit
    behind the same gate (csrc/runtime.hip::mg_tuning_int)."""
    if os.environ.get("MARIGOLD_TUNING") != "1":
        return default
    v = os.environ.get(name)
    if v is None:
        return default
    return v if isinstance(default, str) else type(default)(int(v))


USE_PATCH = _tune("MARIGOLD_PATCH_CONV", True)          # patch-resident conv3x3 kernel where eligible
FUSE_GN = _tune("MARIGOLD_FUSE_GN", "auto")             # auto | all | none: GroupNorm apply inside the conv
GN_BYPRODUCT = _tune("MARIGOLD_GN_BYPRODUCT", True)     # GroupNorm partial sums from the producing convolution's epilogue
VAE_FLASH_SMALL_MIN_BLOCKS = _tune("MARIGOLD_VAE_FLASH_MIN_BLOCKS", 100)   # flash512 for launches of at least this many 128-query blocks
IGEMM73_CONV = _tune("MARIGOLD_IGEMM73_CONV", True)     # plain N = 320 k convolutions on the hand-placed 192 x 320 GEMM tile
IGEMM73_CONV_MIN_TILES = _tune("MARIGOLD_IGEMM73_CONV_MIN_TILES", 120)
IGEMM72_VAE = _tune("MARIGOLD_IGEMM72_VAE", True)       # plain 512-channel convolutions on the hand-placed implicit-GEMM tile
GN_STATS_ONE_LAUNCH = _tune("MARIGOLD_GN_STATS_ONE_LAUNCH", True)   # the two sources of a skip concat in one statistics launch
HEAD_CONV = _tune("MARIGOLD_HEAD_CONV", True)   # conv_norm_out + SiLU + conv_out (<= 4 channels) as one MG_OP_CONV3X3_HEAD launch
HEAD_CONV_MIN_PIXELS = _tune("MARIGOLD_HEAD_CONV_MIN_PIXELS", 1 << 18)
FOLD_SHORTCUT = _tune("MARIGOLD_FOLD_SHORTCUT", True)   # conv_shortcut as extra K of conv2 where conv2 runs on the implicit GEMM
GN_SLAB = _tune("MARIGOLD_GN_SLAB", True)               # GroupNorm as one launch per norm (MG_OP_GN_SLAB) where it applies
GN_SLAB_MIN_WG = _tune("MARIGOLD_GN_SLAB_MIN_WG", 64)   # ... from this many (image, channel window) workgroups,
GN_SLAB_SMALL_KB = _tune("MARIGOLD_GN_SLAB_SMALL_KB", 48)   # or fewer when a workgroup's share of the tensor is at most this (small ensembles)
ROWGEMM = _tune("MARIGOLD_ROWGEMM", True)               # row-resident GEMM (MG_OP_ROWGEMM) for the K = 320 token-local layers
XATTN_KSPLIT = _tune("MARIGOLD_XATTN_KSPLIT", True)     # deep-level collapsed cross-attention as the K-split kernel
ROWGEMM_WIDE = _tune("MARIGOLD_ROWGEMM_WIDE", True)     # ... and its K = 640 form for the 640-channel level's QKV / GEGLU
XATTN_IN_GEGLU = _tune("MARIGOLD_XATTN_IN_GEGLU", True)   # the collapsed cross-attention as the prologue of the row-resident GEGLU launch
ROWGEMM_MIN_M = _tune("MARIGOLD_ROWGEMM_MIN_M", 9216)   # below: the tile GEMM (one 96 x 96 member is 72 128-row workgroups)
# Measurement probe (never the product path): every launch of these op kinds is issued TWICE (idempotent kinds only: GroupNorm
# 2,3,4,9 / flash attention 6,11 write outputs they do not read) - the map's extra time is what that class costs with the
# other lanes running beside it, i.e. the most a faster kernel of that class could return (tools: scripts/gpu_ab_env.sh)
TWICE_KINDS = tuple(int(k) for k in _tune("MARIGOLD_TWICE_KINDS", "").split(",") if k)


class Act:
    """A bf16 NHWC activation [B][H][W][C] living in a pooled buffer."""
    __slots__ = ("t", "B", "H", "W", "C", "gn")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.gn = None   # (partial table, slots per image): its GroupNorm partial sums, left by the convolution that wrote it

    @property
    def M(self):
        return self.B * self.H * self.W

    @property
    def HW(self):
        return self.H * self.W


class Pool:
    def __init__(self, device):
        self.device = device
        self.free_lists = defaultdict(list)
        self.all = []
        self.bytes = 0

    def get(self, nbytes):
        nbytes = (int(nbytes) + 255) // 256 * 256
        fl = self.free_lists[nbytes]
        if fl:
            return fl.pop()
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.all.append(t)
        self.bytes += nbytes
        return t

    def put(self, t):
        self.free_lists[t.numel()].append(t)


class WeightStore:
    """Device-resident, kernel-ready weights derived from a diffusers-layout state dict."""

    def __init__(self, sd, device, dtype=torch.bfloat16):
        self.sd = sd
        self.device = device
        self.dtype = dtype   # the 16-bit operand type of the library build these weights feed (bf16, or fp16: libmarigold_hip_f16.so)
        self.cache = {}
        self._lock = threading.RLock()   # engine replicas (modules._EngineModule.replica) build their programs from one store

    def _memo(self, key, fn):
        with self._lock:
            if key not in self.cache:
                self.cache[key] = fn()
            return self.cache[key]

    def has(self, name):
        return f"{name}.weight" in self.sd

    def conv3x3(self, name):
        return self._memo(("c3", name), lambda: Wm.bf16(Wm.pack_conv3x3(self.sd[f"{name}.weight"].float()), self.device, self.dtype))

    def conv3x3_fold(self, name, shortcut):
        """conv3x3 ``name`` with the 1x1 convolution ``shortcut`` of another tensor folded in as extra K (MG_OP_IGEMM p[12]):
        rows [Cout][9 Cin | Cx] bf16 and the two biases' sum."""
        def f():
            ws_ = self.sd[f"{shortcut}.weight"].float()
            w = torch.cat([Wm.pack_conv3x3(self.sd[f"{name}.weight"].float()), ws_.reshape(ws_.shape[0], -1)], dim=1)
            b = self.sd[f"{name}.bias"].float() + self.sd[f"{shortcut}.bias"].float()
            return Wm.bf16(w, self.device, self.dtype), Wm.f32(b, self.device)
        return self._memo(("c3f", name), f)

    def conv3x3_subpix(self, name):
        return self._memo(("c3s", name), lambda: Wm.bf16(Wm.pack_conv3x3_subpix(self.sd[f"{name}.weight"].float()), self.device, self.dtype))

    def mat(self, name):  # Linear or 1x1 conv -> [N][K]
        def f():
            w = self.sd[f"{name}.weight"].float()
            return Wm.bf16(w.reshape(w.shape[0], -1), self.device, self.dtype)
        return self._memo(("m", name), f)

    def vec(self, key):
        return self._memo(("v", key), lambda: Wm.f32(self.sd[key], self.device))

    def bias(self, name):
        return self.vec(f"{name}.bias") if f"{name}.bias" in self.sd else None

    def time_emb_proj_all(self, names):
        """The ResNet blocks' time_emb_proj layers stacked along N: fp32 [sum cout][temb_dim], bias [sum cout]."""
        def f():
            w = torch.cat([self.sd[f"{n}.time_emb_proj.weight"].float() for n in names], dim=0)
            b = torch.cat([self.sd[f"{n}.time_emb_proj.bias"].float() for n in names], dim=0)
            return Wm.f32(w, self.device), Wm.f32(b, self.device)
        return self._memo(("tep", tuple(names)), f)

    def f32mat(self, name):
        return self._memo(("fm", name), lambda: Wm.f32(self.sd[f"{name}.weight"].reshape(self.sd[f"{name}.weight"].shape[0], -1), self.device))

    def small_conv_mfma(self, key, w4, bias):
        """Cout <= 16 conv3x3 as an MFMA GEMM: weights [Cout,Cin,3,3] -> bf16 [8|16][9*Cin] (rows >= Cout
        zero), bias fp32 [8|16]."""
        def f():
            co = w4.shape[0]
            npad = 8 if co <= 8 else 16
            w = torch.zeros(npad, 9 * w4.shape[1])
            w[:co] = Wm.pack_conv3x3(w4.float())
            b = torch.zeros(npad)
            b[:co] = bias.float()
            return Wm.bf16(w, self.device, self.dtype), Wm.f32(b, self.device)
        return self._memo(("scm", key), f)

    def conv_in_mfma(self, name):
        """conv3x3 from <= 8 channels as a GEMM over an im2col buffer: [Cout,Cin,3,3] -> bf16
        [Cout][Kp] with k = tap*Cin + c, zero-padded to Kp (64 or 128)."""
        def f():
            w = self.sd[f"{name}.weight"].float()
            k = 9 * w.shape[1]
            kp = (k + 63) // 64 * 64
            wp = torch.zeros(w.shape[0], kp)
            wp[:, :k] = Wm.pack_conv3x3(w)
            return Wm.bf16(wp, self.device, self.dtype), kp
        return self._memo(("cim", name), f)

    def qkv(self, prefix, with_bias):
        def f():
            w = Wm.pack_qkv(*(self.sd[f"{prefix}.{q}.weight"].float() for q in ("to_q", "to_k", "to_v")))
            b = None
            if with_bias:
                b = Wm.f32(torch.cat([self.sd[f"{prefix}.{q}.bias"].float() for q in ("to_q", "to_k", "to_v")]), self.device)
            return Wm.bf16(w, self.device, self.dtype), b
        return self._memo(("qkv", prefix), f)

    # ---- Linear layers with the preceding LayerNorm folded in (weights.fold_layernorm) ----
    def _ln(self, norm):
        return self.sd[f"{norm}.weight"].float(), self.sd[f"{norm}.bias"].float()

    def qkv_ln(self, prefix, norm):
        def f():
            w = Wm.pack_qkv(*(self.sd[f"{prefix}.{q}.weight"].float() for q in ("to_q", "to_k", "to_v")))
            wp, g, c = Wm.fold_layernorm(w, None, *self._ln(norm), dtype=self.dtype)
            return wp.to(self.device), Wm.f32(g, self.device), Wm.f32(c, self.device)
        return self._memo(("qkv_ln", prefix), f)

    def geglu_ln(self, name, norm):
        def f():
            w, b = Wm.pack_geglu(self.sd[f"{name}.weight"].float(), self.sd[f"{name}.bias"].float())
            wp, g, c = Wm.fold_layernorm(w, b, *self._ln(norm), dtype=self.dtype)
            return wp.to(self.device), Wm.f32(g, self.device), Wm.f32(c, self.device)
        return self._memo(("gg_ln", name), f)

    # ---- MG_OP_ROWGEMM forms: weights in fragment order + per-stage constants (weights.pack_rowgemm) ----
    def rg_mat(self, name):
        def f():
            w = self.sd[f"{name}.weight"].float()
            w = w.reshape(w.shape[0], -1)
            b = self.sd[f"{name}.bias"].float() if f"{name}.bias" in self.sd else torch.zeros(w.shape[0])
            return Wm.pack_rowgemm(w, b, dtype=self.dtype).to(self.device)
        return self._memo(("rg_m", name), f)

    def rg_qkv_ln(self, prefix, norm):
        def f():
            w = Wm.pack_qkv(*(self.sd[f"{prefix}.{q}.weight"].float() for q in ("to_q", "to_k", "to_v")))
            wp, g, c = Wm.fold_layernorm(w, None, *self._ln(norm), dtype=self.dtype)
            return Wm.pack_rowgemm(wp.float(), c, g, dtype=self.dtype).to(self.device)
        return self._memo(("rg_qkv_ln", prefix), f)

    def rg_geglu_ln(self, name, norm):
        def f():
            w, b = self.sd[f"{name}.weight"].float(), self.sd[f"{name}.bias"].float()
            order = Wm.rowgemm_geglu_order(w.shape[0])
            wp, g, c = Wm.fold_layernorm(w[order], b[order], *self._ln(norm), dtype=self.dtype)
            return Wm.pack_rowgemm(wp.float(), c, g, dtype=self.dtype).to(self.device)
        return self._memo(("rg_gg_ln", name), f)

    def rg_cross_ln(self, prefix, ctx, heads, norm):
        def f():
            wqk, vot, npad = Wm.cross_attention_tables(
                self.sd[f"{prefix}.to_q.weight"], self.sd[f"{prefix}.to_k.weight"],
                self.sd[f"{prefix}.to_v.weight"], self.sd[f"{prefix}.to_out.0.weight"], ctx, heads)
            assert npad == 64
            wp, g, c = Wm.fold_layernorm(wqk, None, *self._ln(norm), dtype=self.dtype)
            pack = Wm.pack_rowgemm_xattn if wqk.shape[1] == 320 else Wm.pack_rowgemm_xattn_ksplit
            return pack(wp.float(), c, g, vot, self.sd[f"{prefix}.to_out.0.bias"].float(), dtype=self.dtype).to(self.device)
        return self._memo(("rg_x_ln", prefix), f)

    def cross_ln(self, prefix, ctx, heads, norm):
        def f():
            wqk, vot, npad = Wm.cross_attention_tables(
                self.sd[f"{prefix}.to_q.weight"], self.sd[f"{prefix}.to_k.weight"],
                self.sd[f"{prefix}.to_v.weight"], self.sd[f"{prefix}.to_out.0.weight"], ctx, heads)
            wp, g, c = Wm.fold_layernorm(wqk, None, *self._ln(norm), dtype=self.dtype)
            return wp.to(self.device), Wm.f32(g, self.device), Wm.f32(c, self.device), Wm.bf16(vot, self.device, self.dtype), npad
        return self._memo(("x_ln", prefix), f)


class Builder:
    """Emits ops into an OpSeq, allocating/recycling activation buffers from a Pool."""

    def __init__(self, seq, pool, ws, groups=32):
        self.seq, self.pool, self.ws, self.groups = seq, pool, ws, groups
        self.dev = pool.device
        self.persist = {}

    # ---- buffers -------------------------------------------------------------------------
    def new(self, B, H, W, C):
        return Act(self.pool.get(B * H * W * C * 2), B, H, W, C)

    def raw(self, nbytes):
        return self.pool.get(nbytes)

    def ln_table(self, M, C):
        """Row-statistics buffer of a tensor [M][C] that feeds a folded LayerNorm: C/32 (sum, sum of squares) slots per
        row written by the producing GEMM's tiles, then [M] (mean, rstd) written by its last column tile per row block
        (MG_OP_IGEMM ln_out)."""
        return self.raw(M * (C // 32 + 1) * 8)

    @staticmethod
    def ln_mean_rstd(table, M, C):
        """Address of the (mean, rstd) rows inside ``ln_table(M, C)`` - the ln_in of the consuming Linear layers."""
        return table.data_ptr() + M * (C // 32) * 8

    def free(self, *xs):
        for x in xs:
            if x is None:
                continue
            if isinstance(x, Act) and x.gn is not None:
                self.pool.put(x.gn[0])
                x.gn = None
            self.pool.put(x.t if isinstance(x, Act) else x)

    def drop_gn(self, x):
        """Forget (and recycle) the GroupNorm partial table cached on ``x``: every op that WRITES into an existing Act calls this,
        so that ``gn_scale_shift`` never finalizes statistics of values the tensor no longer holds."""
        if isinstance(x, Act) and x.gn is not None:
            self.pool.put(x.gn[0])
            x.gn = None

    def zeros_persistent(self, key, nbytes):
        """A dedicated zero-initialised buffer (never recycled): V^T pad columns must stay 0."""
        if key not in self.persist:
            self.persist[key] = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)
        return self.persist[key]

    def add(self, op, label):
        if op.kind == L.OP_IGEMM and not op.p[14]:
            # the program's own split-K workspace (MG_OP_IGEMM p[14]): programs on concurrent streams - two maps in flight -
            # must not share the library's
            op.p[14] = self.zeros_persistent("splitk_ws", O.SPLITK_WS_BYTES).data_ptr()
        self.seq.add(tuning.apply(op), label)   # (MG_OP_IGEMM: the measured tile / split-K choice where the table has one)
        if op.kind in TWICE_KINDS:
            assert op.kind in (L.OP_GN_STATS, L.OP_GN_FINALIZE, L.OP_GN_APPLY, L.OP_GN_SLAB, L.OP_FLASH_ATTN64, L.OP_FLASH_ATTN512)
            self.seq.add(op, label + ".again")

    # ---- primitive layers ----------------------------------------------------------------
    @staticmethod
    def rowgemm_ok(x):
        """Token-local Linear layers of this activation on MG_OP_ROWGEMM?  K = 320 is what the kernel is built for (a wave
        keeps 32 rows x 320 channels in 80 registers); whole 32-row tiles inside an image (the V^T section and the folded
        GroupNorm are per image), enough 384-row workgroups for the chip, and the permuted V^T the QKV form writes."""
        return ROWGEMM and x.C == 320 and x.HW % 32 == 0 and x.M >= ROWGEMM_MIN_M

    @staticmethod
    def rowgemm_wide_ok(x):
        """The 640-channel level: MG_OP_ROWGEMM's K = 640 form (8 waves x 32 rows x 640 channels in 160 registers each) pays
        only where the columns can be split over two workgroups per 256-row block - the QKV projection (97 -> 80 us) and GEGLU
        (216 -> 194 us); the whole-row-statistics layers stay on the tile GEMM (90 workgroups: 64 vs 49 us),
        profiles/r3_rowgemm_k640.log."""
        return ROWGEMM and ROWGEMM_WIDE and x.C == 640 and x.HW % 32 == 0 and x.M >= 60 * 256

    @staticmethod
    def rowgemm_cfg(M, N, whole_rows=False, xattn=False, K=320):
        if K == 640:
            nb = -(-M // 256)
            return dict(waves=8, nsplit=max(1, min(N // 128, 256 // nb)))
        return Builder._rowgemm_cfg320(M, N, whole_rows, xattn)

    @staticmethod
    def _rowgemm_cfg320(M, N, whole_rows=False, xattn=False):
        """-> dict(waves=, nsplit=) of an MG_OP_ROWGEMM launch: 12 waves (384 rows) per workgroup when that still gives the
        chip >= 160 workgroups, else 8, else 4 with the N / 64 column stages shared out over several workgroups per row
        block (not for the forms that take whole-row statistics) - measured per ensemble size,
        profiles/r3_rowgemm_small_batch.log."""
        if M >= 160 * 384:
            return dict(waves=12)
        if M >= 120 * 256 or xattn:
            return dict(waves=8)
        if whole_rows:
            return dict(waves=4)
        nwg = -(-M // 128)
        return dict(waves=4, nsplit=max(1, min(N // 128, round(300 / nwg))))

    def gn_slab_ok(self, srcs, apply):
        """One-launch GroupNorm (MG_OP_GN_SLAB: a workgroup owns whole groups of an image over all rows)?  UNet-sized
        maps with enough (image, channel window) pairs to fill the chip; the normalising form keeps the rows in registers
        (<= 48 rows per thread).  Large tensors (UNet level 0, the VAE) stay on the chunked statistics / apply passes."""
        B, HW = srcs[0].B, srcs[0].HW
        C = sum(x.C for x in srcs)
        if not GN_SLAB or len(srcs) > 2 or C % self.groups or any(x.C % 4 for x in srcs):
            return False
        cpg = C // self.groups
        cw = cpg * (4 // math.gcd(cpg, 4))
        nwg = B * (C // cw)
        # (round 5) few workgroups are fine while each one's share is small: a single member's 24 x 24 / 12 x 12 maps (368 KB -
        # 1.5 MB) took a statistics launch (13-16 us: tickets, last-block finalize) + an apply launch (8-9 us) for lack of 64 of them
        if not (16 <= cw <= 128) or HW > 16384 or (nwg < GN_SLAB_MIN_WG and B * HW * C * 2 > nwg * GN_SLAB_SMALL_KB * 1024):
            return False
        # measured per layer (profiles/r3_groupnorm_slab_vs_chunked.log): a slab pass beats statistics + apply only while
        # the tensor is small enough that the chunked passes are launch/latency-bound (<= 16 MiB: UNet levels 2-3), or
        # when it replaces the TWO statistics launches of a skip concat at levels 1-3; on the big level-0 / VAE tensors
        # its B * C / cw workgroups are too few and the chunked passes win by 1.3-1.8x
        if not ((len(srcs) == 2 and HW <= 2304) or B * HW * C * 2 <= (16 << 20)):
            return False
        if apply:
            nt = 1024 if HW * cw * 2 >= 48 * 1024 else 256
            if -(-HW // (nt // (cw // 4))) > 48:
                return False
        return True

    def gn_slab(self, srcs, name, eps, silu, apply):
        """-> (scale/shift buffer, normalised Act | None) in ONE launch."""
        x = srcs[0]
        C = sum(y.C for y in srcs)
        ss = self.raw(x.B * 2 * C * 4)
        out = self.new(x.B, x.H, x.W, C) if apply else None
        self.add(O.gn_slab(x.t, None if out is None else out.t, ss, B=x.B, HW=x.HW, C=C, groups=self.groups,
                           gamma=self.ws.vec(f"{name}.weight"), beta=self.ws.vec(f"{name}.bias"), eps=eps, silu=silu,
                           x1=srcs[1].t if len(srcs) > 1 else None, C0=x.C), f"{name}.slab" + ("+apply" if apply else ""))
        return ss, out

    def gn_scale_shift(self, srcs, name, eps):
        """GroupNorm statistics over the channel concat of ``srcs`` (never materialised: one statistics launch per
        source into a shared partial table) -> fp32 [B][2][C] (scale, shift) buffer."""
        if self.gn_slab_ok(srcs, False):
            return self.gn_slab(srcs, name, eps, False, False)[0]
        B, HW = srcs[0].B, srcs[0].HW
        C = sum(x.C for x in srcs)
        if len(srcs) == 1 and srcs[0].gn is not None:   # the producing convolution left the partial sums: no pass over the tensor
            part, slots = srcs[0].gn
            ss = self.raw(B * 2 * C * 4)
            self.add(O.gn_finalize(part, self.ws.vec(f"{name}.weight"), self.ws.vec(f"{name}.bias"), ss, B=B, C=C,
                                   groups=self.groups, slots=slots, HW=HW, eps=eps), f"{name}.finalize")
            return ss
        # ~288 (chunk, image) blocks - about one per CU - of >= 32 rows: with eight row loads in flight per thread and the
        # chunk's tail fetched as one batch (round 3) a block streams its rows in 2-4 round trips, and the cost that is left
        # grows with the NUMBER of blocks (tickets, the last block's table reduction): 59 MB at E = 10 takes 16.3 us with 24-32
        # chunks per image against 23.5 with the 76 of round 2 (28.7 before the tail fix), profiles/r3_gn_stats_chunks.log
        chunks = max(1, min(HW // 32, 64, max(8, 288 // B)))
        slots = chunks * len(srcs)
        part = self.raw(B * slots * self.groups * 2 * 4)
        ss = self.raw(B * 2 * C * 4)
        counters = self.zeros_persistent("gn_counters", 4 * max(B, 1024))   # stream-ordered reuse; left zero by every use
        # the image's last-arriving statistics block turns the partial table into scale / shift (no finalize launch); the two
        # sources of a skip concat share one launch
        assert len(srcs) <= 2
        coff = 0
        for k, grp in enumerate([srcs] if GN_STATS_ONE_LAUNCH else [[x] for x in srcs]):
            x1 = grp[1] if len(grp) > 1 else None
            self.add(O.gn_stats(grp[0].t, part, B=B, HW=HW, C=grp[0].C, chunks=chunks, groups=self.groups, Ctot=C, coff=coff,
                                slot0=k * chunks, slots=slots, gamma=self.ws.vec(f"{name}.weight"), beta=self.ws.vec(f"{name}.bias"),
                                ss=ss, counters=counters, eps=eps, x1=x1.t if x1 else None, C1=x1.C if x1 else 0),
                     f"{name}.stats" + (f"{k}" if len(srcs) > 1 and not GN_STATS_ONE_LAUNCH else ""))
            coff += grp[0].C
        self.free(part)
        return ss

    def gn_apply(self, srcs, ss, name, silu):
        """Materialise silu?(norm(concat(srcs))) as one NHWC tensor (the concat happens in the pass' addressing)."""
        x = srcs[0]
        C = sum(y.C for y in srcs)
        out = self.new(x.B, x.H, x.W, C)
        self.add(O.gn_apply(x.t, ss, out.t, B=x.B, HW=x.HW, C=C, silu=silu,
                            x1=srcs[1].t if len(srcs) > 1 else None, C0=x.C), f"{name}.apply")
        return out

    def group_norm(self, x, name, eps, silu):
        if self.gn_slab_ok([x], True):
            ss, out = self.gn_slab([x], name, eps, silu, True)
            self.free(ss)
            return out
        ss = self.gn_scale_shift([x], name, eps)
        out = self.gn_apply([x], ss, name, silu)
        self.free(ss)
        return out

    # ---- patch-resident conv3x3 (MG_OP_CONV3X3) ------------------------------------------------------
    @staticmethod
    def patch_eligible(H, W, B=None, N=None, subpix=False):
        """16-pixel-wide tiles: maps that waste little of them (the 24x24 / 12x12 levels stay on the implicit GEMM,
        whose split-K also fills the chip there) and - when the batch and width are given - enough workgroups for the
        256 CUs (a single member at 96x96 has 36-72 spatial tiles: the implicit GEMM's smaller tiles fill the chip)."""
        if not ((H >= 32 and W >= 32) or (H % 16 == 0 and W % 16 == 0)):
            return False
        if B is None:
            return True
        if N % 256 == 0:
            th, bn = 16, 256
        elif N == 320 or (subpix and N % 320 == 0):
            th, bn = 8, 320
        else:
            th, bn = 16, 128
        grid = B * -(-H // th) * -(-W // 16) * -(-N // bn) * (4 if subpix else 1)
        return grid >= 240

    def fuse_norm_into_conv(self, B, H, W, Cin, N):
        """Apply the GroupNorm affine + SiLU inside the convolution's operand staging?  The fix-up runs once per
        workgroup and channel tile, i.e. (output-channel tiles) x 1.27 (halo) times per element, on VALU that the
        MFMAs do not hide: it pays when one workgroup covers all output channels (N <= 320) or when the separate
        pass would be HBM-bound on a tensor that no cache holds (profiles/r2_sweep3_patch_conv.log)."""
        mode = FUSE_GN
        if mode not in ("auto", "auto5"):
            return mode == "all"
        tiles_n = 1 if N in (128, 256, 320) else -(-N // (256 if N % 256 == 0 else 128))
        if mode == "auto5":   # the rule of rounds 2-5 (A/B)
            return tiles_n == 1 or B * H * W * Cin * 2 >= (192 << 20)
        # Round 6, measured layer by layer with the norm fused everywhere / nowhere (profiles/r6_ab_fuse_gn_per_layer.log): the
        # fix-up is VALU beside the MFMAs, the separate pass is HBM traffic - and with two maps in flight (section 6b) an HBM-bound
        # pass runs under the other map's matrix work.  Fused wins on the VAE's 128 / 256-channel levels (tensors of 0.75-1.5 GB:
        # +0.45 ... +0.97 ms per block unfused, and with two lanes the 256-channel level alone +1.4 ms per map); it LOSES where the plain convolution gets a hand-placed four-wave kernel that the
        # fused one does not - the UNet's 320-channel level from six members (-0.1 ... -0.66 ms per block) - and on the VAE's
        # 512-channel 192 x 192 level (two output-channel tiles repeat the fix-up: -0.17 ... -0.24 ms per block).
        if N == 320:
            return B * -(-H // 12) * -(-W // 16) < 280   # (six members: -1.1 ms with two lanes, -3 ms alone; five: a tie)
        return tiles_n == 1

    def conv3x3p(self, srcs, name, cout, *, ss=None, silu=False, rowvec=None, residual=None, out=None, subpix=False):
        x = srcs[0]
        skip = srcs[1] if len(srcs) > 1 else None
        H, W = (2 * x.H, 2 * x.W) if subpix else (x.H, x.W)
        if out is None:
            out = self.new(x.B, H, W, cout)
        Cin = x.C + (skip.C if skip else 0)
        w = self.ws.conv3x3_subpix(name) if subpix else self.ws.conv3x3(name)
        kw = dict(B=x.B, H=x.H, W=x.W, C0=x.C, N=cout, a1=skip.t if skip else None, C1=skip.C if skip else 0, subpix=subpix, ss=ss,
                  silu=silu, bias=self.ws.bias(name), rowvec=rowvec, rowvec_bcast=rowvec is not None,
                  residual=None if residual is None else residual.t, wz=cout * 4 * Cin if subpix else 0)
        op = O.conv3x3(x.t, w, out.t, **kw)
        # (round 4) the GroupNorm statistics of the output as a by-product of the 12-wave tiles' epilogue - the tensors of the
        # VAE's 768^2 / 384^2 levels are re-read at HBM speed otherwise (4.6 ms of statistics passes per decode at E = 10)
        cpg = cout // self.groups
        self.drop_gn(out)   # this launch rewrites `out`: a table left by an earlier producer describes other values
        if GN_BYPRODUCT and cout % self.groups == 0 and cpg in (4, 8, 16, 32) and out.HW * cout * 2 >= (8 << 20):
            slots = O.conv3x3_gn_slots(op, getattr(self.seq, "f16", False))
            if slots > 0:
                part = self.raw(x.B * slots * self.groups * 2 * 4)
                op = O.conv3x3(x.t, w, out.t, gn_part=part, gn_cpg=cpg, gn_slots=slots, **kw)
                out.gn = (part, slots)
        self.add(op, name)
        return out

    def conv_on_gemm(self, B, H, W, Cin, cout):
        """Does a plain stride-1 3x3 convolution of this shape run on MG_OP_IGEMM (True) or on the patch-resident kernel?
        (the rule of ``conv3x3`` below)"""
        M = B * H * W
        big = (IGEMM72_VAE and cout % 256 == 0 and Cin >= 512 and -(-M // 256) * (cout // 256) >= 720) or \
              (IGEMM73_CONV and cout % 320 == 0 and cout % 256 != 0 and Cin >= 320 and -(-M // 192) * (cout // 320) >= IGEMM73_CONV_MIN_TILES)
        return big or not (USE_PATCH and self.patch_eligible(H, W, B, cout, False))

    def conv3x3(self, x, name, cout, *, stride=1, pad=1, up=None, rowvec=None, residual=None, out=None, fold=None):
        """``fold`` = (shortcut layer name, x0, x1 | None): that 1x1 convolution of x0 (+ x1) rides as extra K (implicit GEMM only:
        the caller asks ``conv_on_gemm`` first)."""
        H, W = (up if up else (x.H, x.W))
        if stride == 1:
            Ho, Wo = H, W
        elif pad == 1:
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        else:  # VAE encoder: pad (0,1,0,1) then stride 2
            Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
        if out is None:
            out = self.new(x.B, Ho, Wo, cout)
        self.drop_gn(out)
        # (round 4) plain 512-channel VAE convolutions with >= 720 tiles of 256 x 256: the hand-placed implicit-GEMM tile
        # (variant 72, picked by the library) runs them at 1 284 TFLOP/s against 1 202 / 1 050 for the patch kernels
        if fold is not None:
            sc, x0, x1 = fold
            assert stride == 1 and pad == 1 and up is None and residual is None and self.conv_on_gemm(x.B, x.H, x.W, x.C, cout)
            wf, bf = self.ws.conv3x3_fold(name, sc)
            cx = x0.C + (x1.C if x1 is not None else 0)
            self.add(O.igemm(x.t, wf, out.t, B=x.B, H=x.H, W=x.W, Cin=x.C, Ho=Ho, Wo=Wo, N=cout, taps=9, stride=1, pad=1, bias=bf,
                             rowvec=rowvec, rowvec_bcast=rowvec is not None,
                             fold=(x0.t, None if x1 is None else x1.t, x0.C, cx)), f"{name}+{sc.rsplit('.', 1)[-1]}")
            return out
        big_gemm = (IGEMM72_VAE and up is None and stride == 1 and pad == 1 and cout % 256 == 0 and x.C >= 512 and
                    -(-x.M // 256) * (cout // 256) >= 720)
        # ... and the plain N = 320 k convolutions with a chip's worth of 192 x 320 tiles (the 640-channel level at 48 x 48) on
        # variant 73: 640 -> 640 1 193 vs 1 147-1 182 for the four-wave patch kernel, 1280 -> 640 1 303 vs 1 267-1 277
        # (round 5: from 120 tiles - six members at 48 x 48; with eight the four-wave patch kernel ran these at 760-790 TFLOP/s
        # where the GEMM tile does 1 170-1 300: the >= 200 of round 4 had been set at E = 10 only)
        big_gemm = big_gemm or (IGEMM73_CONV and up is None and stride == 1 and pad == 1 and cout % 320 == 0 and cout % 256 != 0 and
                                x.C >= 320 and -(-x.M // 192) * (cout // 320) >= IGEMM73_CONV_MIN_TILES)
        if USE_PATCH and not big_gemm and stride == 1 and pad == 1 and self.patch_eligible(x.H, x.W, x.B, cout, up is not None):
            if up is None:
                return self.conv3x3p([x], name, cout, rowvec=rowvec, residual=residual, out=out)
            if up == (2 * x.H, 2 * x.W) and rowvec is None and residual is None:
                return self.conv3x3p([x], name, cout, out=out, subpix=True)
        if up and up == (2 * x.H, 2 * x.W) and stride == 1 and pad == 1 and rowvec is None and residual is None:
            # exact 2x nearest up-sampling: four 2x2 convolutions on the low-resolution input (4/9 of the MACs)
            self.add(O.igemm(x.t, self.ws.conv3x3_subpix(name), out.t, B=x.B, H=x.H, W=x.W, Cin=x.C, Ho=x.H, Wo=x.W,
                             N=cout, taps=4, stride=1, pad=1, bias=self.ws.bias(name), batch_z=4,
                             zstrides=(0, cout * 4 * x.C, 0, 0)), name)
            return out
        self.add(O.igemm(x.t, self.ws.conv3x3(name), out.t, B=x.B, H=x.H, W=x.W, Cin=x.C, Ho=Ho, Wo=Wo,
                         N=cout, taps=9, stride=stride, pad=pad, up=up, bias=self.ws.bias(name),
                         rowvec=rowvec, rowvec_bcast=rowvec is not None,
                         residual=None if residual is None else residual.t), name)
        return out

    def conv_from_nchw(self, src0, src1, name, B, H, W, C0, C1, cout, bcast0=False):
        """conv3x3 (pad 1) from <= 8 fp32 NCHW channels (two sources = the folded torch.cat of
        marigold_depth_pipeline.py:456-458) to bf16 NHWC on the MFMA path: im2col + GEMM."""
        w, kp = self.ws.conv_in_mfma(name)
        col = self.raw(B * H * W * kp * 2)
        self.add(O.im2col_small(src0, src1, col, B=B, H=H, W=W, C0=C0, C1=C1, Kp=kp, bcast0=bcast0), f"{name}.im2col")
        out = self.new(B, H, W, cout)
        self.add(O.linear(col, w, out.t, M=B * H * W, K=kp, N=cout, bias=self.ws.bias(name),
                          k_alg=9 * (C0 + C1)), name)
        self.free(col)
        return out

    def head_conv_ok(self, x, cout):
        """norm + SiLU + conv3x3 to <= 4 channels as one MG_OP_CONV3X3_HEAD launch?  The VAE decoder's head (768^2 maps: 0.80 ms
        against 0.58 + 1.22 ms for the normalising pass + the implicit GEMM at ten members); NOT the UNet's - its 96^2 maps are
        360 workgroups of ten LDS-bound passes (79-102 us at ten members, 64-101 at one) where the pass + GEMM pair takes 54 / 33 us
        (profiles/r5_ops_hipevents*.tsv of the two final sessions)."""
        return HEAD_CONV and cout <= 4 and x.C % 32 == 0 and x.B * x.H * x.W >= HEAD_CONV_MIN_PIXELS

    def norm_conv_to_nchw(self, x, norm, eps, key, w4, bias, out, cout, **kw):
        """conv_norm_out -> SiLU -> conv_out -> the pointwise tail (diffusers' output heads of the UNet and the VAE decoder)."""
        if self.head_conv_ok(x, cout):
            ss = self.gn_scale_shift([x], norm, eps)
            self.conv_to_nchw(x, key, w4, bias, out, cout, head_ss=ss, **kw)
            self.free(ss)
        else:
            g = self.group_norm(x, norm, eps, True)
            self.conv_to_nchw(g, key, w4, bias, out, cout, **kw)
            self.free(g)

    def conv_to_nchw(self, x, key, w4, bias, out, cout, post=L.POST_NONE, scale=1.0, sched=None, head_ss=None):
        """conv3x3 (pad 1) bf16 NHWC -> <= 4 fp32 NCHW channels on the MFMA path: GEMM into a padded
        fp32 [M][8] buffer, then the pointwise tail (MG_OP_POST_NCHW).  ``head_ss``: x is the RAW tensor and the GroupNorm's
        scale / shift + SiLU are applied inside the convolution (MG_OP_CONV3X3_HEAD)."""
        w8, b8 = self.ws.small_conv_mfma(key, w4, bias)
        npad = w8.shape[0]
        tmp = self.raw(x.M * npad * 4)
        if head_ss is not None:
            self.add(O.conv3x3_head(x.t, head_ss, w8, b8, tmp, B=x.B, H=x.H, W=x.W, C=x.C, Cout=cout, ldo=npad, silu=True),
                     f"{key[:-len('conv_out')]}conv_norm_out+silu+conv_out" if key.endswith("conv_out") else f"norm+silu+{key}")
        else:
            self.add(O.igemm(x.t, w8, tmp, B=x.B, H=x.H, W=x.W, Cin=x.C, Ho=x.H, Wo=x.W, N=npad, taps=9, stride=1,
                             pad=1, bias=b8, epi=L.EPI_F32, ldo=npad, n_alg=cout), f"{key}")
        if sched is not None:   # (cx, cm, cn, noise): the scheduler update replaces the store of the model output
            cx, cm, cn, nz = sched
            self.add(O.post_nchw(tmp, out, B=x.B, HW=x.HW, Cout=cout, ldi=npad, post=L.POST_SCHED, scale=scale, noise=nz,
                                 cx=cx, cm=cm, cn=cn), f"{key}.post+scheduler.step")
        else:
            self.add(O.post_nchw(tmp, out, B=x.B, HW=x.HW, Cout=cout, ldi=npad, post=post, scale=scale), f"{key}.post")
        self.free(tmp)

    def dense(self, x, wt, bias, N, *, residual=None, out=None, epi=L.EPI_BF16, label="", K=None, out_dtype_bytes=2,
              skip=None, ln_out=None, ln=None):
        """x: Act viewed as [M][C]; wt: [N][K] bf16.  ``skip``: second channel source (K = x.C + skip.C).  ``ln_out``:
        buffer for the row statistics of the OUTPUT (the next LayerNorm's input); ``ln`` = (stats, g, c): the LayerNorm
        of the INPUT is folded into this layer (x holds the raw rows)."""
        M, K = x.M, (K or x.C + (skip.C if skip is not None else 0))
        n_out = N // 2 if epi == L.EPI_GEGLU else N
        # row-block tickets of the statistics hand-off: this program's own zeroed buffer (never the library-global one: two
        # programs on different streams would draw from the same slots)
        ctr = self.zeros_persistent("ln_counters", 4 * 65536) if ln_out is not None else None
        if out is None:
            out = Act(self.pool.get(M * n_out * out_dtype_bytes), x.B, x.H, x.W, n_out)
        self.drop_gn(out)
        self.add(O.linear(x.t, wt, out.t, M=M, K=K, N=N, bias=bias, epi=epi,
                          residual=None if residual is None else residual.t,
                          a1=None if skip is None else skip.t, C0=x.C if skip is not None else 0, ln_out=ln_out, ln_counters=ctr,
                          ln_in=None if ln is None else ln[0], ln_g=None if ln is None else ln[1],
                          ln_c=None if ln is None else ln[2]), label)
        return out

    # ---- composite blocks ----------------------------------------------------------------
    def resnet(self, x, name, cout, eps, temb_row=None, skip=None):
        """diffusers ResnetBlock2D on the channel concat of ``x`` and ``skip`` (the UNet's up blocks; the concat is
        never materialised on the patch path).  norm -> SiLU is applied inside the convolution's operand staging where
        ``fuse_norm_into_conv`` says it pays, else by one pass that also performs the concat."""
        srcs = [x] + ([skip] if skip is not None else [])
        Cin = sum(y.C for y in srcs)
        patch = USE_PATCH and self.patch_eligible(x.H, x.W, x.B, cout)

        def norm_conv(inputs, norm, conv, rowvec=None, residual=None, out=None, fold=None):
            cin = sum(y.C for y in inputs)
            fused = patch and self.fuse_norm_into_conv(x.B, x.H, x.W, cin, cout)
            assert fold is None or not fused
            if not fused and self.gn_slab_ok(inputs, True):   # statistics + normalisation (+ the concat) in one launch
                ss, h = self.gn_slab(inputs, norm, eps, True, True)
                y = self.conv3x3(h, conv, cout, rowvec=rowvec, residual=residual, out=out, fold=fold)
                self.free(h, ss)
                return y
            ss = self.gn_scale_shift(inputs, norm, eps)
            if fused:
                y = self.conv3x3p(inputs, conv, cout, ss=ss, silu=True, rowvec=rowvec, residual=residual, out=out)
            else:
                h = self.gn_apply(inputs, ss, norm, True)
                y = self.conv3x3(h, conv, cout, rowvec=rowvec, residual=residual, out=out, fold=fold)
                self.free(h)
            self.free(ss)
            return y

        h1 = norm_conv(srcs, f"{name}.norm1", f"{name}.conv1", rowvec=temb_row)
        sc = f"{name}.conv_shortcut"
        # (round 5) where conv2 runs on the implicit GEMM (the 48 x 48 ... 12 x 12 levels) its conv_shortcut - a 1x1 convolution of the
        # block's INPUT - rides as extra K of conv2: one launch instead of two, no residual tensor written and read back
        if (FOLD_SHORTCUT and self.ws.has(sc) and all(y.C % 64 == 0 for y in srcs) and cout % 64 == 0 and
                not (patch and self.fuse_norm_into_conv(x.B, x.H, x.W, cout, cout)) and self.conv_on_gemm(x.B, x.H, x.W, cout, cout)):
            out = norm_conv([h1], f"{name}.norm2", f"{name}.conv2", fold=(sc, x, skip))
        elif self.ws.has(f"{name}.conv_shortcut"):
            res = self.dense(x, self.ws.mat(f"{name}.conv_shortcut"), self.ws.bias(f"{name}.conv_shortcut"),
                             cout, label=f"{name}.conv_shortcut", skip=skip)
            out = norm_conv([h1], f"{name}.norm2", f"{name}.conv2", residual=res, out=res)  # in-place residual add
        else:
            assert skip is None and Cin == cout
            out = norm_conv([h1], f"{name}.norm2", f"{name}.conv2", residual=x)
        self.free(h1)
        return out

    def self_attention(self, h, st, prefix, norm, heads, st_out):
        """h += to_out(attn(LN(h))) with the LayerNorm folded into the fused QKV projection (``st`` = row statistics of
        h from its producer); the to_out epilogue writes the statistics of the new h into ``st_out``.  Head dim 64."""
        C, B, T, M = h.C, h.B, h.HW, h.M
        ldvt = (T + 63) // 64 * 64
        qk = self.raw(M * 2 * C * 2)
        vt = self.zeros_persistent(("vt", B, C, ldvt), B * C * ldvt * 2)
        # V^T with its keys in the QK^T accumulator order inside groups of 16: the QKV epilogue skips its lane regroup and
        # the attention kernel (generation 3) its v_permlane32_swap - a format private to this producer / consumer pair
        perm = T % 16 == 0
        rg = perm and self.rowgemm_ok(h)
        if rg or (perm and self.rowgemm_wide_ok(h)):
            self.add(O.rowgemm(h.t, self.ws.rg_qkv_ln(prefix, norm), qk, M=M, K=C, N=3 * C, form=L.RG_QKV, ldo=2 * C,
                               ln_in=self.ln_mean_rstd(st, M, C), vt=vt, tokens=T, ldt=ldvt, trans_from=2 * C,
                               **self.rowgemm_cfg(M, 3 * C, K=C)), f"{prefix}.qkv")
        else:
            wqkv, g, c = self.ws.qkv_ln(prefix, norm)
            self.add(O.igemm(h.t, wqkv, qk, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=3 * C, ldo=2 * C, out2=vt,
                             trans_from=2 * C, ldt=ldvt, ln_in=self.ln_mean_rstd(st, M, C), ln_g=g, ln_c=c, trans_perm=perm), f"{prefix}.qkv")
        o = self.new(h.B, h.H, h.W, C)
        ws = self.zeros_persistent(("flash_ws",), O.FLASH_WS_BYTES_AUTO)   # key-split blocks of the hand-placed kernel (tickets stay 0)
        self.add(O.flash_attn64(qk, qk.data_ptr() + C * 2, vt, o.t, B=B, heads=heads, Ntok=T, ldq=2 * C,
                                ldo=C, ldvt=ldvt, sq=T * 2 * C, sk=T * 2 * C, svt=C * ldvt, so=T * C,
                                scale=1.0 / math.sqrt(C // heads), vt_perm=perm, ws=ws, ws_bytes=O.FLASH_WS_BYTES_AUTO), f"{prefix}.flash")
        self.free(qk)
        if rg:
            self.add(O.rowgemm(o.t, self.ws.rg_mat(f"{prefix}.to_out.0"), h.t, M=M, K=C, N=C, residual=h.t,
                               ln_out=self.ln_mean_rstd(st_out, M, C), **self.rowgemm_cfg(M, C, whole_rows=True)), f"{prefix}.to_out")
        else:
            self.dense(o, self.ws.mat(f"{prefix}.to_out.0"), self.ws.bias(f"{prefix}.to_out.0"), C,
                       residual=h, out=h, label=f"{prefix}.to_out", ln_out=st_out)
        self.free(o)

    def cross_attention2(self, h, st, prefix, norm, heads, ctx, st_out):
        """h += attn2(LN(h), ctx) with the 2-token context collapsed into two thin GEMMs and the LayerNorm folded into
        the first one (``st``: row statistics of h, ``st_out``: of the new h for the next folded LayerNorm)."""
        C, M = h.C, h.M
        if XATTN_KSPLIT and ROWGEMM and 2 * heads <= 64 and C in (640, 1280) and M % 32 == 0:
            # the deep levels: 32-row workgroups whose four waves split K (scores) and the output channels (blend)
            self.add(O.rowgemm(h.t, self.ws.rg_cross_ln(prefix, ctx, heads, norm), h.t, M=M, K=C, N=64, form=L.RG_XATTN,
                               ln_in=self.ln_mean_rstd(st, M, C), ln_out=self.ln_mean_rstd(st_out, M, C),
                               sm_cols=2 * heads, sm_scale=1.0 / math.sqrt(C // heads)), f"{prefix}.scores+softmax2+blend")
            return
        if 2 * heads <= 64 and self.rowgemm_ok(h):
            # the same single launch in the row-resident form: the residual stream is read once (registers) and written once
            self.add(O.rowgemm(h.t, self.ws.rg_cross_ln(prefix, ctx, heads, norm), h.t, M=M, K=C, N=64, form=L.RG_XATTN,
                               ln_in=self.ln_mean_rstd(st, M, C), ln_out=self.ln_mean_rstd(st_out, M, C),
                               sm_cols=2 * heads, sm_scale=1.0 / math.sqrt(C // heads), **self.rowgemm_cfg(M, C, xattn=True)),
                     f"{prefix}.scores+softmax2+blend")
            return
        # ONE launch on the tile GEMM: scores GEMM with the LayerNorm folded in, the 2-key softmax on its accumulators, the
        # probabilities as the register operand of the blend GEMM (x the context's values pushed through to_out), + bias +
        # residual, in place on the residual stream, (mean, rstd) of the new rows for the next folded LayerNorm
        wqk, g, c, vot, npad = self.ws.cross_ln(prefix, ctx, heads, norm)
        if npad == 64 and C % 32 == 0:
            self.add(O.linear(h.t, wqk, h.t, M=M, K=C, N=npad, epi=L.EPI_XATTN2, ln_in=self.ln_mean_rstd(st, M, C),
                              ln_g=g, ln_c=c, sm_scale=1.0 / math.sqrt(C // heads), sm_cols=2 * heads, out2=vot, c2=C, ldo=C,
                              bias=self.ws.bias(f"{prefix}.to_out.0"), residual=h.t, ldr=C,
                              ln_out=self.ln_mean_rstd(st_out, M, C)),
                     f"{prefix}.scores+softmax2+blend")
            return
        # more than 32 heads (no published checkpoint): two launches - scores + pair softmax, then the blend
        p = self.raw(M * npad * 2)
        self.add(O.linear(h.t, wqk, p, M=M, K=C, N=npad, epi=L.EPI_SOFTMAX2, ln_in=self.ln_mean_rstd(st, M, C), ln_g=g, ln_c=c,
                          sm_scale=1.0 / math.sqrt(C // heads), sm_cols=2 * heads), f"{prefix}.scores+softmax2")
        self.add(O.linear(p, vot, h.t, M=M, K=npad, N=C, bias=self.ws.bias(f"{prefix}.to_out.0"),
                          residual=h.t, ln_out=st_out, ln_counters=self.zeros_persistent("ln_counters", 4 * 65536)), f"{prefix}.blend")
        self.free(p)

    def transformer(self, x, name, heads, ctx):
        """diffusers Transformer2DModel / BasicTransformerBlock.  The three LayerNorms never run as passes: each Linear
        that consumes one takes the raw residual stream and corrects in its epilogue (MG_OP_IGEMM ln_in), with the row
        statistics written by the epilogue of the GEMM that produced the stream (ln_out)."""
        C = x.C
        st = [self.ln_table(x.M, C) for _ in range(3)]
        rg = self.rowgemm_ok(x)
        if rg:
            # the GroupNorm never runs as a pass: statistics only, its scale / shift applied while proj_in loads its rows
            if self.gn_slab_ok([x], False):
                ss, _ = self.gn_slab([x], f"{name}.norm", 1e-6, False, False)
            else:
                ss = self.gn_scale_shift([x], f"{name}.norm", 1e-6)
            h = self.new(x.B, x.H, x.W, C)
            self.add(O.rowgemm(x.t, self.ws.rg_mat(f"{name}.proj_in"), h.t, M=x.M, K=C, N=C, gn_ss=ss, tokens=x.HW,
                               ln_out=self.ln_mean_rstd(st[0], x.M, C), **self.rowgemm_cfg(x.M, C, whole_rows=True)), f"{name}.proj_in")
            self.free(ss)
        else:
            g = self.group_norm(x, f"{name}.norm", 1e-6, False)
            h = self.dense(g, self.ws.mat(f"{name}.proj_in"), self.ws.bias(f"{name}.proj_in"), C,
                           label=f"{name}.proj_in", ln_out=st[0])
            self.free(g)
        b = f"{name}.transformer_blocks.0"
        self.self_attention(h, st[0], f"{b}.attn1", f"{b}.norm1", heads, st[1])
        # (round 6) at the 320-channel level the collapsed cross-attention is the PROLOGUE of the GEGLU launch: that launch holds the
        # rows in registers anyway - it updates them (and stores them once, for ff.out's residual), takes the next LayerNorm's
        # statistics from its own sums and goes on; no cross-attention launch, one read of the residual stream less
        gcfg = self.rowgemm_cfg(h.M, 8 * C, K=C) if (rg or self.rowgemm_wide_ok(h)) else None
        fuse_x = rg and XATTN_IN_GEGLU and 2 * heads <= 64 and gcfg.get("nsplit", 1) <= 1
        if not fuse_x:
            self.cross_attention2(h, st[1], f"{b}.attn2", f"{b}.norm2", heads, ctx, st[2])
        if gcfg is not None:
            ff = self.new(h.B, h.H, h.W, 4 * C)
            xkw = dict(xattn=self.ws.rg_cross_ln(f"{b}.attn2", ctx, heads, f"{b}.norm2"), xout=h.t, sm_cols=2 * heads,
                       sm_scale=1.0 / math.sqrt(C // heads)) if fuse_x else {}
            self.add(O.rowgemm(h.t, self.ws.rg_geglu_ln(f"{b}.ff.net.0.proj", f"{b}.norm3"), ff.t, M=h.M, K=C, N=8 * C,
                               form=L.RG_GEGLU, ln_in=self.ln_mean_rstd(st[1] if fuse_x else st[2], h.M, C), **gcfg, **xkw),
                     f"{b}.attn2+ff.geglu" if fuse_x else f"{b}.ff.geglu")
        else:
            wg, gg, cg = self.ws.geglu_ln(f"{b}.ff.net.0.proj", f"{b}.norm3")
            ff = self.dense(h, wg, None, 8 * C, epi=L.EPI_GEGLU, label=f"{b}.ff.geglu",
                            ln=(self.ln_mean_rstd(st[2], h.M, C), gg, cg))
        self.free(*st)
        # (round 5, tried and NOT kept: GEGLU -> ff.out in two row chunks through a chunk-sized intermediate, so that ff.out reads
        # its 236 MB A operand out of the 256 MB memory-side cache - tools/ubench/mall_probe.py: 6.6-7.2 TB/s for working sets that
        # fit, 5.1-5.4 beyond.  ff.out gained 0.7 ms per map, the two half-size GEGLU launches lost 3.0:
        # profiles/r5_ff_row_chunks_lost.log)
        self.dense(ff, self.ws.mat(f"{b}.ff.net.2"), self.ws.bias(f"{b}.ff.net.2"), C, residual=h, out=h,
                   label=f"{b}.ff.out")
        self.free(ff)
        # NB: the GEMM input must never alias its output (other column tiles still read it)
        if rg:
            self.add(O.rowgemm(h.t, self.ws.rg_mat(f"{name}.proj_out"), x.t, M=h.M, K=C, N=C, residual=x.t,
                               **self.rowgemm_cfg(h.M, C)), f"{name}.proj_out")
            out = x
        else:
            out = self.dense(h, self.ws.mat(f"{name}.proj_out"), self.ws.bias(f"{name}.proj_out"), C,
                             residual=x, out=x, label=f"{name}.proj_out")
        self.free(h)
        return out

    def vae_attention(self, x, name):
        """Single-head d=C attention of the VAE mid block: the flash form (MG_OP_FLASH_ATTN512) for the published width of
        512 channels, the materialised fp32 scores for any other width (the tiny test architecture)."""
        C, B, T, M = x.C, x.B, x.HW, x.M
        g = self.group_norm(x, f"{name}.group_norm", 1e-6, False)
        wqkv, bqkv = self.ws.qkv(name, True)
        ldp = (T + 63) // 64 * 64
        Tn = (T + 7) // 8 * 8           # score columns in whole 8-column store groups: the K rows past the last image's
        qk = self.raw((M + 8) * 2 * C * 2)   # tokens are slack rows, their columns are never read by the softmax
        vt = self.zeros_persistent(("vvt", B, C, ldp), B * C * ldp * 2)
        self.add(O.igemm(g.t, wqkv, qk, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=3 * C, ldo=2 * C, bias=bqkv,
                         out2=vt, trans_from=2 * C, ldt=ldp), f"{name}.qkv")
        self.free(g)
        o = self.new(x.B, x.H, x.W, C)
        if C == 512 and B * ((T + 127) // 128) >= VAE_FLASH_SMALL_MIN_BLOCKS:
            # round 4: flash form - the T x T scores (340 MB of fp32 per image at 96 x 96 latent pixels) never leave the
            # registers.  One workgroup per 128 queries and CU: a launch that does not fill the chip (a single image: 72
            # workgroups - the encoder always, the decoder of a one-member shard) takes 0.93 ms against 0.50 ms for the
            # three-stage form below (profiles/r4_flash512.log; two-wave workgroups of 64 queries: 1.12 ms), so launches of
            # fewer than 100 query blocks take the three-stage form (round 5: -0.5 ms per map at every ensemble size)
            self.add(O.flash_attn512(qk, qk.data_ptr() + C * 2, vt, o.t, B=B, Ntok=T, ldq=2 * C, ldo=C, ldvt=ldp,
                                     sq=T * 2 * C, sk=T * 2 * C, svt=C * ldp, so=T * C, scale=1.0 / math.sqrt(C)), f"{name}.flash")
            self.free(qk)
        else:
            # scores GEMM -> row softmax -> P V, everything in one batched launch per stage
            sc = self.raw(B * T * ldp * 4)
            self.add(O.igemm(qk, qk.data_ptr() + C * 2, sc, B=1, H=T, W=1, Cin=C, Ho=T, Wo=1, N=Tn, epi=L.EPI_F32, ldo=ldp,
                             lda=2 * C, ldw=2 * C, batch_z=B, n_alg=T, zstrides=(T * 2 * C, T * 2 * C, T * ldp, 0),
                             scale=1.0 / math.sqrt(C)), f"{name}.scores")
            self.free(qk)
            p = self.raw(B * T * ldp * 2)
            self.add(O.softmax_rows(sc, p, R=B * T, ncols=T, lds=ldp, ldp=ldp), f"{name}.softmax")
            self.free(sc)   # (the fp32 scores and the bf16 probabilities never live side by side beyond this launch)
            self.add(O.igemm(p, vt, o.t, B=1, H=T, W=1, Cin=ldp, Ho=T, Wo=1, N=C, lda=ldp, ldw=ldp, batch_z=B,
                             zstrides=(T * ldp, C * ldp, T * C, 0)), f"{name}.pv")
            self.free(p)
        out = self.dense(o, self.ws.mat(f"{name}.to_out.0"), self.ws.bias(f"{name}.to_out.0"), C,
                         residual=x, label=f"{name}.to_out")
        self.free(o)
        return out


# ------------------------------------------------------------------------------------------ UNet

def sinusoid_table(timesteps, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin], fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = torch.as_tensor(timesteps, dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def unet_resnet_names(cfg: UNetConfig):
    """[(state-dict prefix, cout)] of every ResNet block, in execution order."""
    boc = list(cfg.block_out_channels)
    n = len(boc)
    names = []
    for i in range(n):
        for j in range(cfg.layers_per_block):
            names.append((f"down_blocks.{i}.resnets.{j}", boc[i]))
    names += [("mid_block.resnets.0", boc[-1]), ("mid_block.resnets.1", boc[-1])]
    for (i, j, _rin, _skip, out) in unet_up_resnet_channels(cfg):
        names.append((f"up_blocks.{i}.resnets.{j}", out))
    return names


def emit_time_embeddings(bld, cfg, timesteps):
    """Ops computing, for all T steps at once, the per-ResNet time-embedding projections
    (sinusoid -> Linear -> SiLU -> Linear -> [SiLU -> Linear] per block).  Returns
    {resnet prefix: fp32 tensor [T][cout]}."""
    ws, dev = bld.ws, bld.dev
    T = len(timesteps)
    c0, td = cfg.block_out_channels[0], cfg.temb_dim
    sin = bld.seq.hold(sinusoid_table(list(timesteps), c0).to(dev))
    e1 = bld.seq.hold(torch.empty(T, td, device=dev))
    emb = bld.seq.hold(torch.empty(T, td, device=dev))
    bld.add(O.linear_small_m(sin, ws.f32mat("time_embedding.linear_1"), ws.bias("time_embedding.linear_1"), e1,
                             M=T, N=td, K=c0, act_out=1), "time_embedding.linear_1")
    bld.add(O.linear_small_m(e1, ws.f32mat("time_embedding.linear_2"), ws.bias("time_embedding.linear_2"), emb,
                             M=T, N=td, K=td), "time_embedding.linear_2")
    # every ResNet block's projection in ONE launch: their weights stacked along N (24 launches of ~20 us each otherwise);
    # a block's row for step s is table[name][s] - a view of the [T][sum of couts] result
    names = list(unet_resnet_names(cfg))
    ntot = sum(cout for _, cout in names)
    wcat, bcat = ws.time_emb_proj_all([name for name, _ in names])
    tall = bld.seq.hold(torch.empty(T, ntot, device=dev))
    bld.add(O.linear_small_m(emb, wcat, bcat, tall, M=T, N=ntot, K=td, act_in=1), "resnets.time_emb_proj")
    table, off = {}, 0
    for name, cout in names:
        table[name] = tall[:, off:off + cout]
        off += cout
    return table


def emit_unet_forward(bld, cfg, ctx, rgb_latent, x_latent, eps_out, temb_table, step, B, h, w, sched=None):
    """One UNet forward: eps_out[B,4,h,w] = unet(cat(rgb_latent, x_latent), t_step, ctx).
    rgb_latent is [1,4,h,w] (shared by all members) or [B,4,h,w].  With ``sched`` = (cx, cm, cn, noise) the model
    output is not stored: the scheduler update x_latent <- cx x_latent + cm out + cn noise is conv_out's tail."""
    ws = bld.ws
    boc = list(cfg.block_out_channels)
    n = len(boc)

    def trow(name, cout):
        assert temb_table[name].shape[1] == cout
        return temb_table[name][step].data_ptr()   # cout contiguous fp32 values

    c0 = boc[0]
    cin = ws.sd["conv_in.weight"].shape[1]
    c_rgb = rgb_latent.shape[1]   # 4 image-latent channels + 4 per predicted modality (IID: 8 / 12)
    x = bld.conv_from_nchw(rgb_latent, x_latent, "conv_in", B, h, w, c_rgb, cin - c_rgb, c0,
                           bcast0=rgb_latent.shape[0] == 1)
    skips = [x]
    force_size = any(d % (2 ** (n - 1)) != 0 for d in (h, w))
    for i in range(n):
        for j in range(cfg.layers_per_block):
            name = f"down_blocks.{i}.resnets.{j}"
            y = bld.resnet(x, name, boc[i], 1e-5, trow(name, boc[i]))
            if i < n - 1:
                y = bld.transformer(y, f"down_blocks.{i}.attentions.{j}", cfg.heads[i], ctx)
            x = y
            skips.append(x)
        if i < n - 1:
            x = bld.conv3x3(x, f"down_blocks.{i}.downsamplers.0.conv", boc[i], stride=2, pad=1)
            skips.append(x)
    name = "mid_block.resnets.0"
    y = bld.resnet(x, name, boc[-1], 1e-5, trow(name, boc[-1]))
    y = bld.transformer(y, "mid_block.attentions.0", cfg.heads[-1], ctx)
    name = "mid_block.resnets.1"
    x2 = bld.resnet(y, name, boc[-1], 1e-5, trow(name, boc[-1]))
    bld.free(y)
    x = x2  # the pre-mid x is still referenced by skips[-1]
    rheads = list(cfg.heads)[::-1]
    for (i, j, _rin, _skip, cout) in unet_up_resnet_channels(cfg):
        skip = skips.pop()
        name = f"up_blocks.{i}.resnets.{j}"
        # torch.cat([hidden, skip], dim=1) is folded into the consumers: two statistics launches, two-source operand
        # staging in conv1 / conv_shortcut (or the normalisation pass that materialises the input where it is not fused)
        y = bld.resnet(x, name, cout, 1e-5, trow(name, cout), skip=skip)
        bld.free(x, skip)
        if i > 0:
            y = bld.transformer(y, f"up_blocks.{i}.attentions.{j}", rheads[i], ctx)
        x = y
        if j == cfg.layers_per_block and i < n - 1:
            if force_size:
                up = (skips[-1].H, skips[-1].W)
            else:
                up = (2 * x.H, 2 * x.W)
            y = bld.conv3x3(x, f"up_blocks.{i}.upsamplers.0.conv", cout, up=up)
            bld.free(x)
            x = y
    bld.norm_conv_to_nchw(x, "conv_norm_out", 1e-5, "conv_out", ws.sd["conv_out.weight"], ws.sd["conv_out.bias"],
                          x_latent if sched else eps_out, cfg.out_channels, sched=sched)
    bld.free(x)


# ------------------------------------------------------------------------------------------ VAE

def emit_vae_encode(bld, cfg: VAEConfig, rgb, lat_out, B, H, W):
    """lat_out[B,4,H/8,W/8] = 0.18215 * mean(quant_conv(encoder(rgb)))  (fp32 NCHW in/out).
    quant_conv (1x1) and the posterior-mean selection are composed into the encoder's conv_out."""
    ws = bld.ws
    boc = list(cfg.block_out_channels)
    x = bld.conv_from_nchw(rgb, None, "encoder.conv_in", B, H, W, 3, 0, boc[0])
    for i, c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            y = bld.resnet(x, f"encoder.down_blocks.{i}.resnets.{j}", c, 1e-6)
            bld.free(x)
            x = y
        if i < len(boc) - 1:
            y = bld.conv3x3(x, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, stride=2, pad=0)
            bld.free(x)
            x = y
    x = emit_vae_mid(bld, x, "encoder.mid_block")
    g = bld.group_norm(x, "encoder.conv_norm_out", 1e-6, True)
    bld.free(x)
    L4 = cfg.latent_channels

    def composed():
        wc = ws.sd["encoder.conv_out.weight"].double()          # [2L, C, 3, 3]
        bc = ws.sd["encoder.conv_out.bias"].double()
        wq = ws.sd["quant_conv.weight"].double().reshape(2 * L4, 2 * L4)[:L4]   # mean rows
        bq = ws.sd["quant_conv.bias"].double()[:L4]
        w = torch.einsum("om,mcyx->ocyx", wq, wc)
        b = wq @ bc + bq
        return w.float(), b.float()
    wcomp, bcomp = ws._memo(("enc_tail",), composed)
    bld.conv_to_nchw(g, "encoder.conv_out+quant_conv", wcomp, bcomp, lat_out, L4, scale=LATENT_SCALE)
    bld.free(g)
    return g.H, g.W


def emit_vae_mid(bld, x, name):
    y = bld.resnet(x, f"{name}.resnets.0", x.C, 1e-6)
    bld.free(x)
    z = bld.vae_attention(y, f"{name}.attentions.0")
    bld.free(y)
    out = bld.resnet(z, f"{name}.resnets.1", z.C, 1e-6)
    bld.free(z)
    return out


def emit_vae_decode(bld, cfg: VAEConfig, latent, out, B, h, w, post):
    """out = tail(decoder(post_quant_conv(latent / 0.18215))); ``post`` fuses the pipeline's
    pointwise tail (depth: mean/clip/shift -> [B,1,H,W]; normals: clip/normalise -> [B,3,H,W])."""
    ws = bld.ws
    rev = list(cfg.block_out_channels)[::-1]
    L4 = cfg.latent_channels
    z = bld.raw(B * L4 * h * w * 4)
    bld.add(O.latent_1x1(latent, ws.f32mat("post_quant_conv"), ws.bias("post_quant_conv"), z, B=B, Ci=L4,
                         Co=L4, HW=h * w, scale=1.0 / LATENT_SCALE), "post_quant_conv")
    x = bld.conv_from_nchw(z, None, "decoder.conv_in", B, h, w, L4, 0, rev[0])
    bld.free(z)
    x = emit_vae_mid(bld, x, "decoder.mid_block")
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            y = bld.resnet(x, f"decoder.up_blocks.{i}.resnets.{j}", c, 1e-6)
            bld.free(x)
            x = y
        if i < len(rev) - 1:
            y = bld.conv3x3(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, up=(2 * x.H, 2 * x.W))
            bld.free(x)
            x = y
    bld.norm_conv_to_nchw(x, "decoder.conv_norm_out", 1e-6, "decoder.conv_out", ws.sd["decoder.conv_out.weight"],
                          ws.sd["decoder.conv_out.bias"], out, 3, post=post)
    bld.free(x)
    return x.H, x.W
