"""Algorithmic work of one ``mg_op`` (FLOPs = 2 x MACs of the contraction; bytes = one read of
every input + one write of every output, weights counted once per launch) and the kernel class
it runs on.  bench.py turns per-op HIP-event timings into per-class achieved TFLOP/s / GB/s
against the gfx950 rooflines; DESIGN.md quotes the same formulas.
"""
from . import _lib as L

MFMA_PEAK_TFLOPS = 2500.0   # bf16 dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # HBM3E spec (6.29 TB/s measured streaming)

CLASS = {
    L.OP_IGEMM: "igemm_mfma", L.OP_ROWGEMM: "rowgemm_mfma", L.OP_CONV3X3: "conv3x3_patch", L.OP_FLASH_ATTN64: "flash_attn64", L.OP_FLASH_ATTN512: "flash_attn512", L.OP_GN_STATS: "groupnorm",
    L.OP_GN_FINALIZE: "groupnorm", L.OP_GN_APPLY: "groupnorm", L.OP_GN_SLAB: "groupnorm",
    L.OP_SOFTMAX_ROWS: "softmax",
    L.OP_SCHED_STEP: "scheduler_step", L.OP_LINEAR_SMALL_M: "time_embedding",
    L.OP_LATENT_1X1: "boundary_conv", L.OP_POST_NCHW: "boundary_conv", L.OP_CONV3X3_HEAD: "boundary_conv", L.OP_IM2COL_SMALL: "boundary_conv", L.OP_ENS_DEPTH_STATS: "ensemble", L.OP_ENS_DEPTH_MEDIAN: "ensemble",
    L.OP_ENS_DEPTH_NORM: "ensemble", L.OP_ENS_NORMALS: "ensemble", L.OP_RESIZE: "resize", L.OP_COLORIZE: "resize", L.OP_MEMSET: "memops", L.OP_COPY: "memops",
}
BOUND = {"igemm_mfma": "mfma", "rowgemm_mfma": "mfma", "conv3x3_patch": "mfma", "flash_attn64": "mfma", "flash_attn512": "mfma"}   # everything else is HBM-bound streaming


def op_cost(op):
    """-> (class name, algorithmic FLOPs, algorithmic HBM bytes) of one launch."""
    k, i, l = op.kind, op.i, op.l
    cls = CLASS.get(k, "other")
    flops = byts = 0
    if k == L.OP_IGEMM:
        B, H, W, Cin, Ho, Wo, N, taps = (i[j] for j in range(8))
        epi, bz = i[12], max(1, i[15])
        M, K = B * Ho * Wo, taps * Cin
        flops = 2 * M * (i[22] or N) * (i[23] or K) * bz   # i[22] / i[23]: un-padded N / K of the boundary convs
        cx = i[32] if op.p[12] else 0                      # a folded 1x1 convolution: extra K, its input and weights read once
        flops += 2 * M * N * cx
        n_out = N // 2 if epi == L.EPI_GEGLU else N
        osz = 4 if epi == L.EPI_F32 else 2   # bf16 for the plain, GEGLU and pair-softmax epilogues
        if taps == 4:   # sub-pixel up-sampling conv: the 4 parities share one input, each writes its own output pixels
            byts = B * H * W * Cin * 2 + bz * (N * K * 2 + M * n_out * osz)
        else:
            byts = bz * (B * H * W * Cin * 2 + N * K * 2 + M * n_out * osz)
        if epi == L.EPI_XATTN2:   # second stage: P [M][N] x W2 [c2][N] -> out [M][c2] (+ residual)
            c2 = i[28]
            flops += 2 * M * c2 * N
            byts = B * H * W * Cin * 2 + N * K * 2 + c2 * N * 2 + M * c2 * 2 + (M * c2 * 2 if op.p[5] else 0)
        elif op.p[5]:
            byts += bz * M * n_out * 2   # fused residual read
        byts += (M + N) * cx * 2
    elif k == L.OP_ROWGEMM:
        M, K, N, form = i[0], i[1], i[2], i[6]
        flops = 2 * M * N * K
        byts = M * K * 2 + N * K * 2 + M * (N // 2 if form == L.RG_GEGLU else N) * 2 + (M * N * 2 if op.p[3] else 0)
        if form == L.RG_XATTN:   # + P [M][64] x VO^T [K][64]; x read once, out [M][K] written once
            flops += 2 * M * K * N
            byts = 2 * M * K * 2 + 2 * N * K * 2
        if form == L.RG_GEGLU and op.p[9]:   # the cross-attention prologue: scores + blend GEMMs, the updated rows written once
            flops += 2 * 2 * M * K * 64
            byts += M * K * 2 + 2 * 64 * K * 2
    elif k == L.OP_CONV3X3:
        B, H, W, C0, C1, N, subpix = (i[j] for j in range(7))
        Cin, par, T = C0 + C1, (4 if subpix else 1), (4 if subpix else 9)
        flops = 2 * B * H * W * N * T * Cin * par
        byts = B * H * W * Cin * 2 + par * (N * T * Cin * 2 + B * H * W * N * 2)
        if op.p[5]:
            byts += B * H * W * N * 2
    elif k == L.OP_CONV3X3_HEAD:
        B, H, W, C, co = (i[j] for j in range(5))
        flops = 2 * B * H * W * co * 9 * C
        byts = B * H * W * (C * 2 + co * 4)
    elif k == L.OP_FLASH_ATTN64:
        B, heads, T = i[0], i[1], i[2]
        flops = 4 * B * heads * T * T * 64
        byts = 4 * B * heads * T * 64 * 2
    elif k == L.OP_FLASH_ATTN512:
        B, T = i[0], i[1]
        flops = 4 * B * T * T * 512
        byts = 4 * B * T * 512 * 2
    elif k == L.OP_GN_STATS:
        byts = i[0] * i[1] * (i[2] + (i[9] if op.p[6] else 0)) * 2
    elif k == L.OP_GN_APPLY:
        byts = 2 * i[0] * i[1] * i[2] * 2
    elif k == L.OP_GN_SLAB:
        byts = (2 if op.p[2] else 1) * i[0] * i[1] * i[2] * 2
    elif k == L.OP_SOFTMAX_ROWS:
        byts = i[0] * i[1] * 4 + i[0] * i[3] * 2
    elif k == L.OP_SCHED_STEP:
        byts = (4 if op.p[2] else 3) * l[0] * 4
    elif k == L.OP_LINEAR_SMALL_M:
        flops = 2 * i[0] * i[1] * i[2]
        byts = i[1] * i[2] * 4
    elif k == L.OP_LATENT_1X1:
        byts = i[0] * (i[1] + i[2]) * i[3] * 4
    elif k == L.OP_IM2COL_SMALL:
        byts = i[0] * i[1] * i[2] * ((i[3] + i[4]) * 4 + i[5] * 2)
    elif k == L.OP_POST_NCHW:
        byts = i[0] * i[1] * (i[3] + (1 if i[4] == L.POST_DEPTH else i[2])) * 4
        if i[4] == L.POST_SCHED:   # reads x_t (and the LCM noise) as well
            byts += i[0] * i[1] * i[2] * 4 * (2 if op.p[2] else 1)
    elif k in (L.OP_ENS_DEPTH_STATS, L.OP_ENS_DEPTH_MEDIAN):
        byts = i[0] * l[0] * 4
    elif k == L.OP_ENS_DEPTH_NORM:
        byts = 2 * l[0] * 4
    elif k == L.OP_ENS_NORMALS:
        byts = (i[0] + 1) * 3 * l[0] * 4
    elif k in (L.OP_MEMSET, L.OP_COPY):
        byts = l[0]
    return cls, flops, byts


def summarize(ops, ms):
    """Aggregate per-op (cost, time) into {class: {launches, ms, flops, bytes, tflops, gbs}}."""
    out = {}
    for op, t in zip(ops, ms):
        cls, f, b = op_cost(op)
        d = out.setdefault(cls, dict(launches=0, ms=0.0, flops=0, bytes=0))
        d["launches"] += 1
        d["ms"] += float(t)
        d["flops"] += f
        d["bytes"] += b
    for d in out.values():
        s = max(d["ms"], 1e-9) * 1e-3
        d["tflops"] = d["flops"] / s / 1e12
        d["gbs"] = d["bytes"] / s / 1e9
    return out


def program_flops(ops):
    return sum(op_cost(op)[1] for op in ops)
