"""Checkpoint -> engine weight layouts (done once at load time, on the host in fp32).

diffusers state-dict tensors (``arch.py`` key scheme) are re-laid for the HIP kernels:
  * conv3x3 [Cout,Cin,3,3] -> bf16 [Cout][(ky*3+kx)*Cin + c]   (K-contiguous rows for igemm)
  * Linear / conv1x1        -> bf16 [N][K]
  * GEGLU proj [8C, C]      -> rows interleaved in 32-row groups (16 "u" rows then their 16 gate
                               rows) so the GEMM epilogue forms u*gelu(g) in registers
  * cross-attention against the constant 2-token empty-prompt embedding
    (marigold_depth_pipeline.py:381-394) -> Wqk [64-padded 2*heads][C], VO^T [C][64]
  * fused Q|K|V projection  -> [3C][C]
"""
import torch


def to_op16(t, dtype=torch.bfloat16):
    """fp32 -> the engine's 16-bit operand type (bf16, or fp16 for the fp16 build - saturating like its kernels' epilogues)."""
    t = t.detach().float()
    if dtype == torch.float16:
        t = t.clamp(-65504.0, 65504.0)
    return t.to(dtype)


def bf16(t, device, dtype=torch.bfloat16):
    """Kernel-ready 16-bit operands on the device (``dtype``: bf16, the product type, or fp16)."""
    return to_op16(t, dtype).to(device=device).contiguous()


def f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def pack_conv3x3(w):
    """[Cout,Cin,3,3] -> [Cout, 9*Cin] with k = (ky*3+kx)*Cin + c."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()


def pack_conv3x3_subpix(w):
    """Nearest-2x up-sampling followed by conv3x3 (diffusers Upsample2D) as four 2x2 convolutions on the
    low-resolution input, one per output parity (a, b): output row 2y+a reads virtual rows 2y+a+dy-1, i.e. source rows
    {y-1, y, y} (a = 0) or {y, y, y+1} (a = 1) for dy = 0, 1, 2 - taps that hit the same source pixel are summed here,
    in fp32, before the bf16 rounding.  [Cout,Cin,3,3] -> [4 (z = 2a+b)][Cout][(ty*2+tx)*Cin + c]; 4/9 of the MACs."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    sets = (((0,), (1, 2)), ((0, 1), (2,)))   # sets[a][ty] = the dy that land on source row y - 1 + a + ty
    out = w.new_zeros(4, co, 2, 2, ci)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    acc = 0
                    for dy in sets[a][ty]:
                        for dx in sets[b][tx]:
                            acc = acc + w[:, :, dy, dx]
                    out[2 * a + b, :, ty, tx, :] = acc
    return out.reshape(4, co, 4 * ci).contiguous()


def pack_conv1x1(w):
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def pack_geglu(w, b, group=None):
    """proj [8C, C] (rows 0..4C-1 = u, 4C..8C-1 = gate) -> rows interleaved in groups of ``group``:
    group/2 consecutive u rows followed by their group/2 gate rows.  ``group`` = 32 is what the GEMM
    epilogue expects (mg_geglu_interleave)."""
    if group is None:
        group = 32
    n2 = w.shape[0]
    h = n2 // 2
    g2 = group // 2
    assert group == 32 and h % g2 == 0
    idx = torch.arange(n2)
    blk, within = idx // group, idx % group
    src = torch.where(within < g2, blk * g2 + within, h + blk * g2 + (within - g2))
    return w[src].contiguous(), b[src].contiguous()


def pack_qkv(wq, wk, wv):
    return torch.cat([wq, wk, wv], dim=0).contiguous()


def cross_attention_tables(wq, wk, wv, wo, ctx, heads):
    """Collapse attn2 (2 context tokens) into two thin matrices (fp32 math on the host).

    wq [C,C], wk/wv [C,cross], wo [C,C], ctx [2,cross].
    Returns Wqk [npad, C] (row 2h+j = Wq_h^T k_{j,h}; zero rows beyond 2*heads) and
            VOt [C, npad] (column 2h+j = Wo[:, h-slice] @ v_{j,h}).
    """
    C = wq.shape[0]
    dh = C // heads
    wq, wk, wv, wo, ctx = (t.double() for t in (wq, wk, wv, wo, ctx))
    k = ctx @ wk.t()           # [2, C]
    v = ctx @ wv.t()           # [2, C]
    npad = ((2 * heads + 63) // 64) * 64
    wqk = torch.zeros(npad, C, dtype=torch.float64)
    vot = torch.zeros(C, npad, dtype=torch.float64)
    for h in range(heads):
        sl = slice(h * dh, (h + 1) * dh)
        for j in range(2):
            wqk[2 * h + j] = k[j, sl] @ wq[sl, :]          # (Wq_h^T k_jh)[c] = sum_d Wq[hd,c] k[j,hd]
            vot[:, 2 * h + j] = wo[:, sl] @ v[j, sl]
    return wqk.float(), vot.float(), npad


def fold_layernorm(w, bias, gamma, beta, dtype=torch.bfloat16):
    """LayerNorm folded into the Linear layer that consumes it (MG_OP_IGEMM ln_in): y = LN(x) W^T + b
    = rstd * (x (W gamma)^T - mean * g) + c  with  g[n] = sum_k (W gamma)[n][k],  c[n] = sum_k beta[k] W[n][k] + b[n].
    ``w`` [N][K] fp32 (rows already in the kernel's order), ``bias`` [N] | None.  Returns (W gamma as bf16, g, c); g sums
    the ROUNDED weights, so for a row x = a (constant) the product x (W gamma)^T cancels against mean g up to the row
    statistics' own rounding: (mean, rstd) are taken by the PRODUCING GEMM's epilogue from its fp32 values BEFORE they are
    rounded to the bf16 row the consumer reads (igemm2_body.h, ln_out), i.e. they are the statistics of x, not of bf16(x) -
    a deviation of <= 2^-9 |x| per element from LayerNorm(bf16(x)), below the bf16 resolution of the output (measured:
    tests/test_gpu_kernels.py::test_igemm_layernorm_fold, heavy-tailed rows included)."""
    wp = to_op16((w.double() * gamma.double()[None, :]).float(), dtype)
    g = wp.double().sum(dim=1).float()
    c = w.double() @ beta.double()
    if bias is not None:
        c = c + bias.double()
    return wp.contiguous(), g.contiguous(), c.float().contiguous()


# ---- MG_OP_ROWGEMM (csrc/rowgemm.hip): weights in MFMA fragment order, streamed in 64-channel stages ----
def _rg_chan():
    """MFMA row index mm -> channel inside a 32-channel tile: bits 2 and 3 exchanged, so that a lane's accumulator registers
    0-7 / 8-15 are 8 consecutive channels each (16-byte stores without a lane exchange)."""
    mm = torch.arange(32)
    return (mm & 0x13) | ((mm & 4) << 1) | ((mm & 8) >> 1)


def pack_rowgemm(w, cb, lg=None, dtype=torch.bfloat16):
    """w [N][K] (rows in the order the kernel's channels take: stage j = rows 64 j .. 64 j + 63, tile 0 its first 32) ->
    uint8 [slots][41 * 1024].  A slot of the kernel's LDS ring is 40 fragments of 1 KB + a 1 KB trailer; fragment (tile t, K step
    s), lane l = 32 g + mm, 8 bf16: w[64 j + 32 t + chan(mm)][16 s + 8 g + 0..7] (the MFMA's own K order).
      K = 320: one slot per stage - [tile 0: 20 fragments][tile 1: 20 fragments][trailer];
      K = 640: two slots per stage - [tile 0: 40 fragments][unused trailer], [tile 1: 40 fragments][trailer].
    Trailer: fp32 [64] per-channel constants ``cb`` (bias, + the folded LayerNorm's c), fp32 [64] ``lg`` (the folded LayerNorm's g;
    zeros without), zero padding."""
    n, k = w.shape
    assert n % 64 == 0 and k in (320, 640)
    nst, ks = n // 64, k // 16
    wb = to_op16(w.cpu(), dtype).contiguous()
    wv = wb.view(nst, 2, 32, ks, 2, 8)[:, :, _rg_chan()]          # [j][t][mm][s][g][i]
    frag = wv.permute(0, 1, 3, 4, 2, 5).contiguous()              # [j][t][s][g][mm][i] = 1 KB per (j, t, s)
    frag = frag.view(torch.uint8).reshape(nst, 2, ks * 1024)
    trl = torch.zeros(nst, 256, dtype=torch.float32)
    trl[:, :64] = cb.detach().float().cpu().reshape(nst, 64)
    if lg is not None:
        trl[:, 64:128] = lg.detach().float().cpu().reshape(nst, 64)
    trl = trl.view(torch.uint8).reshape(nst, 1024)
    if k == 320:
        return torch.cat([frag[:, 0], frag[:, 1], trl], dim=1).contiguous()
    zero = torch.zeros_like(trl)
    return torch.stack([torch.cat([frag[:, 0], zero], dim=1), torch.cat([frag[:, 1], trl], dim=1)], dim=1).reshape(2 * nst, 41 * 1024).contiguous()


def rowgemm_geglu_order(n2):
    """Row order of GEGLU's proj [8C, C] for MG_OP_ROWGEMM's GEGLU form: stage j = value rows 32 j .. 32 j + 31, then
    their gate rows 4C + 32 j .. (diffusers GEGLU: hidden, gate = proj(x).chunk(2))."""
    h = n2 // 2
    j = torch.arange(h // 32)[:, None]
    r = torch.arange(32)[None, :]
    return torch.cat([32 * j + r, h + 32 * j + r], dim=1).reshape(-1)


def pack_rowgemm_xattn(wqk, c, g, vot, bias, dtype=torch.bfloat16):
    """MG_OP_ROWGEMM form RG_XATTN: [scores stage = pack_rowgemm(Wqk with the LayerNorm folded in [64][K], c, g)] followed by the
    second GEMM's weights VO^T [c2][64] as c2/64 sub-stages of 2 tiles x 4 K steps (fragment (jj, t, s), lane 32 g + mm:
    vot[64 jj + 32 t + chan(mm)][16 s + 8 g .. + 8]) and 2 KB of fp32 bias [c2]."""
    c2 = vot.shape[0]
    assert wqk.shape[0] == 64 and vot.shape[1] == 64 and c2 % 64 == 0 and c2 * 4 <= 2048
    s0 = pack_rowgemm(wqk, c, g, dtype=dtype).reshape(-1)
    vb = to_op16(vot.cpu(), dtype).contiguous()
    fr = vb.view(c2 // 64, 2, 32, 4, 2, 8)[:, :, _rg_chan()].permute(0, 1, 3, 4, 2, 5).contiguous().view(torch.uint8).reshape(-1)
    tr = torch.zeros(512, dtype=torch.float32)
    tr[:c2] = bias.detach().float().cpu()
    return torch.cat([s0, fr, tr.view(torch.uint8)]).contiguous()


def pack_rowgemm_xattn_ksplit(wqk, c, g, vot, bias, dtype=torch.bfloat16):
    """MG_OP_ROWGEMM form RG_XATTN at K = c2 = 640 / 1280 (csrc/rowgemm.hip::rowgemm_xattn_ksplit_kernel): wave w of a 32-row
    workgroup owns the channels [K/4 w, K/4 (w + 1)) on both sides.  [score fragments [w][tile t][K step s][lane 32 g + mm][8]:
    wqk[32 t + chan(mm)][K/4 w + 16 s + 8 g + i]] [VO^T fragments [w][tile tt][step s'][lane][8]: vot[K/4 w + 32 tt + chan(mm)]
    [16 s' + 8 g + i]] [fp32 c[64], g[64] of the folded LayerNorm, bias[K]]."""
    k = wqk.shape[1]
    assert wqk.shape[0] == 64 and vot.shape == (k, 64) and k in (640, 1280)
    kq = k // 4
    ksq, nt2 = kq // 16, kq // 32
    ch = _rg_chan()
    wb = to_op16(wqk.cpu(), dtype).contiguous()
    f1 = wb.view(2, 32, 4, ksq, 2, 8)[:, ch].permute(2, 0, 3, 4, 1, 5).contiguous().view(torch.uint8).reshape(-1)     # [w][t][s][g][mm][i]
    vb = to_op16(vot.cpu(), dtype).contiguous()
    f2 = vb.view(4, nt2, 32, 4, 2, 8)[:, :, ch].permute(0, 1, 3, 4, 2, 5).contiguous().view(torch.uint8).reshape(-1)   # [w][tt][s'][g][mm][i]
    fl = torch.cat([c.detach().float().cpu().reshape(64), g.detach().float().cpu().reshape(64), bias.detach().float().cpu().reshape(k)])
    return torch.cat([f1, f2, fl.contiguous().view(torch.uint8)]).contiguous()
