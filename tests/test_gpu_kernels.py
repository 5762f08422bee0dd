"""Per-kernel parity on a real MI355X: every HIP op (through the C ABI, mg_launch) vs a plain
PyTorch fp32 CPU reference of the same op on the same (bf16-rounded) inputs.

Tolerances: bf16 storage (8 mantissa bits) with fp32 accumulation -> |err| <= 1.5e-2 * scale
of the output, where scale = max|ref|; fp32 kernels (scheduler, ensembling, latents) 1e-5.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# The 16-bit operand type under test: bf16 (libmarigold_hip.so, the product build) or - MARIGOLD_TEST_OPERANDS=fp16, set by
# tests/test_gpu_fp16.py, which runs this file a second time - IEEE fp16 (libmarigold_hip_f16.so: the same kernels, the reference's
# --fp16 arithmetic).  Inputs are rounded to that type, the references computed in fp32 from the rounded values.
import os
F16 = os.environ.get("MARIGOLD_TEST_OPERANDS", "bf16") == "fp16"
OP16 = torch.float16 if F16 else torch.bfloat16
if F16:
    import functools
    from marigold_amd import weights as _Wm
    for _n in ("fold_layernorm", "pack_rowgemm", "pack_rowgemm_xattn", "pack_rowgemm_xattn_ksplit", "bf16"):
        setattr(_Wm, _n, functools.partial(getattr(_Wm, _n), dtype=torch.float16))
bf16_only = pytest.mark.skipif(F16, reason="exercises the bf16 exponent range (fp16 operands top out at 65504) or a bf16-only kernel form")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from marigold_amd import _lib
    _lib.init(0, F16)
    return torch.device("cuda:0")


def _bf(x):
    return x.to(OP16).float()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _close(name, got, ref, tol=1.5e-2):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    print(f"[parity] {name}: max|err|={err:.3e} scale={scale:.3e} rel={err / scale:.3e}")
    assert err <= tol * scale, f"{name}: max|err| {err:.4e} > {tol} * {scale:.4e}"


def _run(op):
    from marigold_amd import _lib, ops
    ops.launch(op, lib=_lib.load(F16))
    torch.cuda.synchronize()


# --------------------------------------------------------------------------- igemm: conv
CONV_CASES = [
    # name, B, H, W, Cin, Cout, stride, pad, up, variant
    ("stride2_pad1", 2, 16, 16, 64, 64, 2, 1, None, 0),
    ("stride2_pad0_asym", 1, 16, 24, 128, 128, 2, 0, None, 0),
    ("stride2_odd", 1, 15, 11, 64, 64, 2, 1, None, 23),
    ("up2", 2, 6, 10, 64, 128, 1, 1, (12, 20), 0),
    ("up_to_size", 1, 4, 6, 64, 64, 1, 1, (7, 11), 0),
    ("auto_big", 4, 32, 32, 192, 256, 1, 1, None, 0),
    # every tile of the implicit-GEMM kernel a launch can name (igemm2.hip::dispatch_tile): LDS ring + counted vmcnt + 16-byte epilogue
    ("g2_256x128x3", 2, 16, 24, 128, 128, 1, 1, None, 36),
    ("g2_128x128x2", 2, 12, 20, 64, 128, 1, 1, None, 32),
    ("g2_128x128w8", 2, 12, 20, 64, 192, 1, 1, None, 22),
    ("g2_64x64", 1, 9, 7, 64, 64, 1, 1, None, 23),
    ("g2_256x64", 2, 16, 24, 128, 320, 1, 1, None, 35),
    # the deeper LDS rings (4 / 4 / 3 stages): short K (fewer tiles than stages), ragged M / N, stride 2, up-sampling
    ("g2_64x64_ring4", 1, 9, 7, 64, 64, 1, 1, None, 24),
    ("g2_64x64_ring4_kt1_stride2", 2, 16, 16, 64, 192, 2, 1, None, 24),
    ("g2_64x64_ring4_big_k", 1, 12, 12, 1280, 128, 1, 1, None, 24),
    ("g2_128x64_ring4", 2, 12, 20, 192, 64, 1, 1, None, 25),
    ("g2_128x64_ring4_up", 1, 4, 6, 64, 320, 1, 1, (7, 11), 25),
    ("g2_128x64_ring4_big_k_medge", 1, 13, 12, 1280, 256, 1, 1, None, 25),
    ("g2_128x128_ring3", 2, 12, 20, 320, 320, 1, 1, None, 26),
    ("g2_128x128_ring3_stride2_pad0", 1, 16, 24, 128, 128, 2, 0, None, 26),
    ("g2_128x128_ring3_big_k_medge", 1, 13, 12, 1280, 192, 1, 1, None, 26),
    ("g2_128x64", 2, 12, 20, 192, 64, 1, 1, None, 35),
    ("g2_256x128x2", 1, 24, 24, 64, 128, 1, 1, None, 36),
    ("g2_128x128x3", 2, 12, 20, 320, 320, 1, 1, None, 32),
    ("g2_kt1_stride2", 2, 16, 16, 64, 64, 2, 1, None, 36),
    ("g2_stride2_pad0", 1, 16, 24, 128, 128, 2, 0, None, 32),
    ("g2_up2", 2, 6, 10, 64, 128, 1, 1, (12, 20), 36),
    ("g2_up_to_size", 1, 4, 6, 64, 64, 1, 1, (7, 11), 23),
    ("g2_big_k", 1, 12, 12, 1280, 128, 1, 1, None, 36),
    ("g2_256x256", 2, 16, 24, 128, 320, 1, 1, None, 62),
    ("g2_split_256x128", 2, 16, 24, 128, 192, 1, 1, None, 36),
    ("g2_split_3stage", 2, 16, 24, 128, 128, 2, 1, None, 36),
    ("g2_split_256x256", 2, 16, 24, 192, 320, 1, 1, (32, 48), 62),
    ("g2_split_128x64", 1, 9, 7, 128, 64, 1, 1, None, 35),
    ("g2_pipe_128x64_kt1", 2, 12, 12, 64, 320, 2, 1, None, 35),
    ("g2_pipe_128x128", 1, 14, 14, 320, 128, 1, 1, None, 32),
    ("g2_pipe_128x128w8_up2", 2, 6, 10, 64, 128, 1, 1, (12, 20), 22),
    ("g2_auto_split_n320", 10, 72, 72, 64, 320, 1, 1, None, 0),   # 128k+64 columns: main + 64-wide tail launch
    ("g2_auto_small_m", 2, 12, 12, 128, 640, 1, 1, None, 0),
    ("g2_bk32_256x128_w4", 2, 16, 24, 128, 192, 1, 1, None, 51),
    ("g2_bk32_3stage_stride2", 2, 16, 16, 64, 128, 2, 1, None, 51),
    ("g2_bk32_w8_up2", 2, 6, 10, 64, 128, 1, 1, (12, 20), 51),
    ("g2_pingpong_kt9_stride2", 2, 16, 16, 64, 64, 2, 1, None, 62),
    ("g2_pingpong_big_k_medge", 1, 13, 12, 1280, 256, 1, 1, None, 62),   # 180 K tiles, M = 156 (one partial tile)
    ("g2_big_bk32_up_nedge", 2, 16, 24, 192, 320, 1, 1, (32, 48), 72),   # 128 x 128 wave tile (igemm2_big.hip), 4 stages of 32
    ("g2_big_bk32_kt9_stride2", 2, 16, 16, 64, 64, 2, 1, None, 72),
    ("g2_big_bk64_big_k_medge", 1, 13, 12, 1280, 256, 1, 1, None, 72),   # 2 stages of 64, M = 156 (one partial tile)
    # the hand-placed one-wave-per-SIMD K loop (buffer-load LDS-DMA: padding rows / M / N edges are out-of-range offsets)
    ("g2_k4w_up_to_size", 1, 4, 6, 64, 64, 1, 1, (7, 11), 72),
    ("g2_k4w_stride2_pad0", 1, 16, 24, 128, 128, 2, 0, None, 72),
    ("g2_k4wb_up_nedge", 2, 16, 24, 192, 640, 1, 1, (32, 48), 73),       # 192 x 320 tile: 27 K tiles over 9 taps, 16 M tiles
    ("g2_k4wb_kt9_stride2", 2, 16, 16, 64, 320, 2, 1, None, 73),
    ("g2_k4wb_big_k_medge", 1, 13, 12, 1280, 320, 1, 1, None, 73),       # 180 K tiles, M = 156 (one partial tile)
    ("g2_k4wb_nedge", 1, 14, 14, 128, 256, 1, 1, None, 73),              # N = 256 < 320: out-of-range weight rows
    ("g2_128x320", 2, 12, 20, 128, 320, 1, 1, None, 46),
    ("g2_256x320", 2, 12, 20, 64, 640, 1, 1, None, 46),
    ("g2_128x320_burst", 1, 12, 12, 320, 320, 1, 1, None, 46),
    ("g2_auto_splitk", 1, 12, 12, 1280, 128, 1, 1, None, 0),        # 6 tiles x 8 K-splits + reduce kernel
    ("g2_auto_splitk_n320", 1, 9, 7, 640, 320, 1, 1, None, 0),
    ("g2_auto_splitk_up2", 1, 6, 6, 1280, 256, 1, 1, (12, 12), 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_igemm_conv3x3(dev, case):
    from marigold_amd import ops, weights as Wm
    name, B, H, W, Cin, Cout, stride, pad, up, variant = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    xin = x
    if up:
        xin = F.interpolate(x, size=up, mode="nearest")
    if stride == 2 and pad == 0:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, bias, stride=2)
    else:
        ref = F.conv2d(xin, w, bias, stride=stride, padding=pad)
    Ho, Wo = ref.shape[-2:]
    temb = torch.randn(B, Cout, generator=g) * 0.2
    res = _bf(torch.randn(B, Cout, Ho, Wo, generator=g))
    ref = ref + temb[:, :, None, None] + res
    xd = _nhwc(x).to(dev, OP16)
    wd = Wm.pack_conv3x3(w).to(dev, OP16)
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=dev, dtype=OP16)
    op = ops.igemm(xd, wd, out, B=B, H=H, W=W, Cin=Cin, Ho=Ho, Wo=Wo, N=Cout, taps=9, stride=stride,
                   pad=pad, up=up, bias=bias.to(dev), rowvec=temb.to(dev).contiguous(),
                   residual=_nhwc(res).to(dev, OP16), variant=variant)
    _run(op)
    _close(f"conv3x3/{name}", out.float().permute(0, 3, 1, 2), ref)


# The dominant shapes of the benchmark configuration, at their real spatial size, against
# torch conv2d fp32 on the host cores: VAE decoder 128->128 @768^2, 512->512 @192^2 (the 256x256 ping-pong tile),
# UNet 320->320 @96^2 at B = 10 (the 128x320 full-width tile) and the stride-1 up-sampling convolution.
DOMINANT_CONVS = [
    ("vae_128_128_768sq_b2", 2, 768, 768, 128, 128, None),
    ("vae_512_512_192sq_b2", 2, 192, 192, 512, 512, None),
    ("unet_320_320_96sq_b10", 10, 96, 96, 320, 320, None),
    ("unet_up_640_640_48to96_b2", 2, 48, 48, 640, 640, (96, 96)),
]


@pytest.mark.parametrize("case", DOMINANT_CONVS, ids=[c[0] for c in DOMINANT_CONVS])
def test_igemm_conv3x3_dominant_shapes(dev, case):
    from marigold_amd import ops, weights as Wm
    name, B, H, W, Cin, Cout, up = case
    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(min(32, usable_cores()))
    g = torch.Generator().manual_seed(len(name))
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, size=up, mode="nearest") if up else x
    Ho, Wo = xin.shape[-2:]
    res = _bf(torch.randn(B, Cout, Ho, Wo, generator=g))
    ref = F.conv2d(xin, w, bias, padding=1) + res
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=dev, dtype=OP16)
    op = ops.igemm(_nhwc(x).to(dev, OP16), Wm.pack_conv3x3(w).to(dev, OP16), out, B=B, H=H, W=W,
                   Cin=Cin, Ho=Ho, Wo=Wo, N=Cout, taps=9, stride=1, pad=1, up=up, bias=bias.to(dev),
                   residual=_nhwc(res).to(dev, OP16))
    _run(op)
    _close(f"conv3x3/{name}", out.float().permute(0, 3, 1, 2), ref)


# conv2 + conv_shortcut of a ResNet block in one launch: the 1x1 convolution of the block's input (one tensor, or the
# decoder's [x | skip] pair) as extra K tiles after the nine taps (MG_OP_IGEMM p[12] / p[13]).
FOLD_CASES = [
    # name, B, H, W, Cin, Cout, Cx0, Cx1, variant, splits
    ("fold_auto_one_source", 2, 12, 12, 128, 128, 64, 0, 0, 0),
    ("fold_auto_two_sources", 1, 24, 24, 320, 320, 320, 320, 0, 0),
    ("fold_v36_two_sources_ragged", 1, 9, 7, 128, 192, 64, 128, 36, -1),
    ("fold_v32", 2, 12, 20, 64, 128, 128, 0, 32, -1),
    ("fold_v23", 1, 13, 12, 128, 128, 64, 64, 23, -1),
    ("fold_v46", 2, 12, 20, 128, 320, 192, 128, 46, -1),
    ("fold_v72", 1, 24, 24, 128, 256, 192, 64, 72, -1),
    ("fold_v73_nedge", 1, 14, 14, 128, 256, 64, 128, 73, -1),
    ("fold_v73_split3", 1, 24, 24, 640, 640, 640, 320, 73, 3),      # a split that starts inside the folded K range
    ("fold_v36_split_in_fold", 1, 12, 12, 64, 128, 512, 256, 36, 4),   # most of K is the folded part
    ("fold_v72_split2", 1, 12, 12, 1280, 256, 1280, 1280, 72, 2),
    ("fold_auto_split", 1, 12, 12, 1280, 1280, 1280, 1280, 0, 0),
    # the tile / split pairs the tuning table names for folded launches of small ensembles (marigold_amd/tuning/gfx950.json)
    ("fold_v22_split8", 1, 12, 12, 1280, 1280, 1280, 1280, 22, 8),
    ("fold_v22_split6", 2, 12, 12, 1280, 1280, 1280, 0, 22, 6),
    ("fold_v24_split8", 1, 12, 12, 1280, 1280, 1280, 1280, 24, 8),
    ("fold_v32_split8", 1, 24, 24, 1280, 1280, 640, 640, 32, 8),
    ("fold_v73_split16", 1, 24, 24, 1280, 640, 1280, 640, 73, 16),
]


@pytest.mark.parametrize("case", FOLD_CASES, ids=[c[0] for c in FOLD_CASES])
def test_igemm_conv3x3_folded_shortcut(dev, case):
    from marigold_amd import ops, weights as Wm
    name, B, H, W, Cin, Cout, Cx0, Cx1, variant, splits = case
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cx = Cx0 + Cx1
    h = _bf(torch.randn(B, Cin, H, W, generator=g))
    xs = _bf(torch.randn(B, Cx, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    wsc = _bf(torch.randn(Cout, Cx, 1, 1, generator=g) / math.sqrt(Cx))
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(h, w, bias, padding=1) + F.conv2d(xs, wsc)
    wd = torch.cat([Wm.pack_conv3x3(w), wsc.reshape(Cout, Cx)], dim=1).to(dev, OP16).contiguous()
    x0 = _nhwc(xs[:, :Cx0]).to(dev, OP16).contiguous()
    x1 = _nhwc(xs[:, Cx0:]).to(dev, OP16).contiguous() if Cx1 else None
    out = torch.full((B, H, W, Cout), float("nan"), device=dev, dtype=OP16)
    op = ops.igemm(_nhwc(h).to(dev, OP16), wd, out, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=Cout, taps=9, stride=1, pad=1,
                   bias=bias.to(dev), variant=variant, splits=splits, fold=(x0, x1, Cx0, Cx))
    _run(op)
    _close(f"conv3x3+shortcut/{name}", out.float().permute(0, 3, 1, 2), ref)
    # the same through strided sources (the shortcut input as a channel window of a wider tensor)
    wide = torch.full((B, H, W, Cx0 + 64), float("nan"), device=dev, dtype=OP16)
    wide[..., :Cx0] = x0
    out2 = torch.full_like(out, float("nan"))
    _run(ops.igemm(_nhwc(h).to(dev, OP16), wd, out2, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=Cout, taps=9, stride=1, pad=1,
                   bias=bias.to(dev), variant=variant, splits=splits, fold=(wide, x1, Cx0, Cx, Cx0 + 64, 0)))
    assert torch.equal(out, out2)


def test_igemm_fold_rejects_what_it_cannot_run(dev):
    from marigold_amd import ops, _lib as L
    x = torch.zeros(1, 8, 8, 64, device=dev, dtype=OP16)
    w = torch.zeros(64, 9 * 64 + 64, device=dev, dtype=OP16)
    out = torch.zeros(1, 8, 8, 64, device=dev, dtype=OP16)
    kw = dict(B=1, H=8, W=8, Cin=64, Ho=8, Wo=8, N=64, taps=9, stride=1, pad=1)
    for bad in (dict(fold=(x, None, 64, 96)),            # Cx not a multiple of the K tile
                dict(fold=(x, x, 64, 64)),               # a second source with nothing left for it
                dict(fold=(x, None, 64, 64), epi=L.EPI_F32)):
        with pytest.raises(L.MarigoldHipError):
            _run(ops.igemm(x, w, out, **{**kw, **bad}))


@pytest.mark.parametrize("B,H,W,Cin,Cout,variant", [(2, 6, 10, 64, 128, 0), (1, 16, 16, 192, 320, 0), (2, 24, 20, 128, 256, 62),
                                                     (1, 9, 7, 64, 64, 23), (2, 48, 48, 640, 640, 0)])
def test_igemm_subpixel_upsample_conv(dev, B, H, W, Cin, Cout, variant):
    """Nearest-2x + conv3x3 (diffusers Upsample2D) in its sub-pixel form: four 2x2 convolutions on the low-resolution
    input with taps pre-summed in the weights (weights.pack_conv3x3_subpix), vs conv2d on the up-sampled input."""
    from marigold_amd import ops, weights as Wm
    g = torch.Generator().manual_seed(H * 100 + W)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device=dev, dtype=OP16)
    ws = Wm.pack_conv3x3_subpix(w).to(dev, OP16)
    _run(ops.igemm(_nhwc(x).to(dev, OP16), ws, out, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=Cout, taps=4, stride=1,
                   pad=1, bias=bias.to(dev), batch_z=4, zstrides=(0, Cout * 4 * Cin, 0, 0), variant=variant))
    # the pre-summed taps are rounded to bf16 once more than the reference's weights: 2e-2 instead of 1.5e-2
    _close(f"subpixel up-conv B{B} {H}x{W} {Cin}->{Cout} v{variant}", out.float().permute(0, 3, 1, 2), ref, tol=2e-2)


# --------------------------------------------------------------------------- patch-resident conv3x3 (MG_OP_CONV3X3)
def _ref_fused_conv(x, w, bias, ss, silu, temb, res, up2=False):
    """torch fp32 reference of the fused chain on bf16-rounded inputs: (GroupNorm affine [+ SiLU], rounded to bf16 like
    gn_apply's output) -> [nearest 2x] -> conv3x3 pad 1 -> + bias + temb + residual."""
    h = x
    if ss is not None:
        sc, sh = ss[:, 0][:, :, None, None], ss[:, 1][:, :, None, None]
        h = h * sc + sh
        if silu:
            h = h * torch.sigmoid(h)
        h = _bf(h)
    if up2:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    out = F.conv2d(h, w, bias, padding=1)
    if temb is not None:
        out = out + temb[:, :, None, None]
    if res is not None:
        out = out + res
    return out


PATCH_CASES = [
    # name, B, H, W, C0, C1, N, fused-norm, silu, temb, residual, variant
    ("v5_small", 1, 8, 16, 64, 0, 128, True, True, False, False, 5),
    ("auto_n640", 1, 16, 16, 64, 0, 640, False, False, False, False, 0),
    ("deep_k_fused", 1, 16, 16, 1280, 640, 256, True, True, True, True, 1),
    # 12 x 16 pixels x 320 channels (12 / 6 waves; the weight stage does not divide evenly over the threads): whole and
    # ragged tiles, fused norm, second source, time embedding, residual
    ("v6_320_fused", 2, 24, 32, 320, 0, 320, True, True, True, True, 6),
    ("v6_320_ragged_concat", 1, 31, 21, 320, 320, 320, True, True, False, True, 6),
    ("v6_320_plain", 1, 12, 16, 64, 0, 640, False, False, False, False, 6),
    ("v7_320_fused", 1, 20, 24, 320, 0, 320, True, True, True, False, 7),
    # 24 x 16 pixels x 128 channels / 12 waves (the VAE's 128-channel level)
    ("v8_128_fused", 1, 48, 32, 256, 0, 128, True, True, False, True, 8),
    ("v8_128_ragged", 2, 29, 19, 128, 0, 256, False, False, True, False, 8),
    ("v9_256_fused", 1, 36, 32, 128, 128, 512, True, True, True, True, 9),
    ("v9_256_ragged", 1, 13, 17, 64, 0, 256, False, False, False, False, 9),
    # four waves, one per SIMD, hand-placed streams (conv_patch4w.hip): 16 x 16 x 256 (variant 10) and 12 x 16 x 320 (11) -
    # whole / ragged tiles, one and several channel tiles (patch double buffer, in-stream fix-up), second source, epilogue terms
    ("v10_256_plain_1tile", 1, 16, 16, 64, 0, 256, False, False, False, False, 10),
    ("v10_256_plain", 2, 32, 32, 192, 0, 512, False, False, True, True, 10),
    ("v10_256_fused", 1, 36, 32, 128, 128, 512, True, True, True, True, 10),
    ("v10_256_fused_ragged", 2, 13, 17, 320, 0, 256, True, True, False, True, 10),
    ("v10_256_fused_1tile", 1, 16, 16, 64, 0, 256, True, True, False, False, 10),
    ("v11_320_plain_1tile", 1, 12, 16, 64, 0, 320, False, False, False, False, 11),
    ("v11_320_plain_concat", 1, 31, 21, 320, 320, 640, False, False, True, True, 11),
    ("v11_320_fused", 2, 24, 32, 320, 0, 320, True, True, True, True, 11),
    ("v11_320_fused_ragged_concat", 1, 31, 21, 320, 320, 320, True, True, False, True, 11),
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
def test_conv3x3_patch(dev, case):
    from marigold_amd import ops, weights as Wm
    name, B, H, W, C0, C1, N, fused, silu, use_temb, use_res, variant = case
    if F16 and fused and variant in (10, 11):
        pytest.skip("the four-wave tiles' in-stream GroupNorm fix-up unpacks bf16: the fp16 build runs fused norms on the 12-wave tiles")
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cin = C0 + C1
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(N, generator=g) * 0.1
    ss = torch.stack([1.0 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], dim=1) if fused else None
    temb = torch.randn(B, N, generator=g) * 0.2 if use_temb else None
    res = _bf(torch.randn(B, N, H, W, generator=g)) if use_res else None
    ref = _ref_fused_conv(x, w, bias, ss, silu, temb, res)
    xh = _nhwc(x)
    a0 = xh[..., :C0].contiguous().to(dev, OP16)
    a1 = xh[..., C0:].contiguous().to(dev, OP16) if C1 else None
    out = torch.full((B, H, W, N), float("nan"), device=dev, dtype=OP16)
    # the op holds raw device pointers: keep every operand alive for both launches
    wd, bd = Wm.pack_conv3x3(w).to(dev, OP16), bias.to(dev)
    ssd = None if ss is None else ss.to(dev).contiguous()
    td = None if temb is None else temb.to(dev).contiguous()
    rd = None if res is None else _nhwc(res).to(dev, OP16)
    op = ops.conv3x3(a0, wd, out, B=B, H=H, W=W, C0=C0, N=N, a1=a1, C1=C1, ss=ssd, silu=silu, bias=bd, rowvec=td, residual=rd,
                     variant=variant)
    _run(op)
    first = out.clone()
    _close(f"conv3x3p/{name}", out.float().permute(0, 3, 1, 2), ref)
    out.fill_(float("nan"))
    _run(op)
    assert torch.equal(first, out), "bit-repeatable"


@pytest.mark.parametrize("B,H,W,Cin,N,variant", [(2, 16, 16, 64, 256, 1), (1, 9, 21, 192, 128, 2), (2, 24, 24, 320, 320, 3),
                                                 (2, 16, 16, 64, 256, 10), (1, 19, 21, 192, 512, 10), (2, 24, 24, 320, 320, 11), (1, 13, 9, 128, 640, 11),
                                                 (1, 48, 48, 640, 640, 0), (1, 25, 19, 128, 640, 6), (1, 29, 20, 64, 128, 8)])
def test_conv3x3_patch_subpixel(dev, B, H, W, Cin, N, variant):
    from marigold_amd import ops, weights as Wm
    g = torch.Generator().manual_seed(H + W + Cin)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(N, generator=g) * 0.1
    ref = _ref_fused_conv(x, w, bias, None, False, None, None, up2=True)
    out = torch.full((B, 2 * H, 2 * W, N), float("nan"), device=dev, dtype=OP16)
    _run(ops.conv3x3(_nhwc(x).to(dev, OP16), Wm.pack_conv3x3_subpix(w).to(dev, OP16), out, B=B, H=H, W=W,
                     C0=Cin, N=N, subpix=True, bias=bias.to(dev), wz=N * 4 * Cin, variant=variant))
    _close(f"conv3x3p/subpixel B{B} {H}x{W} {Cin}->{N} v{variant}", out.float().permute(0, 3, 1, 2), ref, tol=2e-2)


@pytest.mark.parametrize("B,H,W,Cin,N,variant,fused,use_res,subpix", [
    (2, 48, 32, 128, 128, 8, True, True, False),     # 24 x 16 x 128 tiles: 4-channel groups (two per 8-channel vector)
    (1, 50, 37, 64, 128, 8, False, False, False),    # ragged tiles: the pixels outside the image stay out of the sums
    (2, 24, 32, 128, 256, 9, True, True, False),     # 12 x 16 x 256 tiles: 8-channel groups
    (1, 24, 16, 64, 512, 9, False, False, False),    # two channel tiles, 16-channel groups (both halves of a lane pair)
    (1, 13, 21, 64, 256, 9, False, False, True),     # the sub-pixel up-sampling: four parities = four slots per tile
    (2, 18, 33, 128, 256, 1, False, False, True),    # ... on the 16 x 16 x 256 / 8-wave tile the VAE's up-sampling runs on
    (1, 32, 16, 64, 256, 1, True, True, False),
])
def test_conv3x3_patch_output_groupnorm_statistics(dev, B, H, W, Cin, N, variant, fused, use_res, subpix):
    """MG_OP_CONV3X3 p[8]: the GroupNorm partial sums of the convolution's OUTPUT as a by-product of the 12-wave tiles'
    epilogue, reduced by MG_OP_GN_FINALIZE, against (a) MG_OP_GN_STATS + FINALIZE over the stored tensor and (b) torch's
    statistics of it; the convolution's output itself must be bit-identical with and without the by-product, and the table
    bit-stable across launches."""
    from marigold_amd import ops, weights as Wm
    g = torch.Generator().manual_seed(H * W + N)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = (torch.randn(N, generator=g) * 0.5 + 0.3).to(dev)
    ss_in = (torch.stack([1.0 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], dim=1).to(dev).contiguous()
             if fused else None)
    Ho, Wo = (2 * H, 2 * W) if subpix else (H, W)
    res = torch.randn(B, Ho, Wo, N, generator=g).to(dev, OP16) if use_res else None
    xd = _nhwc(x).to(dev, OP16)
    wd = (Wm.pack_conv3x3_subpix(w) if subpix else Wm.pack_conv3x3(w)).to(dev, OP16)
    kw = dict(B=B, H=H, W=W, C0=Cin, N=N, ss=ss_in, silu=fused, bias=bias, residual=res, subpix=subpix, wz=N * 4 * Cin if subpix else 0,
              variant=variant)
    plain = torch.full((B, Ho, Wo, N), float("nan"), device=dev, dtype=OP16)
    _run(ops.conv3x3(xd, wd, plain, **kw))
    slots = ops.conv3x3_gn_slots(ops.conv3x3(xd, wd, plain, **kw))
    assert slots > 0
    cpg = N // 32
    gamma, beta = (1 + 0.2 * torch.randn(N, generator=g)).to(dev), (0.2 * torch.randn(N, generator=g)).to(dev)
    HW = Ho * Wo
    tables = []
    for rep in range(2):
        out = torch.full_like(plain, float("nan"))
        part = torch.full((B, slots, 32, 2), float("nan"), device=dev)
        _run(ops.conv3x3(xd, wd, out, gn_part=part, gn_cpg=cpg, gn_slots=slots, **kw))
        assert torch.equal(out, plain), "the by-product must not change the convolution's output"
        tables.append(part.clone())
    assert torch.equal(tables[0], tables[1]), "partial table not bit-stable"
    ss = torch.full((B, 2, N), float("nan"), device=dev)
    _run(ops.gn_finalize(tables[0], gamma, beta, ss, B=B, C=N, groups=32, slots=slots, HW=HW, eps=1e-6))
    # (a) the statistics pass over the stored tensor
    chunks = 8
    part2 = torch.empty(B, chunks, 32, 2, device=dev)
    ss2 = torch.full((B, 2, N), float("nan"), device=dev)
    _run(ops.gn_stats(plain, part2, B=B, HW=HW, C=N, chunks=chunks, groups=32))
    _run(ops.gn_finalize(part2, gamma, beta, ss2, B=B, C=N, groups=32, slots=chunks, HW=HW, eps=1e-6))
    _close(f"conv3x3p/gn by-product vs statistics pass/N{N}v{variant}", ss, ss2, tol=1e-5)
    # (b) torch on the stored values
    o = plain.float().cpu().reshape(B, HW, 32, cpg)
    mean, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    sc_ref = rstd.repeat_interleave(cpg, 1) * gamma.cpu()
    sh_ref = beta.cpu() - mean.repeat_interleave(cpg, 1) * sc_ref
    _close(f"conv3x3p/gn by-product scale/N{N}v{variant}", ss[:, 0], sc_ref, tol=2e-4)
    _close(f"conv3x3p/gn by-product shift/N{N}v{variant}", ss[:, 1], sh_ref, tol=2e-4)


def test_conv3x3_patch_equals_unfused_chain(dev):
    """The fused kernel feeds the MFMAs the same operands as gn_apply -> concat -> implicit GEMM (same fp32 affine +
    SiLU, same bf16 rounding of the normalised activation); only the fp32 accumulation order differs (channel tile
    outermost here, tap outermost there), so the outputs agree to one bf16 rounding of the result."""
    from marigold_amd import ops, weights as Wm
    B, H, W, C0, C1, N = 2, 32, 32, 128, 64, 256
    Cin = C0 + C1
    g = torch.Generator().manual_seed(5)
    a0 = torch.randn(B, H, W, C0, generator=g).to(dev, OP16)
    a1 = torch.randn(B, H, W, C1, generator=g).to(dev, OP16)
    w = Wm.pack_conv3x3(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev, OP16)
    ss = torch.stack([1.0 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], dim=1).to(dev).contiguous()
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    h = torch.empty(B, H, W, Cin, device=dev, dtype=OP16)
    _run(ops.gn_apply(a0, ss, h, B=B, HW=H * W, C=Cin, silu=True, x1=a1, C0=C0))
    ref = torch.empty(B, H, W, N, device=dev, dtype=OP16)
    _run(ops.igemm(h, w, ref, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=N, taps=9, stride=1, pad=1, bias=bias))
    out = torch.empty_like(ref)
    _run(ops.conv3x3(a0, w, out, B=B, H=H, W=W, C0=C0, N=N, a1=a1, C1=C1, ss=ss, silu=True, bias=bias))
    d = (out.float() - ref.float()).abs().max().item()
    scale = ref.float().abs().max().item()
    frac = (out != ref).float().mean().item()
    print(f"[parity] fused conv3x3p vs gn_apply+concat+igemm: max|diff| {d:.3e} (scale {scale:.2f}), {100 * frac:.2f} % of outputs differ")
    assert d <= scale / 128 and frac < 0.2


def test_igemm_two_sources_and_gn_channel_windows(dev):
    """conv_shortcut over the un-materialised skip concat (1x1 and 3x3, two A sources) and GroupNorm statistics /
    apply over two sources."""
    from marigold_amd import ops, weights as Wm
    B, H, W, C0, C1, N = 2, 12, 20, 128, 192, 320
    Cin = C0 + C1
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.3)
    xh = _nhwc(x)
    a0 = xh[..., :C0].contiguous().to(dev, OP16)
    a1 = xh[..., C0:].contiguous().to(dev, OP16)
    for taps, variant in ((1, 0), (9, 0), (9, 62), (1, 51), (9, 36), (9, 72), (1, 72), (9, 73), (1, 73)):
        w = _bf(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(taps * Cin)) if taps == 9 else _bf(torch.randn(N, Cin, 1, 1, generator=g) / math.sqrt(Cin))
        ref = F.conv2d(x, w, None, padding=1 if taps == 9 else 0)
        wd = (Wm.pack_conv3x3(w) if taps == 9 else w.reshape(N, Cin)).to(dev, OP16)
        out = torch.full((B, H, W, N), float("nan"), device=dev, dtype=OP16)
        _run(ops.igemm(a0, wd, out, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=N, taps=taps, stride=1, pad=1 if taps == 9 else 0,
                       a1=a1, C0=C0, variant=variant))
        _close(f"igemm two sources taps={taps} v{variant}", out.float().permute(0, 3, 1, 2), ref)
    # GroupNorm over the concat: statistics per source into one partial table, finalize, two-source apply
    groups, chunks, eps = 32, 4, 1e-5
    gamma, beta = (1.0 + 0.1 * torch.randn(Cin, generator=g)), 0.1 * torch.randn(Cin, generator=g)
    part = torch.zeros(B * 2 * chunks * groups * 2, device=dev)
    ss = torch.zeros(B, 2, Cin, device=dev)
    _run(ops.gn_stats(a0, part, B=B, HW=H * W, C=C0, chunks=chunks, groups=groups, Ctot=Cin, coff=0, slot0=0, slots=2 * chunks))
    _run(ops.gn_stats(a1, part, B=B, HW=H * W, C=C1, chunks=chunks, groups=groups, Ctot=Cin, coff=C0, slot0=chunks, slots=2 * chunks))
    _run(ops.gn_finalize(part, gamma.to(dev), beta.to(dev), ss, B=B, C=Cin, groups=groups, slots=2 * chunks, HW=H * W, eps=eps))
    out = torch.full((B, H, W, Cin), float("nan"), device=dev, dtype=OP16)
    _run(ops.gn_apply(a0, ss, out, B=B, HW=H * W, C=Cin, silu=True, x1=a1, C0=C0))
    ref = F.silu(F.group_norm(x, groups, gamma, beta, eps))
    _close("groupnorm over two sources", out.float().permute(0, 3, 1, 2), ref)
    # the same table finalized by the last-arriving statistics block (no finalize launch): same scale / shift up to
    # the summation order of the fp64 reduction, identical bits from launch to launch, counters left at zero
    gd, bd = gamma.to(dev), beta.to(dev)
    cnt = torch.zeros(1024, dtype=torch.int32, device=dev)
    runs = []
    for _ in range(3):
        part.fill_(float("nan"))
        ss2 = torch.full((B, 2, Cin), float("nan"), device=dev)
        for k, (src, C, coff) in enumerate(((a0, C0, 0), (a1, C1, C0))):
            _run(ops.gn_stats(src, part, B=B, HW=H * W, C=C, chunks=chunks, groups=groups, Ctot=Cin, coff=coff, slot0=k * chunks,
                              slots=2 * chunks, gamma=gd, beta=bd, ss=ss2, counters=cnt, eps=eps))
        runs.append(ss2.clone())
        assert int(cnt.abs().sum()) == 0
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    _close("groupnorm fused finalize vs finalize launch", runs[0], ss, tol=1e-5)
    # both sources in ONE statistics launch (p[6]: what the engine emits for a skip concat): the same partial table and the same
    # scale / shift, bit for bit - a block's arithmetic does not depend on which launch it belongs to
    part_two = part.clone()
    for C0_, C1_, s0, s1 in ((C0, C1, a0, a1), (C1, C0, a1, a0)):   # (the wider source first / second: the LDS size follows the max)
        if s0 is a1:   # swapped order = another concat: its own two-launch reference
            part_two.fill_(float("nan"))
            ss_ref = torch.full((B, 2, Cin), float("nan"), device=dev)
            for k, (src, C, coff) in enumerate(((s0, C0_, 0), (s1, C1_, C0_))):
                _run(ops.gn_stats(src, part_two, B=B, HW=H * W, C=C, chunks=chunks, groups=groups, Ctot=Cin, coff=coff, slot0=k * chunks,
                                  slots=2 * chunks, gamma=gd, beta=bd, ss=ss_ref, counters=cnt, eps=eps))
        else:
            ss_ref = runs[0]
        part1 = torch.full_like(part, float("nan"))
        ss1 = torch.full((B, 2, Cin), float("nan"), device=dev)
        _run(ops.gn_stats(s0, part1, B=B, HW=H * W, C=C0_, chunks=chunks, groups=groups, Ctot=Cin, coff=0, slot0=0, slots=2 * chunks,
                          gamma=gd, beta=bd, ss=ss1, counters=cnt, eps=eps, x1=s1, C1=C1_))
        assert int(cnt.abs().sum()) == 0
        assert torch.equal(part1, part_two) and torch.equal(ss1, ss_ref)


def test_flash_attn64_benchmark_shape(dev):
    """The level-0 self-attention of the 768^2 map: 5 heads x 9216 tokens (96^2 latent), vs CPU SDPA fp32."""
    from marigold_amd import ops
    B, heads, T = 1, 5, 9216
    C = heads * 64
    g = torch.Generator().manual_seed(11)
    qkv = _bf(torch.randn(B, T, 3 * C, generator=g))
    q, k, v = qkv.split(C, dim=-1)
    qh, kh, vh = (t.reshape(B, T, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, T, C)
    qkd = qkv.to(dev, OP16)
    vt = v.permute(0, 2, 1).contiguous().to(dev, OP16)
    out = torch.full((B, T, C), float("nan"), device=dev, dtype=OP16)
    _run(ops.flash_attn64(qkd, qkd[:, :, C:], vt, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T,
                          sq=T * 3 * C, sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125))
    _close("flash_attn64/5 heads x 9216 tokens", out, ref)


def test_igemm_linear_geglu_f32_trans_batched(dev):
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(5)
    # --- plain linear with M / N edges, and every tile variant ---
    M, K, N = 300, 192, 320
    x = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    ref = x @ w.t() + b
    for variant in (0, 22, 23, 24, 25, 26, 32, 35, 36, 46, 51, 62, 72, 73):
        out = torch.full((M, N), float("nan"), device=dev, dtype=OP16)
        _run(ops.linear(x.to(dev, OP16), w.to(dev, OP16), out, M=M, K=K, N=N,
                        bias=b.to(dev), variant=variant))
        _close(f"linear/v{variant}", out, ref)
    # --- GEGLU epilogue ---
    C = 64
    wg = _bf(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    bg = torch.randn(8 * C, generator=g) * 0.1
    xg = _bf(torch.randn(200, C, generator=g))
    u, gt = (xg @ wg.t() + bg).chunk(2, dim=-1)
    ref = u * F.gelu(gt)
    for group, variants in ((32, (22, 23, 24, 25, 26, 32, 35, 36, 51, 62, 72, 73, 0)),):
        wp, bp = Wm.pack_geglu(wg, bg, group)
        for variant in variants:
            out = torch.full((200, 4 * C), float("nan"), device=dev, dtype=OP16)
            _run(ops.linear(xg.to(dev, OP16), wp.to(dev, OP16), out, M=200, K=C,
                            N=8 * C, bias=bp.to(dev), epi=L.EPI_GEGLU, variant=variant))
            _close(f"geglu/v{variant}", out, ref)
    # --- fused QKV with transposed V section (+ bias), per-image token blocks ---
    B, T, C = 2, 144, 128
    xq = _bf(torch.randn(B * T, C, generator=g))
    wqkv = _bf(torch.randn(3 * C, C, generator=g) / math.sqrt(C))
    bq = torch.randn(3 * C, generator=g) * 0.1
    refq = xq @ wqkv.t() + bq
    ldt = 192
    qk = torch.full((B * T, 3 * C), float("nan"), device=dev, dtype=OP16)
    vt = torch.zeros((B, C, ldt), device=dev, dtype=OP16)
    refv = refq[:, 2 * C:].reshape(B, T, C).permute(0, 2, 1)
    for variant in (0, 23, 24, 25, 26, 32, 35, 62):
        qk.fill_(float("nan"))
        vt.zero_()
        _run(ops.igemm(xq.to(dev, OP16), wqkv.to(dev, OP16), qk, B=B, H=T, W=1, Cin=C,
                       Ho=T, Wo=1, N=3 * C, bias=bq.to(dev), out2=vt, trans_from=2 * C, ldt=ldt, variant=variant))
        _close(f"qkv/qk/v{variant}", qk[:, :2 * C], refq[:, :2 * C])
        _close(f"qkv/vt/v{variant}", vt[:, :, :T], refv)
        assert (vt[:, :, T:] == 0).all()
        # the same section with its tokens in accumulator order inside groups of 16 (flash_attn64 generation 3's V^T)
        vt.zero_()
        _run(ops.igemm(xq.to(dev, OP16), wqkv.to(dev, OP16), qk, B=B, H=T, W=1, Cin=C,
                       Ho=T, Wo=1, N=3 * C, bias=bq.to(dev), out2=vt, trans_from=2 * C, ldt=ldt, variant=variant, trans_perm=True))
        _close(f"qkv/vt_perm/v{variant}", vt[:, :, :T], ops.permute_vt_keys(refv.contiguous()))
        assert (vt[:, :, T:] == 0).all()
    # token count not a multiple of 8 (scalar transposed tail)
    T2 = 36
    xq2 = xq[:B * T2]
    vt2 = torch.zeros((B, C, 64), device=dev, dtype=OP16)
    qk2 = torch.full((B * T2, 3 * C), float("nan"), device=dev, dtype=OP16)
    _run(ops.igemm(xq2.to(dev, OP16), wqkv.to(dev, OP16), qk2, B=B, H=T2, W=1, Cin=C,
                   Ho=T2, Wo=1, N=3 * C, bias=bq.to(dev), out2=vt2, trans_from=2 * C, ldt=64, variant=23))
    _close("qkv/vt/odd_tokens", vt2[:, :, :T2], refq[:B * T2, 2 * C:].reshape(B, T2, C).permute(0, 2, 1))
    # --- batched fp32 scores: S_z = scale * Q_z K_z^T with strided operands ---
    Z, T, D = 3, 160, 128
    qkv = _bf(torch.randn(Z, T, 3 * D, generator=g))
    refS = torch.einsum("ztd,zsd->zts", qkv[..., :D], qkv[..., D:2 * D]) * 0.25
    qd = qkv.to(dev, OP16)
    S = torch.full((Z, T, T), float("nan"), device=dev, dtype=torch.float32)
    for variant in (0, 23, 32, 62):
        S.fill_(float("nan"))
        _run(ops.igemm(qd, qd[:, :, D:], S, B=1, H=T, W=1, Cin=D, Ho=T, Wo=1, N=T, epi=L.EPI_F32, ldo=T,
                       lda=3 * D, ldw=3 * D, batch_z=Z, zstrides=(T * 3 * D, T * 3 * D, T * T, 0),
                       scale=0.25, variant=variant))
        _close(f"scores_f32/v{variant}", S, refS, tol=2e-3)


def test_igemm_layernorm_fold(dev):
    """LayerNorm folded into the Linear that consumes it (MG_OP_IGEMM ln_in) and the row statistics taken in the
    producer's epilogue (ln_out), against torch layer_norm + linear in fp32: bf16 / fp32 / GEGLU epilogues, the
    transposed (V^T) section and several tiles."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(21)
    M, C = 700, 320
    x = _bf(torch.randn(M, C, generator=g) * 1.3 + 0.4 * torch.randn(M, 1, generator=g))    # rows with their own means
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    xd = x.to(dev, OP16)
    # --- producer: out = a @ w^T + b + res, ln_out = per-row (sum, sum of squares) over 32-column slots, then (mean, rstd)
    # of every row, reduced by the row block's last column tile
    K0 = 192
    a = _bf(torch.randn(M, K0, generator=g))
    w0 = _bf(torch.randn(C, K0, generator=g) / math.sqrt(K0))
    b0 = torch.randn(C, generator=g) * 0.1
    res = _bf(torch.randn(M, C, generator=g))
    ref0 = a @ w0.t() + b0 + res
    for variant in (0, 46, 51, 35, 24, 25, 26, 62, 72, 73):
        out = torch.empty(M, C, device=dev, dtype=OP16)
        st = torch.full((M * (C // 32 + 1), 2), float("nan"), device=dev)
        for rep_ in range(3):   # the tickets reset themselves: every launch finalizes again
            if rep_:
                st[M * (C // 32):] = float("nan")
            _run(ops.linear(a.to(dev, OP16), w0.to(dev, OP16), out, M=M, K=K0, N=C, bias=b0.to(dev),
                            residual=res.to(dev, OP16), ln_out=st, variant=variant))
            want = torch.stack([ref0.reshape(M, C // 32, 32).sum(-1), (ref0 ** 2).reshape(M, C // 32, 32).sum(-1)], dim=-1)
            _close(f"ln_out/slots/v{variant}", st[:M * (C // 32)].reshape(M, C // 32, 2), want, tol=2e-4)
            o32 = out.float().cpu()   # statistics are taken on the fp32 values before the bf16 rounding: compare loosely
            mr = st[M * (C // 32):].cpu()
            _close(f"ln_out/mean/v{variant}", mr[:, 0], ref0.mean(-1), tol=2e-4)
            _close(f"ln_out/rstd/v{variant}", mr[:, 1], 1.0 / torch.sqrt(ref0.var(-1, unbiased=False) + 1e-5), tol=2e-4)
            assert torch.isfinite(o32).all()
    # (mean, rstd) of x itself for the consumers (what a producer would have written)
    stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
    # --- consumer, bf16 and fp32 epilogues
    N = 640
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    b = torch.randn(N, generator=g) * 0.1
    wp, gv, cv = Wm.fold_layernorm(w, b, gamma, beta)
    ref = y @ w.t() + b
    for variant in (0, 36, 46, 51, 62, 72, 73):
        out = torch.full((M, N), float("nan"), device=dev, dtype=OP16)
        _run(ops.linear(xd, wp.to(dev), out, M=M, K=C, N=N, ln_in=stx, ln_g=gv.to(dev), ln_c=cv.to(dev), variant=variant))
        _close(f"ln_fold/bf16/v{variant}", out, ref, tol=2e-2)
    outf = torch.full((M, N), float("nan"), device=dev)
    _run(ops.linear(xd, wp.to(dev), outf, M=M, K=C, N=N, epi=L.EPI_F32, ln_in=stx, ln_g=gv.to(dev), ln_c=cv.to(dev)))
    _close("ln_fold/f32", outf, ref, tol=1.5e-2)
    # --- GEGLU epilogue
    wg = torch.randn(8 * C, C, generator=g) / math.sqrt(C)
    bg = torch.randn(8 * C, generator=g) * 0.1
    u, gt = (y @ wg.t() + bg).chunk(2, dim=-1)
    refg = u * F.gelu(gt)
    wpk, bpk = Wm.pack_geglu(wg, bg)
    wpg, gg, cg = Wm.fold_layernorm(wpk, bpk, gamma, beta)
    for variant in (0, 51, 62, 72, 73):
        og = torch.full((M, 4 * C), float("nan"), device=dev, dtype=OP16)
        _run(ops.linear(xd, wpg.to(dev), og, M=M, K=C, N=8 * C, epi=L.EPI_GEGLU, ln_in=stx, ln_g=gg.to(dev), ln_c=cg.to(dev),
                        variant=variant))
        _close(f"ln_fold/geglu/v{variant}", og, refg, tol=2e-2)
    # --- fused QKV with the transposed V section (B images x T tokens)
    B, T = 2, 350
    wq = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    wpq, gq, cq = Wm.fold_layernorm(wq, None, gamma, beta)
    refq = y @ wq.t()
    ldt = 384
    qk = torch.full((M, 2 * C), float("nan"), device=dev, dtype=OP16)
    vt = torch.zeros((B, C, ldt), device=dev, dtype=OP16)
    _run(ops.igemm(xd, wpq.to(dev), qk, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=3 * C, ldo=2 * C, out2=vt, trans_from=2 * C, ldt=ldt,
                   ln_in=stx, ln_g=gq.to(dev), ln_c=cq.to(dev)))
    _close("ln_fold/qkv/qk", qk, refq[:, :2 * C], tol=2e-2)
    _close("ln_fold/qkv/vt", vt[:, :, :T], refq[:, 2 * C:].reshape(B, T, C).permute(0, 2, 1), tol=2e-2)
    # permuted token order (flash_attn64 generation 3's V^T): the per-token LayerNorm statistics must follow the permutation
    T2 = 688
    vt2 = torch.zeros((1, C, 704), device=dev, dtype=OP16)
    _run(ops.igemm(xd, wpq.to(dev), qk, B=1, H=T2, W=1, Cin=C, Ho=T2, Wo=1, N=3 * C, ldo=2 * C, out2=vt2, trans_from=2 * C, ldt=704,
                   ln_in=stx, ln_g=gq.to(dev), ln_c=cq.to(dev), trans_perm=True))
    _close("ln_fold/qkv/vt_perm", vt2[:, :, :T2], ops.permute_vt_keys(refq[:T2, 2 * C:].reshape(1, T2, C).permute(0, 2, 1).contiguous()), tol=2e-2)


@pytest.mark.parametrize("C,waves", [(320, 12), (320, 8), (320, 4), (640, 8)])
def test_rowgemm_all_forms(dev, C, waves):
    """MG_OP_ROWGEMM (row-resident GEMM, K = 320) against torch fp32 on the same bf16 operands: every form the transformer
    uses at the 320-channel level - GroupNorm folded into the load + bias + row statistics (proj_in), QKV with the folded
    LayerNorm and the permuted V^T section, bias + in-place residual + row statistics (to_out), GEGLU with the folded
    LayerNorm, bias + residual (proj_out).  M is not a multiple of the workgroup's 384 / 256 / 128 rows: the surplus waves
    recompute the last row tile (identical stores)."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(31 + waves + C)
    B, T = 2, 1056
    M = B * T
    x = _bf(torch.randn(M, C, generator=g) * 0.9 + 0.3 * torch.randn(M, 1, generator=g))
    xd = x.to(dev, OP16)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
    nan_bf = lambda *sh: torch.full(sh, float("nan"), device=dev, dtype=OP16)

    # proj_in: GroupNorm scale / shift per (image, channel) applied while the rows are loaded, + bias, + row statistics
    ss = torch.stack([1 + 0.3 * torch.randn(B, C, generator=g), 0.2 * torch.randn(B, C, generator=g)], 1).contiguous()
    w, b = torch.randn(C, C, generator=g) / math.sqrt(C), 0.1 * torch.randn(C, generator=g)
    pk, ssd = Wm.pack_rowgemm(w, b).to(dev), ss.to(dev)
    out, so = nan_bf(M, C), torch.full((M, 2), float("nan"), device=dev)
    _run(ops.rowgemm(xd, pk, out, M=M, K=C, N=C, gn_ss=ssd, tokens=T, ln_out=so, waves=waves))
    xn = _bf(x.view(B, T, C) * ss[:, 0, None, :] + ss[:, 1, None, :]).view(M, C)
    ref = xn @ _bf(w).t() + b
    _close(f"rowgemm/gn+bias/{waves}w", out, ref)
    _close(f"rowgemm/gn+bias/stats/{waves}w", so, torch.stack([ref.mean(-1), 1.0 / torch.sqrt(ref.var(-1, unbiased=False) + 1e-5)], -1), tol=2e-4)

    # QKV: folded LayerNorm; columns >= 2C go to V^T [B][C][ldt] in the attention kernel's permuted key order
    wq = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    wp, lg, lc = Wm.fold_layernorm(wq, None, gamma, beta)
    pk = Wm.pack_rowgemm(wp.float(), lc, lg).to(dev)
    ldt = T + 32
    qk, vt = nan_bf(M, 2 * C), torch.zeros(B, C, ldt, device=dev, dtype=OP16)
    _run(ops.rowgemm(xd, pk, qk, M=M, K=C, N=3 * C, form=L.RG_QKV, ldo=2 * C, ln_in=stx, vt=vt, tokens=T, ldt=ldt, trans_from=2 * C, waves=waves))
    ref = y @ wq.t()
    _close(f"rowgemm/qkv/qk/{waves}w", qk, ref[:, :2 * C], tol=2e-2)
    want_vt = ops.permute_vt_keys(ref[:, 2 * C:].view(B, T, C).transpose(1, 2).contiguous())
    _close(f"rowgemm/qkv/vt/{waves}w", vt[:, :, :T], want_vt, tol=2e-2)
    assert not vt[:, :, T:].any(), "V^T pad columns were written"
    qk2, vt2 = nan_bf(M, 2 * C), torch.zeros_like(vt)
    _run(ops.rowgemm(xd, pk, qk2, M=M, K=C, N=3 * C, form=L.RG_QKV, ldo=2 * C, ln_in=stx, vt=vt2, tokens=T, ldt=ldt, trans_from=2 * C,
                     waves=waves, nsplit=4))
    assert torch.equal(qk, qk2) and torch.equal(vt, vt2), "QKV with the columns split over 4 workgroups per row block"

    # to_out: bias + residual IN PLACE + (mean, rstd) of the new rows
    h0 = _bf(torch.randn(M, C, generator=g))
    h, so = h0.to(dev, OP16).clone(), torch.full((M, 2), float("nan"), device=dev)
    pk = Wm.pack_rowgemm(w, b).to(dev)
    _run(ops.rowgemm(xd, pk, h, M=M, K=C, N=C, residual=h, ln_out=so, waves=waves))
    ref = x @ _bf(w).t() + b + h0
    _close(f"rowgemm/residual/{waves}w", h, ref)
    _close(f"rowgemm/residual/stats/{waves}w", so, torch.stack([ref.mean(-1), 1.0 / torch.sqrt(ref.var(-1, unbiased=False) + 1e-5)], -1), tol=2e-4)
    # proj_out: bias + residual from another tensor, no statistics; plain bias
    res = h0.to(dev, OP16)
    out = nan_bf(M, C)
    _run(ops.rowgemm(xd, pk, out, M=M, K=C, N=C, residual=res, waves=waves))
    _close(f"rowgemm/residual2/{waves}w", out, ref)
    out = nan_bf(M, C)
    _run(ops.rowgemm(xd, pk, out, M=M, K=C, N=C, waves=waves))
    _close(f"rowgemm/bias/{waves}w", out, x @ _bf(w).t() + b)
    # folded LayerNorm + bias (+ residual)
    wl, bl = torch.randn(2 * C, C, generator=g) / math.sqrt(C), 0.1 * torch.randn(2 * C, generator=g)
    wp, lg, lc = Wm.fold_layernorm(wl, bl, gamma, beta)
    pk = Wm.pack_rowgemm(wp.float(), lc, lg).to(dev)
    out = nan_bf(M, 2 * C)
    _run(ops.rowgemm(xd, pk, out, M=M, K=C, N=2 * C, ln_in=stx, waves=waves))
    _close(f"rowgemm/ln/{waves}w", out, y @ wl.t() + bl, tol=2e-2)

    # GEGLU: stage = 32 value channels + their 32 gates, folded LayerNorm
    wg, bg = torch.randn(8 * C, C, generator=g) / math.sqrt(C), 0.1 * torch.randn(8 * C, generator=g)
    order = Wm.rowgemm_geglu_order(8 * C)
    wp, lg, lc = Wm.fold_layernorm(wg[order], bg[order], gamma, beta)
    pk = Wm.pack_rowgemm(wp.float(), lc, lg).to(dev)
    hid = nan_bf(M, 4 * C)
    _run(ops.rowgemm(xd, pk, hid, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=waves))
    u, gt = (y @ wg.t() + bg).chunk(2, dim=-1)
    _close(f"rowgemm/geglu/{waves}w", hid, u * F.gelu(gt), tol=2e-2)
    # the column stages shared out over 3 / 7 workgroups per row block (few rows, many columns): same bits
    for nsplit in (3, 7):
        hs = nan_bf(M, 4 * C)
        _run(ops.rowgemm(xd, pk, hs, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=waves, nsplit=nsplit))
        assert torch.equal(hid, hs), f"column split {nsplit}"
    # run-to-run bit stability (counted waits on the weight stream: a race would show as differing elements)
    hid2 = nan_bf(M, 4 * C)
    _run(ops.rowgemm(xd, pk, hid2, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=waves))
    assert torch.equal(hid, hid2)


@pytest.mark.parametrize("C,heads,waves", [(320, 5, 12), (320, 5, 8), (640, 10, 0), (1280, 20, 0)])
def test_rowgemm_fused_cross_attention(dev, C, heads, waves):
    """MG_OP_ROWGEMM form RG_XATTN (the collapsed 2-token cross-attention, in place on the residual stream) against the
    unfused fp32 chain: LayerNorm -> scores -> pair softmax -> x VO^T + bias + residual, and the new rows' statistics."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(41 + waves + C)
    M = 2112
    x = _bf(torch.randn(M, C, generator=g) * 0.9 + 0.3 * torch.randn(M, 1, generator=g))
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    ctx = torch.randn(2, 1024, generator=g)
    wq, wo = torch.randn(C, C, generator=g) / math.sqrt(C), torch.randn(C, C, generator=g) / math.sqrt(C)
    wk, wv = torch.randn(C, 1024, generator=g) / 32, torch.randn(C, 1024, generator=g) / 32
    bo = 0.1 * torch.randn(C, generator=g)
    # reference: diffusers Attention with the 2-token context, fp32
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    q = (y @ wq.t()).view(M, heads, C // heads)
    k = (ctx @ wk.t()).view(2, heads, C // heads)
    v = (ctx @ wv.t()).view(2, heads, C // heads)
    p = torch.softmax(torch.einsum("mhd,jhd->mhj", q, k) / math.sqrt(C // heads), dim=-1)
    ref = torch.einsum("mhj,jhd->mhd", p, v).reshape(M, C) @ wo.t() + bo + x
    wqk, vot, npad = Wm.cross_attention_tables(wq, wk, wv, wo, ctx, heads)
    wp, lg, lc = Wm.fold_layernorm(wqk, None, gamma, beta)
    pk = (Wm.pack_rowgemm_xattn if C == 320 else Wm.pack_rowgemm_xattn_ksplit)(wp.float(), lc, lg, vot, bo).to(dev)
    stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
    h = x.to(dev, OP16).clone()
    so = torch.full((M, 2), float("nan"), device=dev)
    _run(ops.rowgemm(h, pk, h, M=M, K=C, N=64, form=L.RG_XATTN, ln_in=stx, ln_out=so, sm_cols=2 * heads,
                     sm_scale=1.0 / math.sqrt(C // heads), waves=waves))
    _close(f"rowgemm/xattn/C{C}/{waves}w", h, ref)
    _close(f"rowgemm/xattn/stats/C{C}/{waves}w", so, torch.stack([ref.mean(-1), 1.0 / torch.sqrt(ref.var(-1, unbiased=False) + 1e-5)], -1), tol=2e-3)


@pytest.mark.parametrize("waves", [12, 8, 4])
def test_rowgemm_cross_attention_as_geglu_prologue(dev, waves):
    """(round 6) MG_OP_ROWGEMM form RG_GEGLU with p[9]: the collapsed cross-attention runs on the rows in registers BEFORE the GEGLU
    projection - against the two launches it replaces (form RG_XATTN in place, then form RG_GEGLU on its rows and statistics) bit
    for bit (updated rows and hidden activations), against the unfused fp32 chain, and bit-stable over repeat launches.  M is not
    a multiple of the workgroup's rows: surplus waves keep the barriers and move weights only."""
    from marigold_amd import _lib as L, ops, weights as Wm
    C, heads = 320, 5
    g = torch.Generator().manual_seed(53 + waves)
    M = 2112
    x = _bf(torch.randn(M, C, generator=g) * 0.9 + 0.3 * torch.randn(M, 1, generator=g))
    g2, b2 = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    g3, b3 = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    ctx = torch.randn(2, 1024, generator=g)
    wq, wo = torch.randn(C, C, generator=g) / math.sqrt(C), torch.randn(C, C, generator=g) / math.sqrt(C)
    wk, wv = torch.randn(C, 1024, generator=g) / 32, torch.randn(C, 1024, generator=g) / 32
    bo = 0.1 * torch.randn(C, generator=g)
    wg, bg = torch.randn(8 * C, C, generator=g) / math.sqrt(C), 0.1 * torch.randn(8 * C, generator=g)
    # fp32 reference: x2 = x + attn2(LN2(x)); hidden = GEGLU(LN3(x2))
    y = F.layer_norm(x, (C,), g2, b2, 1e-5)
    q = (y @ wq.t()).view(M, heads, C // heads)
    k = (ctx @ wk.t()).view(2, heads, C // heads)
    v = (ctx @ wv.t()).view(2, heads, C // heads)
    p = torch.softmax(torch.einsum("mhd,jhd->mhj", q, k) / math.sqrt(C // heads), dim=-1)
    x2 = torch.einsum("mhj,jhd->mhd", p, v).reshape(M, C) @ wo.t() + bo + x
    u, gt = (F.layer_norm(x2, (C,), g3, b3, 1e-5) @ wg.t() + bg).chunk(2, dim=-1)
    ref_hid = u * F.gelu(gt)
    wqk, vot, npad = Wm.cross_attention_tables(wq, wk, wv, wo, ctx, heads)
    wpx, lgx, lcx = Wm.fold_layernorm(wqk, None, g2, b2)
    pkx = Wm.pack_rowgemm_xattn(wpx.float(), lcx, lgx, vot, bo).to(dev)
    order = Wm.rowgemm_geglu_order(8 * C)
    wpg, lgg, lcg = Wm.fold_layernorm(wg[order], bg[order], g3, b3)
    pkg = Wm.pack_rowgemm(wpg.float(), lcg, lgg).to(dev)
    stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
    nan_bf = lambda *sh: torch.full(sh, float("nan"), device=dev, dtype=OP16)
    kw = dict(sm_cols=2 * heads, sm_scale=1.0 / math.sqrt(C // heads))
    # the two launches
    h_a, so = x.to(dev, OP16).clone(), torch.full((M, 2), float("nan"), device=dev)
    _run(ops.rowgemm(h_a, pkx, h_a, M=M, K=C, N=64, form=L.RG_XATTN, ln_in=stx, ln_out=so, waves=12 if waves == 12 else 8, **kw))
    hid_a = nan_bf(M, 4 * C)
    _run(ops.rowgemm(h_a, pkg, hid_a, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=so, waves=waves))
    # the one launch, in place on the rows
    outs = []
    for rep in range(3):
        h_b, hid_b = x.to(dev, OP16).clone(), nan_bf(M, 4 * C)
        _run(ops.rowgemm(h_b, pkg, hid_b, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=waves, xattn=pkx, xout=h_b, **kw))
        outs.append((h_b, hid_b))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]), "repeat launches differ"
    h_b, hid_b = outs[0]
    assert torch.equal(h_a, h_b), "updated rows differ from the separate cross-attention launch"
    assert torch.equal(hid_a, hid_b), "hidden activations differ from the two-launch chain"
    _close(f"rowgemm/xattn+geglu/rows/{waves}w", h_b, x2)
    _close(f"rowgemm/xattn+geglu/hidden/{waves}w", hid_b, ref_hid, tol=2e-2)
    # out of place: the rows stay, the updated ones go to another buffer
    h_c, hid_c, x_in = nan_bf(M, C), nan_bf(M, 4 * C), x.to(dev, OP16).clone()
    _run(ops.rowgemm(x_in, pkg, hid_c, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=waves, xattn=pkx, xout=h_c, **kw))
    assert torch.equal(h_c, h_b) and torch.equal(hid_c, hid_b) and torch.equal(x_in, x.to(dev, OP16))
    # a column split would let one workgroup overwrite rows another still reads: refused
    with pytest.raises(Exception):
        _run(ops.rowgemm(h_b, pkg, hid_b, M=M, K=C, N=8 * C, form=L.RG_GEGLU, ln_in=stx, waves=4, nsplit=3, xattn=pkx, xout=h_b, **kw))


def test_rowgemm_rejects_shapes_outside_its_contract(dev):
    from marigold_amd import _lib as L, ops, weights as Wm
    x = torch.zeros(64, 320, device=dev, dtype=OP16)
    pk = Wm.pack_rowgemm(torch.zeros(128, 320), torch.zeros(128)).to(dev)
    out = torch.zeros(64, 128, device=dev, dtype=OP16)
    with pytest.raises(L.MarigoldHipError):
        ops.launch(ops.rowgemm(x, pk, out, M=48, K=320, N=128))       # M % 32
    with pytest.raises(L.MarigoldHipError):
        ops.launch(ops.rowgemm(x, pk, out, M=64, K=1280, N=128))      # K
    with pytest.raises(L.MarigoldHipError):
        ops.launch(ops.rowgemm(x, pk, out, M=64, K=320, N=96))        # N % 64


def test_layernorm_fold_heavy_tailed(dev):
    """The folded LayerNorm  rstd (acc - mean g[n]) + c[n]  cancels two large numbers when |mean| / std >> 1, and real
    SD-v2 residual streams have outlier channels and row means the synthetic weights never produce.  Rows with 2 % outlier
    channels at 30-100x, row means up to 20 sigma, gamma / beta with large entries, GEGLU gate pre-activations out to
    +-12: the fold against fp32 torch AND against the unfused chain (LayerNorm as its own pass with bf16 output -> the same
    Linear kernel), bf16 / GEGLU / pair-softmax-blend / transposed epilogues.  Bound: the fold may lose at
    most 2x against the unfused chain (else the operand needs centring)."""
    from marigold_amd import _lib as L, ops, synthetic as syn, weights as Wm
    M, C = 704, 320
    x = _bf(syn.heavy_tailed_rows(M, C, seed=3))
    gamma, beta = syn.heavy_tailed_affine(C, seed=3)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    xd = x.to(dev, OP16)
    stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
    yd = y.to(dev, OP16)   # the unfused chain's normalised rows: LayerNorm as its own pass, rounded to bf16
    g = torch.Generator().manual_seed(31)

    def rel(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))

    print(f"[stress] rows: |mean|/std up to {float((x.mean(-1).abs() / x.std(-1)).max()):.1f}, max |x| {float(x.abs().max()):.0f}")
    # --- bf16 epilogue
    N = 640
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    b = torch.randn(N, generator=g) * 0.1
    ref = y @ w.t() + b
    wp, gv, cv = Wm.fold_layernorm(w, b, gamma, beta)
    out = torch.empty(M, N, device=dev, dtype=OP16)
    _run(ops.linear(xd, wp.to(dev), out, M=M, K=C, N=N, ln_in=stx, ln_g=gv.to(dev), ln_c=cv.to(dev)))
    unf = torch.empty(M, N, device=dev, dtype=OP16)
    _run(ops.linear(yd, w.to(dev, OP16), unf, M=M, K=C, N=N, bias=b.to(dev)))
    e_f, e_u = rel(out, ref), rel(unf, ref)
    print(f"[stress] ln_fold/bf16: fold {e_f:.3e}  unfused {e_u:.3e}  (relative to max|ref| {float(ref.abs().max()):.1f})")
    assert e_f <= max(2 * e_u, 1.5e-2)
    # --- GEGLU with gate pre-activations out to +-12 (the polynomial CDF is clamped at |x| = 4)
    wg = torch.randn(8 * C, C, generator=g) / math.sqrt(C)
    wg[4 * C:] *= 12.0 / float((y @ wg[4 * C:].t()).abs().max())
    bg = torch.randn(8 * C, generator=g) * 0.1
    u, gt = (y @ wg.t() + bg).chunk(2, dim=-1)
    refg = u * F.gelu(gt)
    wpk, bpk = Wm.pack_geglu(wg, bg)
    wpg, gg, cg = Wm.fold_layernorm(wpk, bpk, gamma, beta)
    og = torch.empty(M, 4 * C, device=dev, dtype=OP16)
    _run(ops.linear(xd, wpg.to(dev), og, M=M, K=C, N=8 * C, epi=L.EPI_GEGLU, ln_in=stx, ln_g=gg.to(dev), ln_c=cg.to(dev)))
    ug = torch.empty(M, 4 * C, device=dev, dtype=OP16)
    _run(ops.linear(yd, wpk.to(dev, OP16), ug, M=M, K=C, N=8 * C, bias=bpk.to(dev), epi=L.EPI_GEGLU))
    e_f, e_u = rel(og, refg), rel(ug, refg)
    print(f"[stress] ln_fold/geglu (gate range +-{float(gt.abs().max()):.1f}): fold {e_f:.3e}  unfused {e_u:.3e}")
    assert e_f <= max(2 * e_u, 2e-2)
    # --- transposed (V^T) section of the fused QKV projection
    wq = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    wpq, gq, cq = Wm.fold_layernorm(wq, None, gamma, beta)
    refq = y @ wq.t()
    qk = torch.empty(M, 2 * C, device=dev, dtype=OP16)
    vt = torch.zeros(1, C, M, device=dev, dtype=OP16)
    _run(ops.igemm(xd, wpq.to(dev), qk, B=1, H=M, W=1, Cin=C, Ho=M, Wo=1, N=3 * C, ldo=2 * C, out2=vt, trans_from=2 * C, ldt=M,
                   ln_in=stx, ln_g=gq.to(dev), ln_c=cq.to(dev)))
    e_q, e_v = rel(qk, refq[:, :2 * C]), rel(vt[0], refq[:, 2 * C:].t())
    print(f"[stress] ln_fold/qkv: qk {e_q:.3e}  vt {e_v:.3e}")
    assert e_q <= 2e-2 and e_v <= 2e-2
    # --- the collapsed cross-attention in one launch (scores with the fold -> pair softmax -> blend + residual, in place)
    heads = C // 64
    ctx = torch.randn(2, 64, generator=g)
    wq2, wk2, wv2, wo2 = (torch.randn(C, C, generator=g) / math.sqrt(C), torch.randn(C, 64, generator=g) / 8,
                          torch.randn(C, 64, generator=g) / 8, torch.randn(C, C, generator=g) / math.sqrt(C))
    bo = torch.randn(C, generator=g) * 0.1
    wqk, vot, npad = Wm.cross_attention_tables(wq2, wk2, wv2, wo2, ctx, heads)
    wpx, gx, cx = Wm.fold_layernorm(wqk, None, gamma, beta)
    sc = (y @ wqk.t())[:, :2 * heads].reshape(M, heads, 2) / math.sqrt(C // heads)
    pr = torch.softmax(sc, dim=-1).reshape(M, 2 * heads)
    refx = x + pr @ vot[:, :2 * heads].t() + bo
    h = xd.clone()
    _run(ops.linear(h, wpx.to(dev), h, M=M, K=C, N=npad, epi=L.EPI_XATTN2, ln_in=stx, ln_g=gx.to(dev), ln_c=cx.to(dev),
                    sm_scale=1.0 / math.sqrt(C // heads), sm_cols=2 * heads, out2=vot.to(dev, OP16), c2=C, ldo=C,
                    bias=bo.to(dev), residual=h, ldr=C))
    e_x = rel(h, refx)
    print(f"[stress] ln_fold/xattn2: {e_x:.3e}")
    assert e_x <= 1.5e-2


def test_conv3x3_patch_fused_norm_heavy_tailed(dev):
    """The in-LDS GroupNorm fix-up of the patch convolution rounds silu(x scale + shift) to bf16 before the MFMAs, as the
    stand-alone pass does: with outlier channels (x50) and scales up to x30 the fused kernel must still agree with the
    unfused chain to one bf16 step of the result, and with fp32 torch within the bf16 bound."""
    from marigold_amd import ops, synthetic as syn, weights as Wm
    B, H, W, Cin, N = 2, 32, 32, 320, 320
    g = torch.Generator().manual_seed(17)
    x = _bf(syn.heavy_tailed_rows(B * H * W, Cin, seed=9, mean_sigma=3.0).reshape(B, H, W, Cin))
    gamma, beta = syn.heavy_tailed_affine(Cin, seed=9, gain=30.0)
    xg = x.permute(0, 3, 1, 2).reshape(B, 32, -1)
    mean, rstd = xg.mean(-1), (xg.var(-1, unbiased=False) + 1e-5).rsqrt()
    sc = rstd.repeat_interleave(Cin // 32, 1) * gamma
    sh = beta - mean.repeat_interleave(Cin // 32, 1) * sc
    ss = torch.stack([sc, sh], dim=1).contiguous()
    w4 = _bf(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(N, generator=g) * 0.1
    ref = _ref_fused_conv(x.permute(0, 3, 1, 2), w4, bias, ss, True, None, None)
    xd, ssd, wd, bd = x.to(dev, OP16), ss.to(dev), Wm.pack_conv3x3(w4).to(dev, OP16), bias.to(dev)
    out = torch.empty(B, H, W, N, device=dev, dtype=OP16)
    _run(ops.conv3x3(xd, wd, out, B=B, H=H, W=W, C0=Cin, N=N, ss=ssd, silu=True, bias=bd))
    hh = torch.empty(B, H, W, Cin, device=dev, dtype=OP16)
    _run(ops.gn_apply(xd, ssd, hh, B=B, HW=H * W, C=Cin, silu=True))
    unf = torch.empty_like(out)
    _run(ops.igemm(hh, wd, unf, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=N, taps=9, stride=1, pad=1, bias=bd))
    scale = ref.abs().max().item()
    d_fu = (out.float() - unf.float()).abs().max().item()
    print(f"[stress] fused conv3x3p vs unfused chain: max|diff| {d_fu:.3e} (scale {scale:.1f}); normalised input up to "
          f"{float(F.silu(x * sc[:, None, None, :] + sh[:, None, None, :]).abs().max()):.0f}")
    assert d_fu <= scale / 128
    _close("conv3x3p/heavy-tailed vs fp32", out.float().permute(0, 3, 1, 2), ref)


def test_conv_patch_and_flash_bit_stable_at_scale(dev):
    """Repeat-launch bit stability at benchmark scale for the two other kernels that regroup accumulators with
    v_permlane32_swap (the packed-fp32 build that differed from run to run in round 2 was only ever stressed on igemm2):
    the level-0 ResNet convolution with the fused GroupNorm fix-up (10 x 96 x 96, 320 -> 320) and the level-0 self-attention
    (10 x 5 heads x 9216 tokens), six launches each."""
    from marigold_amd import ops, weights as Wm
    g = torch.Generator().manual_seed(8)
    B, H, W, C = 10, 96, 96, 320
    x = torch.randn(B, H, W, C, generator=g).to(dev, OP16)
    w = Wm.pack_conv3x3(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev, OP16)
    ss = torch.stack([1.0 + 0.3 * torch.randn(B, C, generator=g), 0.3 * torch.randn(B, C, generator=g)], dim=1).to(dev).contiguous()
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    res = torch.randn(B, H, W, C, generator=g).to(dev, OP16)
    first = None
    for _ in range(6):
        out = torch.full((B, H, W, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.conv3x3(x, w, out, B=B, H=H, W=W, C0=C, N=C, ss=ss, silu=True, bias=bias, residual=res))
        if first is None:
            first = out.clone()
        assert torch.equal(first, out), "conv_patch: launches differ"
    heads, T = 5, 9216
    qkv = torch.randn(B, T, 3 * C, generator=g).to(dev, OP16)
    vt = ops.permute_vt_keys(qkv[:, :, 2 * C:].permute(0, 2, 1).contiguous())
    first = None
    for _ in range(6):
        out = torch.full((B, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkv, qkv[:, :, C:], vt, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T,
                              sq=T * 3 * C, sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125, vt_perm=True))
        if first is None:
            first = out.clone()
        assert torch.equal(first, out), "flash_attn64: launches differ"
    assert torch.isfinite(first.float()).all()


def test_rowgemm_bit_stable_at_scale_and_equal_to_tile_gemm(dev):
    """MG_OP_ROWGEMM at the benchmark's shapes (92 160 rows x 320 channels; 240 workgroups x 12 waves on hand-counted vmcnt /
    lgkmcnt waits: a race would show as launches that differ): QKV with the folded LayerNorm six times - and bit-identical to
    the tile GEMM's output, whose reduction order it keeps - the in-place cross-attention and its K-split form at 1 280
    channels, six launches each from the same input."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(12)
    B, T, C, heads = 10, 9216, 320, 5
    M = B * T
    x = (torch.randn(M, C, generator=g) * 0.8).to(OP16)
    xd = x.to(dev)
    st = torch.stack([x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous().to(dev)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wq = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    wp, lg, lc = Wm.fold_layernorm(wq, None, gamma, beta)
    pk = Wm.pack_rowgemm(wp.float(), lc, lg).to(dev)
    first = None
    for _ in range(6):
        qk = torch.full((M, 2 * C), float("nan"), device=dev, dtype=OP16)
        vt = torch.zeros(B, C, T, device=dev, dtype=OP16)
        _run(ops.rowgemm(xd, pk, qk, M=M, K=C, N=3 * C, form=L.RG_QKV, ldo=2 * C, ln_in=st, vt=vt, tokens=T, ldt=T, trans_from=2 * C))
        if first is None:
            first = (qk.clone(), vt.clone())
        assert torch.equal(first[0], qk) and torch.equal(first[1], vt), "rowgemm QKV: launches differ"
    wpd, lgd, lcd = wp.to(dev), lg.to(dev), lc.to(dev)
    qk2 = torch.full((M, 2 * C), float("nan"), device=dev, dtype=OP16)
    vt2 = torch.zeros(B, C, T, device=dev, dtype=OP16)
    _run(ops.igemm(xd, wpd, qk2, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=3 * C, ldo=2 * C, out2=vt2, trans_from=2 * C, ldt=T,
                   ln_in=st, ln_g=lgd, ln_c=lcd, trans_perm=True))
    assert torch.equal(first[0], qk2) and torch.equal(first[1], vt2), "rowgemm QKV differs from the tile GEMM"
    # cross-attention, in place: row-resident (K = 320) and K-split (K = 1280, the 24 x 24 level's 5 760 rows)
    for (Cx, hx, Mx) in ((320, 5, M), (1280, 20, 5760)):
        xx = (torch.randn(Mx, Cx, generator=g) * 0.8).to(OP16)
        stx = torch.stack([xx.float().mean(1), (xx.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous().to(dev)
        ga, be = 1 + 0.2 * torch.randn(Cx, generator=g), 0.1 * torch.randn(Cx, generator=g)
        ctx = torch.randn(2, 1024, generator=g)
        wq2, wo2 = torch.randn(Cx, Cx, generator=g) / math.sqrt(Cx), torch.randn(Cx, Cx, generator=g) / math.sqrt(Cx)
        wk2, wv2 = torch.randn(Cx, 1024, generator=g) / 32, torch.randn(Cx, 1024, generator=g) / 32
        wqk, vot, _ = Wm.cross_attention_tables(wq2, wk2, wv2, wo2, ctx, hx)
        wpx, lgx, lcx = Wm.fold_layernorm(wqk, None, ga, be)
        pack = Wm.pack_rowgemm_xattn if Cx == 320 else Wm.pack_rowgemm_xattn_ksplit
        pkx = pack(wpx.float(), lcx, lgx, vot, 0.1 * torch.randn(Cx, generator=g)).to(dev)
        first = None
        for _ in range(6):
            h = xx.to(dev).clone()
            so = torch.full((Mx, 2), float("nan"), device=dev)
            _run(ops.rowgemm(h, pkx, h, M=Mx, K=Cx, N=64, form=L.RG_XATTN, ln_in=stx, ln_out=so, sm_cols=2 * hx,
                             sm_scale=1.0 / math.sqrt(Cx // hx)))
            if first is None:
                first = (h.clone(), so.clone())
            assert torch.equal(first[0], h) and torch.equal(first[1], so), f"rowgemm cross-attention (C = {Cx}): launches differ"
        assert torch.isfinite(first[0].float()).all() and torch.isfinite(first[1]).all()


def test_igemm_row_statistics_bit_stable_at_scale(dev):
    """In-place Linear + residual with row statistics at benchmark sizes, on the tiles that share a CU between two
    workgroups: every launch must give the same bits (outputs and (mean, rstd)), and the statistics must match fp32.
    (Round 2: an SLP-packed build of the interior epilogue gave run-to-run different single elements here - the igemm2
    object is built with -fno-slp-vectorize since; tools/det_stress.py is the long form of this test.)"""
    from marigold_amd import ops
    g = torch.Generator().manual_seed(3)
    for M, K, N, variants in ((5760, 1280, 1280, (35, 0)), (23040, 640, 640, (35, 32)), (92160, 64, 320, (35, 0))):
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev, OP16)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, OP16)
        b = torch.randn(N, generator=g).to(dev)
        h0 = torch.randn(M, N, generator=g).to(dev, OP16)
        ref = a.float() @ w.float().t() + b + h0.float()
        for variant in variants:
            first = None
            for _ in range(6):
                h = h0.clone()
                st = torch.full((M * (N // 32 + 1), 2), float("nan"), device=dev)
                _run(ops.linear(a, w, h, M=M, K=K, N=N, bias=b, residual=h, ln_out=st, variant=variant))
                got = (h.clone(), st[M * (N // 32):].clone())
                if first is None:
                    first = got
                    _close(f"stats at scale/mean/M{M}/v{variant}", got[1][:, 0], ref.mean(-1), tol=1e-4)
                    _close(f"stats at scale/rstd/M{M}/v{variant}", got[1][:, 1], 1.0 / torch.sqrt(ref.var(-1, unbiased=False) + 1e-5), tol=1e-4)
                    _close(f"stats at scale/out/M{M}/v{variant}", got[0], ref, tol=1.5e-2)
                else:
                    assert torch.equal(first[0], got[0]), f"M{M} v{variant}: outputs differ between launches"
                    assert torch.equal(first[1], got[1]), f"M{M} v{variant}: (mean, rstd) differ between launches"


def test_igemm_pair_softmax_epilogue(dev):
    """MG_EPI_SOFTMAX2: the collapsed cross-attention's probabilities straight from the scores GEMM (LayerNorm folded in),
    against layer_norm -> linear -> softmax over the key pair in torch fp32; pad columns must come out zero."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(31)
    for M, C, heads in ((900, 320, 5), (333, 640, 10), (130, 1280, 20)):
        npad = 64
        x = _bf(torch.randn(M, C, generator=g) * 1.2 + 0.3 * torch.randn(M, 1, generator=g))
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
        wqk = torch.zeros(npad, C)
        wqk[:2 * heads] = torch.randn(2 * heads, C, generator=g) * (3.0 / math.sqrt(C))
        scale = 1.0 / math.sqrt(64)
        y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
        sc = (y @ wqk.t())[:, :2 * heads].reshape(M, heads, 2) * scale
        ref = torch.zeros(M, npad)
        ref[:, :2 * heads] = torch.softmax(sc, dim=-1).reshape(M, 2 * heads)
        wp, gv, cv = Wm.fold_layernorm(wqk, None, gamma, beta)
        stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
        out = torch.full((M, npad), float("nan"), device=dev, dtype=OP16)
        _run(ops.linear(x.to(dev, OP16), wp.to(dev), out, M=M, K=C, N=npad, epi=L.EPI_SOFTMAX2, ln_in=stx,
                        ln_g=gv.to(dev), ln_c=cv.to(dev), sm_scale=scale, sm_cols=2 * heads))
        _close(f"pair softmax epilogue C{C}", out, ref, tol=1.5e-2)
        assert (out[:, 2 * heads:] == 0).all()


def test_igemm_fused_cross_attention(dev):
    """MG_EPI_XATTN2: LayerNorm -> 2-token scores -> pair softmax -> blend with the pushed-through values + bias + residual,
    in place on the residual stream, and (mean, rstd) of the new rows - against the same chain in torch fp32 (probabilities
    rounded to bf16 where the kernel packs them into the second MFMA stage's operand)."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(41)
    for M, C, heads in ((900, 320, 5), (333, 640, 10), (130, 1280, 20), (128, 64, 1)):
        npad = 64
        x = _bf(torch.randn(M, C, generator=g) * 1.2 + 0.3 * torch.randn(M, 1, generator=g))
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
        wqk = torch.zeros(npad, C)
        wqk[:2 * heads] = torch.randn(2 * heads, C, generator=g) * (3.0 / math.sqrt(C))
        vot = torch.zeros(C, npad)
        vot[:, :2 * heads] = torch.randn(C, 2 * heads, generator=g) * 0.5
        vot = _bf(vot)
        bias = torch.randn(C, generator=g) * 0.1
        scale = 1.0 / math.sqrt(64)
        y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
        sc = (y @ wqk.t())[:, :2 * heads].reshape(M, heads, 2) * scale
        P = torch.zeros(M, npad)
        P[:, :2 * heads] = torch.softmax(sc, dim=-1).reshape(M, 2 * heads)
        ref = _bf(P) @ vot.t() + bias + x
        wp, gv, cv = Wm.fold_layernorm(wqk, None, gamma, beta)
        stx = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], dim=-1).to(dev).contiguous()
        first = None
        for rep_ in range(3):
            h = x.to(dev, OP16).clone()
            mr = torch.full((M, 2), float("nan"), device=dev)
            keep = (wp.to(dev), gv.to(dev), cv.to(dev), vot.to(dev, OP16), bias.to(dev))
            _run(ops.linear(h, keep[0], h, M=M, K=C, N=npad, epi=L.EPI_XATTN2, ln_in=stx, ln_g=keep[1], ln_c=keep[2], sm_scale=scale,
                            sm_cols=2 * heads, out2=keep[3], c2=C, ldo=C, bias=keep[4], residual=h, ldr=C, ln_out=mr))
            if first is None:
                first = (h.clone(), mr.clone())
                _close(f"fused cross-attention C{C}", h, ref, tol=1.5e-2)
                _close(f"fused cross-attention C{C} mean", mr[:, 0], ref.mean(-1), tol=3e-3)
                _close(f"fused cross-attention C{C} rstd", mr[:, 1], 1.0 / torch.sqrt(ref.var(-1, unbiased=False) + 1e-5), tol=3e-3)
            else:
                assert torch.equal(first[0], h) and torch.equal(first[1], mr), "fused cross-attention is not bit-repeatable"
    # without residual / bias / statistics, not in place
    M, C, heads = 200, 320, 5
    x = _bf(torch.randn(M, C, generator=g))
    wqk = torch.zeros(64, C); wqk[:10] = torch.randn(10, C, generator=g) / math.sqrt(C)
    vot = _bf(torch.randn(C, 64, generator=g) * 0.5)
    sc = (x @ _bf(wqk).t())[:, :10].reshape(M, heads, 2) * 0.125
    P = torch.zeros(M, 64); P[:, :10] = torch.softmax(sc, dim=-1).reshape(M, 10)
    out = torch.full((M, C), float("nan"), device=dev, dtype=OP16)
    keep = (x.to(dev, OP16), _bf(wqk).to(dev, OP16), vot.to(dev, OP16))
    _run(ops.linear(keep[0], keep[1], out, M=M, K=C, N=64, epi=L.EPI_XATTN2, sm_scale=0.125, sm_cols=10, out2=keep[2], c2=C, ldo=C))
    _close("fused cross-attention plain", out, _bf(P) @ vot.t(), tol=1.5e-2)


def test_igemm_pingpong_short_k_and_repeatability(dev):
    """Tile variant 62 (two wave groups one barrier apart, four phases per K tile): the one- and two-tile K loops
    (prologue / drain only), and a many-tile problem launched repeatedly - every launch must give the same bits
    (a staged half tile read before its DMA landed would show up as run-to-run differences)."""
    from marigold_amd import ops
    g = torch.Generator().manual_seed(11)
    for M, K, N in ((300, 64, 320), (700, 128, 256), (513, 192, 512)):
        x = _bf(torch.randn(M, K, generator=g))
        w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        b = torch.randn(N, generator=g) * 0.1
        out = torch.full((M, N), float("nan"), device=dev, dtype=OP16)
        _run(ops.linear(x.to(dev, OP16), w.to(dev, OP16), out, M=M, K=K, N=N, bias=b.to(dev),
                        variant=62))
        _close(f"pingpong/linear K={K}", out, x @ w.t() + b)
        for v in (72, 73):
            out.fill_(float("nan"))
            _run(ops.linear(x.to(dev, OP16), w.to(dev, OP16), out, M=M, K=K, N=N, bias=b.to(dev),
                            variant=v))
            _close(f"pingpong/linear K={K} v{v}", out, x @ w.t() + b)
    M, K, N = 8192, 2304, 768
    x = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    xd, wd = x.to(dev, OP16), w.to(dev, OP16)
    ref = x @ w.t()
    out62 = torch.empty((M, N), device=dev, dtype=OP16)
    _run(ops.linear(xd, wd, out62, M=M, K=K, N=N, variant=62))
    _close("linear 8192x2304x768 v62", out62, ref)
    for v in (62, 72):
        for rep in range(4):
            out = torch.full((M, N), float("nan"), device=dev, dtype=OP16)
            _run(ops.linear(xd, wd, out, M=M, K=K, N=N, variant=v))
            # same MFMA sequence per accumulator in the ping-pong and the hand-placed 256 x 256 tiles -> identical bits, every launch
            assert torch.equal(out, out62), f"v{v} launch {rep} differs from the first ping-pong launch"
    # split-K named by the op (i[31]; the tuning table's second field): fp32 partials + the fixed-order reduce launch, any tile
    for v, sp in ((36, 2), (73, 3), (72, 4), (23, 8), (0, 6), (24, 12), (25, 5), (26, 16)):
        first = None
        for rep in range(2):
            out = torch.full((M, N), float("nan"), device=dev, dtype=OP16)
            _run(ops.linear(xd, wd, out, M=M, K=K, N=N, variant=v, splits=sp))
            if first is None:
                first = out.clone()
                _close(f"linear 8192x2304x768 v{v} split {sp}", out, ref)
            assert torch.equal(out, first), f"v{v} split {sp}: launches differ"


# --------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,H,W,C,silu,eps", [(2, 12, 20, 64, True, 1e-5), (1, 9, 7, 320, True, 1e-6),
                                              (3, 8, 8, 960, False, 1e-6), (1, 16, 16, 2560, True, 1e-5),
                                              (2, 24, 24, 128, True, 1e-6)])
def test_groupnorm(dev, B, H, W, C, silu, eps):
    from marigold_amd import ops
    g = torch.Generator().manual_seed(C)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    HW = H * W
    chunks = min(HW, 8)
    xd = _nhwc(x).to(dev, OP16)
    part = torch.empty(B, chunks, C, 2, device=dev)
    ss = torch.empty(B, 2, C, device=dev)
    out = torch.full_like(xd, float("nan"))
    for op in (ops.gn_stats(xd, part, B=B, HW=HW, C=C, chunks=chunks, groups=32),
               ops.gn_finalize(part, gamma.to(dev), beta.to(dev), ss, B=B, C=C, groups=32,
                               slots=chunks, HW=HW, eps=eps),
               ops.gn_apply(xd, ss, out, B=B, HW=HW, C=C, silu=silu)):
        _run(op)
    _close(f"groupnorm/C{C}", out.float().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("B,H,W,C0,C1,silu,eps", [
    (2, 48, 48, 640, 0, True, 1e-5),       # level 1: 20-channel groups, 92 KB slabs, 12 rows per thread in registers
    (2, 24, 24, 1280, 1280, True, 1e-5),   # skip concat 2560: 80-channel groups, a window never straddles the sources
    (3, 12, 12, 1280, 0, False, 1e-6),     # 256-thread workgroups
    (1, 96, 96, 320, 0, False, 1e-6),      # level 0: 10-channel groups -> 20-channel windows (a vector straddles two groups)
    (2, 48, 48, 640, 320, True, 1e-5),     # 960 = 30-channel groups -> 60-channel windows, second source from channel 640
    (1, 13, 9, 320, 320, True, 1e-5),      # odd map
])
def test_groupnorm_one_launch(dev, B, H, W, C0, C1, silu, eps):
    """MG_OP_GN_SLAB (statistics + scale/shift + normalised output in one launch, rows resident in registers) against
    torch group_norm in fp32; the statistics-only form against the same scale / shift; both bit-stable across launches."""
    from marigold_amd import ops
    C = C0 + C1
    g = torch.Generator().manual_seed(C + H)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3 + torch.randn(1, C, 1, 1, generator=g))
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xg = x.reshape(B, 32, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = (var + eps).rsqrt()
    cpg = C // 32
    sc_ref = rstd.repeat_interleave(cpg, 1) * gamma
    sh_ref = beta - mean.repeat_interleave(cpg, 1) * sc_ref
    HW = H * W
    xn = _nhwc(x)
    x0 = xn[..., :C0].contiguous().to(dev, OP16)
    x1 = xn[..., C0:].contiguous().to(dev, OP16) if C1 else None
    outs = []
    for rep in range(2):
        ss = torch.full((B, 2, C), float("nan"), device=dev)
        out = torch.full((B, HW, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.gn_slab(x0, out, ss, B=B, HW=HW, C=C, groups=32, gamma=gamma.to(dev), beta=beta.to(dev), eps=eps, silu=silu,
                         x1=x1, C0=C0))
        outs.append((ss.clone(), out.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ss, out = outs[0]
    _close(f"gn_slab/scale/C{C}", ss[:, 0], sc_ref, tol=2e-4)
    _close(f"gn_slab/shift/C{C}", ss[:, 1], sh_ref, tol=2e-4)
    _close(f"gn_slab/out/C{C}", out.reshape(B, H, W, C).float().permute(0, 3, 1, 2), ref)
    ss2 = torch.full((B, 2, C), float("nan"), device=dev)
    _run(ops.gn_slab(x0, None, ss2, B=B, HW=HW, C=C, groups=32, gamma=gamma.to(dev), beta=beta.to(dev), eps=eps, x1=x1, C0=C0))
    _close(f"gn_slab/stats_only/C{C}", ss2, ss, tol=1e-5)


# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,heads,T", [(1, 1, 64), (2, 2, 144), (1, 5, 200), (1, 2, 576), (2, 1, 1000)])
def test_flash_attn64(dev, B, heads, T):
    from marigold_amd import ops
    C = heads * 64
    g = torch.Generator().manual_seed(T)
    qkv = _bf(torch.randn(B, T, 3 * C, generator=g))
    q, k, v = qkv.split(C, dim=-1)
    qh, kh, vh = (t.reshape(B, T, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, T, C)
    ldvt = ((T + 63) // 64) * 64
    qkd = qkv.to(dev, OP16)
    vt = torch.zeros(B, C, ldvt, device=dev, dtype=OP16)
    vt[:, :, :T] = v.permute(0, 2, 1).to(dev, OP16)
    vtp = ops.permute_vt_keys(vt)   # the key order generation 3 reads without a lane exchange (MG_OP_IGEMM trans_perm)
    # 0 = automatic; 21 = the compiled kernel on the natural V^T; 19 / 20 / 25 (and 0 with vt_perm) = on the permuted V^T
    for variant, perm in [(0, False), (21, False)] + ([(v, True) for v in (0, 19, 20, 25)] if T % 16 == 0 else []):
        out = torch.full((B, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkd, qkd[:, :, C:], vtp if perm else vt, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C,
                              ldvt=ldvt, sq=T * 3 * C, sk=T * 3 * C, svt=C * ldvt, so=T * C, scale=0.125,
                              variant=variant, vt_perm=perm))
        _close(f"flash_attn64/B{B}h{heads}T{T}/v{variant}{'p' if perm else ''}", out, ref)


def test_flash_attn64_spiky_scores(dev):
    """Online-softmax rescale path: one key dominates late in the sequence."""
    from marigold_amd import ops
    T, C = 320, 64
    g = torch.Generator().manual_seed(3)
    q = _bf(torch.randn(1, T, C, generator=g))
    k = _bf(torch.randn(1, T, C, generator=g))
    v = _bf(torch.randn(1, T, C, generator=g))
    k[0, 250] = q[0, 17] * 4.0   # huge score for query 17 at key 250 (4th tile)
    k[0, 5] = q[0, 100] * 3.0
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    qkv = torch.cat([q, k], dim=-1).to(dev, OP16)
    vt = v.permute(0, 2, 1).contiguous().to(dev, OP16)
    vtp = ops.permute_vt_keys(vt)
    for variant, perm in [(0, False), (21, False)] + [(v, True) for v in (0, 19, 20, 25)]:
        out = torch.full((1, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkv, qkv[:, :, C:], vtp if perm else vt, out, B=1, heads=1, Ntok=T, ldq=2 * C, ldo=C, ldvt=T,
                              sq=0, sk=0, svt=0, so=0, scale=0.125, variant=variant, vt_perm=perm))
        _close(f"flash_attn64/spiky/v{variant}{'p' if perm else ''}", out, ref)


@pytest.mark.parametrize("case", ["all_negative", "growing", "first_tile_spike", "threshold_edge"])
def test_flash_attn64_running_max_paths(dev, case):
    """The compiled kernel (generation 2.5) keeps the running max inside the MFMA's C operand and only raises it when a row maximum exceeds it by
    more than 2^3 (log2 units): inputs that force each branch of that logic, against fp32 SDPA on the host.
      all_negative     - every logit is around -60 (the max of the first tile is the reference whatever its sign; with a
                         zero-initialised max every probability would underflow);
      growing          - the row maximum grows by ~2.5 log2 units per tile (below the threshold every time: probabilities
                         reach 2^3 times the reference before the max is raised) and then by ~6 (above);
      first_tile_spike - the largest logit of some rows sits in tile 0, of others in the last (ragged) tile;
      threshold_edge   - jumps of exactly 3.0 +- one bf16 step around the threshold."""
    from marigold_amd import ops
    T, C = 1008, 64   # 16 tiles, the last one ragged (48 keys), 4 full ring rounds
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, T, C, generator=g)
    k = torch.randn(1, T, C, generator=g)
    v = _bf(torch.randn(1, T, C, generator=g))
    if case == "all_negative":
        u = torch.nn.functional.normalize(torch.randn(C, generator=g), dim=0)
        q = q * 0.2 + u * 22.0
        k = k * 0.2 - u * 22.0          # q.k ~ -484, logits ~ -60
    elif case == "growing":
        u = torch.nn.functional.normalize(torch.randn(C, generator=g), dim=0)
        q = q * 0.3 + u * 8.0
        step = torch.arange(T) // 64    # tile index
        amp = torch.where(step < 10, step * 1.75, 17.5 + (step - 10) * 4.2)   # logit = amp * 8 / 8: +1.75 (2.5 log2), then +4.2 (6 log2)
        k = k * 0.3 + u[None, :] * amp[:, None]
    elif case == "first_tile_spike":
        k[0, 3] = q[0, 10] * 5.0
        k[0, 990] = q[0, 500] * 5.0
        k[0, 1007] = q[0, 700] * 6.0
    else:
        u = torch.nn.functional.normalize(torch.randn(C, generator=g), dim=0)
        q = q * 0.05 + u * 8.0
        step = (torch.arange(T) // 64).float()
        amp = step * (3.0 / 1.4426950408889634)                          # +3.0 log2 units per tile ...
        amp = amp + torch.where(step % 2 == 0, 0.02, -0.02)              # ... give or take
        k = k * 0.05 + u[None, :] * amp[:, None]
    q, k = _bf(q), _bf(k)
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    assert torch.isfinite(ref).all()
    qkv = torch.cat([q, k], dim=-1).to(dev, OP16)
    ldvt = 1024
    vt = torch.zeros(1, C, ldvt, device=dev, dtype=OP16)
    vt[:, :, :T] = v.permute(0, 2, 1).to(dev, OP16)
    vtp = ops.permute_vt_keys(vt)
    for variant, perm in [(0, False), (21, False)] + [(v, True) for v in (0, 19, 20, 25)]:
        out = torch.full((1, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkv, qkv[:, :, C:], vtp if perm else vt, out, B=1, heads=1, Ntok=T, ldq=2 * C, ldo=C, ldvt=ldvt,
                              sq=0, sk=0, svt=0, so=0, scale=0.125, variant=variant, vt_perm=perm))
        _close(f"flash_attn64/{case}/v{variant}{'p' if perm else ''}", out, ref)


@pytest.mark.parametrize("B,heads,T", [(1, 1, 256), (2, 2, 512), (1, 5, 1024), (1, 2, 2304), (2, 1, 768), (13, 5, 1024), (3, 23, 1280)])
@pytest.mark.parametrize("redo_thr", [0.0, 1e-30])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("variant", [26, 27], ids=["mfma32x32x16", "mfma16x16x32"])
def test_flash_attn64_hand_placed(dev, B, heads, T, redo_thr, split, variant):
    """Variants 26 / 27 (the key loop on 32x32x16 / on 16x16x32 MFMAs; flash4w.hip: 64 queries per wave, the key loop as one hand-placed instruction stream, softmax against the
    row maximum over the first 32 keys) against fp32 SDPA on the host; ``redo_thr`` = 1e-30 sends every workgroup through its
    running-maximum fallback as well; ``split``: with the workspace, i.e. the blocks of queries beyond a multiple of the CU
    count split along the keys (13 x 5 x 4 = 260 and 3 x 23 x 5 = 345 blocks: whole ones AND pieces in one launch; the small
    shapes: pieces only) - twice, the tickets must be back at zero."""
    from marigold_amd import ops
    C = heads * 64
    g = torch.Generator().manual_seed(T + heads)
    qkv = _bf(torch.randn(B, T, 3 * C, generator=g) * 1.3)
    q, k, v = qkv.split(C, dim=-1)
    qh, kh, vh = (t.reshape(B, T, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, T, C)
    qkd = qkv.to(dev, OP16)
    vtp = ops.permute_vt_keys(v.permute(0, 2, 1).contiguous().to(dev, OP16))
    ws = torch.zeros(ops.FLASH_WS_BYTES, dtype=torch.uint8, device=dev) if split else None
    first = None
    for rnd in range(3 if split else 1):
        out = torch.full((B, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkd, qkd[:, :, C:], vtp, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T, sq=T * 3 * C,
                              sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125, variant=variant, vt_perm=True, redo_thr=redo_thr,
                              ws=ws, ws_bytes=ops.FLASH_WS_BYTES if split else 0, split=1 if split else 2))
        _close(f"flash_attn64/hand-placed/v{variant}/B{B}h{heads}T{T}/thr{redo_thr}/split{int(split)}/{rnd}", out, ref)
        if split:
            assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, "tickets not back at zero"
            if first is None:
                first = out.clone()
            assert torch.equal(first, out), "key-split launches must be bit-identical (the pieces are summed in a fixed order)"


def test_flash_split_handoff_stress(dev):
    """The key-split hand-off (sc1 slab stores -> vmcnt(0) -> barrier -> one relaxed ticket -> sc1 loads, INTEGRATION.md
    "Hardware assumption") at the benchmark's shape - 10 members x 5 heads x 9 216 tokens = 1 800 blocks = 7 x 256 + 8: the eight
    left-over blocks go out as key pieces over every XCD - 200 automatic-split launches back to back with the operands rewritten
    in between (so that a stale slab or a lost ticket shows), each bit-identical to the first and close to the unsplit launch."""
    from marigold_amd import ops
    B, heads, T = 10, 5, 9216
    C = heads * 64
    g = torch.Generator(device=dev).manual_seed(11)
    qkd = (torch.randn(B, T, 3 * C, device=dev, generator=g) * 1.2).to(OP16)
    vtp = ops.permute_vt_keys(qkd[:, :, 2 * C:].permute(0, 2, 1).contiguous())
    ws = torch.zeros(ops.FLASH_WS_BYTES_AUTO, dtype=torch.uint8, device=dev)

    def run(split, out):
        _run(ops.flash_attn64(qkd, qkd[:, :, C:], vtp, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T, sq=T * 3 * C,
                              sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125, vt_perm=True, ws=ws if split else None,
                              ws_bytes=ops.FLASH_WS_BYTES_AUTO if split else 0, split=0 if split else 2))
    whole = torch.empty(B, T, C, device=dev, dtype=OP16)
    run(False, whole)
    first = torch.empty_like(whole)
    run(True, first)
    torch.cuda.synchronize()
    d = (first.float() - whole.float()).abs().max().item()
    assert d < 2e-2 * whole.float().abs().max().item(), d     # pieces add in piece order: last-bit differences only
    out = torch.empty_like(whole)
    bad = 0
    for it in range(200):
        out.fill_(float("nan"))
        run(True, out)
        if it % 20 == 19:
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, first))
            assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, "tickets not back at zero"
    torch.cuda.synchronize()
    bad += int(not torch.equal(out, first))
    assert bad == 0, f"{bad} of the sampled split launches differ from the first"


@pytest.mark.parametrize("case", ["all_negative", "late_spike_2^40", "growing_2^90", "spike_past_2^100", "first_tile_spike"])
@pytest.mark.parametrize("variant", [26, 27], ids=["mfma32x32x16", "mfma16x16x32"])
def test_flash_attn64_hand_placed_reference_paths(dev, case, variant):
    """Variants 26 / 27 keep ONE reference per query (the first key tile's row maximum):
      all_negative     - logits around -60: the reference is the first tile's maximum whatever its sign;
      late_spike_2^40  - some rows' largest logit sits in the last tile, 2^40 above the first tile's: probabilities up to 2^40
                         in the bf16 P operand and the fp32 sums, no fallback;
      growing_2^90     - the row maximum grows by ~6 log2 units per tile over 16 tiles (still below the 2^100 bound);
      spike_past_2^100 - a few rows exceed their reference by ~2^115: their row sums pass 2^100 and the workgroup redoes its
                         256 queries with the running-maximum loop (the other workgroups do not);
      first_tile_spike - the maximum is in tile 0: everything else underflows towards 0 relative to it."""
    from marigold_amd import ops
    T, C = 1024, 64
    g = torch.Generator().manual_seed(9)
    q = torch.randn(1, T, C, generator=g)
    k = torch.randn(1, T, C, generator=g)
    v = _bf(torch.randn(1, T, C, generator=g))
    u = torch.nn.functional.normalize(torch.randn(C, generator=g), dim=0)
    tile = (torch.arange(T) // 64).float()
    if case == "all_negative":
        q = q * 0.2 + u * 22.0
        k = k * 0.2 - u * 22.0
    elif case == "late_spike_2^40":
        for i, row in enumerate((3, 300, 301, 777, 1023)):
            k[0, 1000 + 4 * i] = q[0, row] * (28.0 * 8.0 / q[0, row].pow(2).sum())     # logit 28 = 2^40.4
    elif case == "growing_2^90":
        q = q * 0.1 + u * 8.0
        k = k * 0.3 + u[None, :] * (tile * 4.1)[:, None]     # logit +4.1 (5.9 log2 units) per tile: 2^89 at tile 15
    elif case == "spike_past_2^100":
        for i, row in enumerate((5, 6, 400)):
            k[0, 900 + 8 * i] = q[0, row] * (80.0 * 8.0 / q[0, row].pow(2).sum())      # logit 80 = 2^115
    else:
        for i, row in enumerate((10, 500, 900)):
            k[0, 7 + 9 * i] = q[0, row] * (60.0 * 8.0 / q[0, row].pow(2).sum())
    q, k = _bf(q), _bf(k)
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    assert torch.isfinite(ref).all()
    qkv = torch.cat([q, k], dim=-1).to(dev, OP16)
    vtp = ops.permute_vt_keys(v.permute(0, 2, 1).contiguous().to(dev, OP16))
    for split in (False, True):
        ws = torch.zeros(ops.FLASH_WS_BYTES, dtype=torch.uint8, device=dev) if split else None
        out = torch.full((1, T, C), float("nan"), device=dev, dtype=OP16)
        _run(ops.flash_attn64(qkv, qkv[:, :, C:], vtp, out, B=1, heads=1, Ntok=T, ldq=2 * C, ldo=C, ldvt=T, sq=0, sk=0, svt=0, so=0,
                              scale=0.125, variant=variant, vt_perm=True, ws=ws, ws_bytes=ops.FLASH_WS_BYTES if split else 0, split=1 if split else 2))
        _close(f"flash_attn64/hand-placed/v{variant}/{case}/split{int(split)}", out, ref)


def _flash512_run(dev, q, k, v, B, T):
    from marigold_amd import ops
    C = 512
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    assert torch.isfinite(ref).all()
    ldvt = ((T + 63) // 64) * 64
    qk = torch.zeros(B * T + 8, 2 * C, device=dev, dtype=OP16)   # the engine's layout: [Q | K] rows, slack rows
    qk[:B * T] = torch.cat([q, k], dim=-1).reshape(B * T, 2 * C).to(dev, OP16)
    vt = torch.zeros(B, C, ldvt, device=dev, dtype=OP16)
    vt[:, :, :T] = v.permute(0, 2, 1).to(dev, OP16)
    out = torch.full((B, T, C), float("nan"), device=dev, dtype=OP16)
    _run(ops.flash_attn512(qk, qk.data_ptr() + C * 2, vt, out, B=B, Ntok=T, ldq=2 * C, ldo=C, ldvt=ldvt,
                           sq=T * 2 * C, sk=T * 2 * C, svt=C * ldvt, so=T * C, scale=1.0 / math.sqrt(C)))
    return out, ref


@pytest.mark.parametrize("B,T", [(1, 32), (1, 31), (2, 130), (3, 576), (1, 1000), (2, 2304), (13, 1984), (7, 3700)])
def test_flash_attn512(dev, B, T):
    """MG_OP_FLASH_ATTN512 (the VAE mid-block attention, one head of width 512) against fp32 SDPA on the host: whole and
    ragged key tiles, ragged query blocks, several images per launch."""
    g = torch.Generator().manual_seed(T)
    q = _bf(torch.randn(B, T, 512, generator=g) * 1.5)
    k = _bf(torch.randn(B, T, 512, generator=g) * 1.5)
    v = _bf(torch.randn(B, T, 512, generator=g))
    out, ref = _flash512_run(dev, q, k, v, B, T)
    _close(f"flash_attn512/B{B}T{T}", out, ref)


@pytest.mark.parametrize("case", ["all_negative", "late_spike", "growing_below_retry", "retry_some_rows", "retry_all_rows"])
def test_flash_attn512_reference_paths(dev, case):
    """The kernel's softmax runs against a fixed per-query reference (the first key tile's maximum) and redoes a block of
    128 queries with the true maxima when a later score tops its reference by more than 2^60:
      all_negative        - every logit around -60 (log2 units: -87): the reference must be the first tile's maximum whatever
                            its sign;
      late_spike          - the largest logit of some rows sits in the last (ragged) tile, ~2^30 above the first tile's;
      growing_below_retry - the row maximum grows by ~3 log2 units per tile up to ~2^55 times the reference: probabilities
                            up to 2^55 in the bf16 P operand and the fp32 sums, no retry;
      retry_some_rows     - a few rows exceed the reference by ~2^80 (their block is redone; its other rows must come out
                            the same);
      retry_all_rows      - every row does."""
    T, C = 1000, 512   # 32 tiles, the last one ragged (8 keys); 8 query blocks, the last one ragged
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, T, C, generator=g)
    k = torch.randn(1, T, C, generator=g)
    v = _bf(torch.randn(1, T, C, generator=g))
    u = torch.nn.functional.normalize(torch.randn(C, generator=g), dim=0)
    sc = math.sqrt(C)   # logit = q.k / sqrt(C)
    tile = (torch.arange(T) // 32).float()
    if case == "all_negative":
        q = q * 0.2 + u * 37.0
        k = k * 0.2 - u * 37.0                     # q.k ~ -1369, logit ~ -60
    elif case == "late_spike":
        k[0, 999] = q[0, 500] * (21.0 * sc / q[0, 500].pow(2).sum())     # logit 21 (2^30) for query 500 at the last key
        k[0, 996] = q[0, 3] * (25.0 * sc / q[0, 3].pow(2).sum())
        k[0, 2] = q[0, 900] * (25.0 * sc / q[0, 900].pow(2).sum())       # and one inside tile 0
    elif case == "growing_below_retry":
        q = q * 0.1 + u * 8.0
        k = k * 0.3 + u[None, :] * (tile * (1.75 * 0.6931472 * sc / 8.0))[:, None]   # +1.75 log2 units per tile: 2^54 at tile 31
    elif case == "retry_some_rows":
        for i, row in enumerate((5, 130, 131, 640, 999)):
            k[0, 700 + 3 * i] = q[0, row] * (56.0 * sc / q[0, row].pow(2).sum())   # logit 56 = 2^80.8
    else:
        q = q * 0.1 + u * 8.0
        k = k * 0.3 + u[None, :] * (tile * (2.0 * sc / 8.0))[:, None]   # +2.0 (2.9 log2 units) per tile: 2^89 over the sequence
    q, k = _bf(q), _bf(k)
    out, ref = _flash512_run(dev, q, k, v, 1, T)
    _close(f"flash_attn512/{case}", out, ref)


def test_softmax_rows(dev):
    from marigold_amd import ops
    g = torch.Generator().manual_seed(1)
    R, n, lds, ldp = 33, 1000, 1000, 1024
    s = torch.randn(R, lds, generator=g) * 3
    ref = torch.softmax(s[:, :n], dim=-1)
    p = torch.full((R, ldp), float("nan"), device=dev, dtype=OP16)
    _run(ops.softmax_rows(s.to(dev), p, R=R, ncols=n, lds=lds, ldp=ldp))
    _close("softmax_rows", p[:, :n], ref)
    assert (p[:, n:] == 0).all()


def test_small_ops(dev):
    from marigold_amd import ops
    g = torch.Generator().manual_seed(6)
    x, m, nz = (torch.randn(3, 4, 8, 8, generator=g) for _ in range(3))
    o = torch.empty(3, 4, 8, 8, device=dev)
    _run(ops.sched_step(x.to(dev), m.to(dev), nz.to(dev), o, n=x.numel(), cx=0.7, cm=-0.3, cn=0.2))
    _close("sched_step", o, 0.7 * x - 0.3 * m + 0.2 * nz, tol=1e-6)
    _run(ops.sched_step(x.to(dev), m.to(dev), None, o, n=x.numel(), cx=0.5, cm=0.25))
    _close("sched_step/no-noise", o, 0.5 * x + 0.25 * m, tol=1e-6)
    # the same update as the tail of conv_out's pointwise pass (MG_POST_SCHED): in = [B*HW][8] padded GEMM output
    from marigold_amd import _lib as L
    tmp = torch.zeros(3 * 64, 8)
    tmp[:, :4] = m.permute(0, 2, 3, 1).reshape(-1, 4)
    xd, nzd = x.to(dev).clone(), nz.to(dev)
    _run(ops.post_nchw(tmp.to(dev), xd, B=3, HW=64, Cout=4, ldi=8, post=L.POST_SCHED, noise=nzd, cx=0.7, cm=-0.3, cn=0.2))
    _close("post_nchw/scheduler tail", xd, 0.7 * x - 0.3 * m + 0.2 * nz, tol=1e-6)
    xi = torch.randn(5, 320, generator=g)
    w = torch.randn(100, 320, generator=g) / 18
    bb = torch.randn(100, generator=g)
    o2 = torch.empty(5, 100, device=dev)
    _run(ops.linear_small_m(xi.to(dev), w.to(dev), bb.to(dev), o2, M=5, N=100, K=320, act_in=1, act_out=1))
    _close("linear_small_m", o2, F.silu(F.silu(xi) @ w.t() + bb), tol=1e-5)
    # more rows than one row block (16), the last block partial; a strided output (a column window of a wider table);
    # a row's value does not depend on how many rows share the launch
    xi = torch.randn(37, 320, generator=g)
    o3 = torch.full((37, 164), float("nan"), device=dev)
    _run(ops.linear_small_m(xi.to(dev), w.to(dev), bb.to(dev), o3[:, 64:], M=37, N=100, K=320, act_in=1, ldo=164))
    _close("linear_small_m/row blocks", o3[:, 64:], F.silu(xi) @ w.t() + bb, tol=1e-5)
    assert torch.isnan(o3[:, :64]).all()
    o4 = torch.empty(1, 100, device=dev)
    _run(ops.linear_small_m(xi[20:21].to(dev), w.to(dev), bb.to(dev), o4, M=1, N=100, K=320, act_in=1))
    assert torch.equal(o4[0], o3[20, 64:])
    lat = torch.randn(2, 4, 6, 5, generator=g)
    w1 = torch.randn(4, 4, generator=g)
    b1 = torch.randn(4, generator=g)
    o3 = torch.empty(2, 4, 6, 5, device=dev)
    _run(ops.latent_1x1(lat.to(dev), w1.to(dev), b1.to(dev), o3, B=2, Ci=4, Co=4, HW=30, scale=1 / 0.18215))
    _close("latent_1x1", o3, F.conv2d(lat / 0.18215, w1[:, :, None, None], b1), tol=1e-5)


# --------------------------------------------------------------------------- ensembling kernels
def test_ensemble_kernels_vs_oracle(dev, golden_dir):
    import os
    from marigold_amd import ops
    from oracle import ensemble as oens
    gold = np.load(os.path.join(golden_dir, "ensemble_ref.npz"))
    d = torch.from_numpy(gold["d_e10_in"])          # [10,1,32,40]
    E, HW = d.shape[0], d.shape[2] * d.shape[3]
    dd = d.reshape(E, HW).to(dev)
    scratch = torch.empty(128 * E * 35, device=dev, dtype=torch.float64)
    stats = torch.empty(3 * E + E * E, device=dev, dtype=torch.float64)
    _run(ops.ens_depth_stats(dd, scratch, stats, E=E, HW=HW))
    st = stats.cpu()
    flat = d.reshape(E, HW).double()
    np.testing.assert_allclose(st[:E], flat.min(1).values, rtol=0, atol=0)
    np.testing.assert_allclose(st[E:2 * E], flat.max(1).values, rtol=0, atol=0)
    np.testing.assert_allclose(st[2 * E:3 * E], flat.mean(1), atol=1e-7)
    cen = flat - flat.mean(1, keepdim=True)
    np.testing.assert_allclose(st[3 * E:].reshape(E, E), cen @ cen.t() / HW, atol=1e-8)
    # align + median + MAD + min/max vs the oracle helpers, for arbitrary (s, t)
    g = torch.Generator().manual_seed(0)
    param = np.concatenate([1 + 0.3 * torch.rand(E, generator=g).numpy(), 0.1 * torch.randn(E, generator=g).numpy()])
    al = oens.depth_align(d, param, True, True)
    pred, mad = oens.depth_reduce(al, "median", True)
    stv = torch.from_numpy(param).float().to(dev)
    med = torch.empty(HW, device=dev)
    madd = torch.empty(HW, device=dev)
    mm = torch.empty(2 + 2 * 10, device=dev)
    scr = torch.empty(12288, device=dev, dtype=torch.uint8)
    _run(ops.ens_depth_median(dd, stv, med, madd, mm, scr, E=E, HW=HW))
    nbad = (med.cpu() != pred.reshape(-1)).sum().item()
    print(f"[parity] ens_depth_median: {nbad} / {HW} pixels differ from the oracle, "
          f"max|diff| {(med.cpu() - pred.reshape(-1)).abs().max().item():.3e}")
    assert torch.equal(med.cpu(), pred.reshape(-1)), "median must be bit-exact (fp32, same op order)"
    assert torch.equal(madd.cpu(), mad.reshape(-1))
    assert mm[0].item() == pred.min().item() and mm[1].item() == pred.max().item()
    pmin, pmax = int(pred.reshape(-1).argmin()), int(pred.reshape(-1).argmax())
    assert torch.equal(mm[2:2 + E].cpu(), d.reshape(E, HW)[:, pmin])
    assert torch.equal(mm[2 + E:2 + 2 * E].cpu(), d.reshape(E, HW)[:, pmax])
    _run(ops.ens_depth_norm(med, madd, mm, HW=HW))
    rng = (pred.max() - pred.min()).clamp(min=1e-6)
    _close("ens_depth_norm", med, ((pred - pred.min()) / rng).reshape(-1), tol=1e-6)
    # mean / std reduction, odd E, no-alignment path
    d3 = torch.from_numpy(gold["d_e3_in"])
    E3, HW3 = 3, d3.shape[2] * d3.shape[3]
    pm, ps = oens.depth_reduce(d3, "mean", True)
    med3 = torch.empty(HW3, device=dev)
    std3 = torch.empty(HW3, device=dev)
    _run(ops.ens_depth_median(d3.reshape(E3, HW3).to(dev), None, med3, std3, mm, scr, E=E3, HW=HW3, reduction=1))
    _close("ens_depth_mean", med3, pm.reshape(-1), tol=1e-6)
    _close("ens_depth_std", std3, ps.reshape(-1), tol=1e-5)
    # normals vs the reference's own outputs
    for name in ("n_e4", "n_e10"):
        n = torch.from_numpy(gold[f"{name}_in"])
        E, H, W = n.shape[0], n.shape[2], n.shape[3]
        nd = n.reshape(E, 3, H * W).to(dev)
        out = torch.empty(3, H * W, device=dev)
        unc = torch.empty(H * W, device=dev)
        _run(ops.ens_normals(nd, out, unc, E=E, HW=H * W))
        same = (out.cpu().reshape(3, H, W) == torch.from_numpy(gold[f"{name}_closest"])[0]).all(0).float().mean()
        print(f"[parity] ens_normals/{name}: identical closest member on {same.item() * 100:.2f}% of pixels")
        assert same.item() >= 0.999
        _close(f"ens_normals_unc/{name}", unc, torch.from_numpy(gold[f"{name}_unc"]).reshape(-1), tol=1e-5)
        _run(ops.ens_normals(nd, out, None, E=E, HW=H * W, reduction=1))
        _close(f"ens_normals_mean/{name}", out, torch.from_numpy(gold[f"{name}_mean"]).reshape(3, -1), tol=1e-5)


def test_program_and_graph_replay(dev):
    """A 3-op program replayed natively and as a captured hipGraph gives identical output."""
    from marigold_amd import ops
    g = torch.Generator().manual_seed(8)
    B, HW, C = 2, 64, 64
    x = torch.randn(B, HW, C, generator=g).to(dev, OP16)
    part = torch.empty(B, 4, C, 2, device=dev)
    ss = torch.empty(B, 2, C, device=dev)
    out = torch.zeros_like(x)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    seq = ops.OpSeq("gn", f16=F16)
    seq.add(ops.gn_stats(x, part, B=B, HW=HW, C=C, chunks=4, groups=32))
    seq.add(ops.gn_finalize(part, gamma, beta, ss, B=B, C=C, groups=32, slots=4, HW=HW, eps=1e-5))
    seq.add(ops.gn_apply(x, ss, out, B=B, HW=HW, C=C, silu=True))
    seq.run()
    torch.cuda.synchronize()
    ref = out.clone()
    out.zero_()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        seq.capture()
        out.zero_()
        seq.run()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    ms = seq.profile()
    assert len(ms) == 3 and all(m >= 0 for m in ms)


HEAD_CASES = [
    # B, H, W, C, Cout, normalise, silu
    (2, 20, 37, 64, 3, True, True),        # ragged 16 x 16 tiles
    (1, 96, 96, 320, 4, True, True),       # the UNet's conv_norm_out -> conv_out at one member
    (3, 33, 16, 32, 1, True, False),
    (1, 64, 80, 128, 2, False, False),     # no normalisation: a plain conv3x3 to 2 channels
    (4, 256, 256, 64, 3, True, True),      # >= 512 tiles of 16 x 32: two pixels per thread
    (5, 250, 300, 32, 3, True, True),      # the same, ragged in both directions
    (2, 256, 512, 128, 4, True, True),
]


@pytest.mark.parametrize("case", HEAD_CASES, ids=[f"B{c[0]}_{c[1]}x{c[2]}_C{c[3]}_to{c[4]}" + ("" if c[5] else "_raw") for c in HEAD_CASES])
def test_conv3x3_head(dev, case):
    """MG_OP_CONV3X3_HEAD (conv_norm_out -> SiLU -> conv_out in one launch) vs torch fp32 on the same bf16-rounded operands:
    the normalised input is rounded to bf16 exactly where the materialising pass (MG_OP_GN_APPLY) rounds it."""
    from marigold_amd import ops, weights as Wm
    B, H, W, C, Cout, norm, silu = case
    g = torch.Generator().manual_seed(B * 1000 + H + C)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3)
    w = _bf(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C))
    bias = torch.randn(Cout, generator=g) * 0.1
    scale = 1.0 + 0.3 * torch.randn(B, C, generator=g)
    shift = 0.3 * torch.randn(B, C, generator=g)
    h = x
    if norm:
        h = x * scale[:, :, None, None] + shift[:, :, None, None]
        if silu:
            h = F.silu(h)
        h = _bf(h)
    ref = F.conv2d(h, w, bias, padding=1)
    npad = 8
    wd = torch.zeros(npad, 9 * C)
    wd[:Cout] = Wm.pack_conv3x3(w)
    out = torch.full((B * H * W, npad), float("nan"), device=dev)
    ss = torch.stack([scale, shift], dim=1).contiguous().to(dev) if norm else None
    _run(ops.conv3x3_head(_nhwc(x).to(dev, OP16), ss, wd.to(dev, OP16), bias.to(dev), out,
                          B=B, H=H, W=W, C=C, Cout=Cout, ldo=npad, silu=silu))
    got = out[:, :Cout].reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    # (SiLU through v_exp / v_rcp before the bf16 rounding: a rounding boundary may fall the other way on single elements)
    _close(f"conv3x3_head {case}", got, ref, tol=4e-3)
    assert torch.isnan(out[:, Cout:]).all()   # the padding columns are not touched


def test_conv3x3_head_rejects_what_it_cannot_run(dev):
    from marigold_amd import ops, _lib as L
    x = torch.zeros(1, 8, 8, 48, device=dev, dtype=OP16)
    w = torch.zeros(8, 9 * 48, device=dev, dtype=OP16)
    out = torch.zeros(64, 8, device=dev)
    with pytest.raises(L.MarigoldHipError):   # C not a multiple of 32
        _run(ops.conv3x3_head(x, None, w, None, out, B=1, H=8, W=8, C=48, Cout=3, ldo=8))
    x = torch.zeros(1, 8, 8, 64, device=dev, dtype=OP16)
    w = torch.zeros(8, 9 * 64, device=dev, dtype=OP16)
    with pytest.raises(L.MarigoldHipError):   # more than 4 output channels
        _run(ops.conv3x3_head(x, None, w, None, out, B=1, H=8, W=8, C=64, Cout=5, ldo=8))


def test_small_cout_conv_on_mfma_path(dev):
    """conv3x3 to <= 4 fp32 NCHW channels = GEMM into a padded fp32 [M][8] buffer (tile variant 29)
    + MG_OP_POST_NCHW with the pipeline's pointwise tails (depth mean/clip/shift, normals L2)."""
    from marigold_amd import _lib as L, ops, weights as Wm
    g = torch.Generator().manual_seed(12)
    B, H, W, Cin = 2, 20, 28, 128
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    for cout, post in ((3, L.POST_DEPTH), (3, L.POST_NORMALS), (4, L.POST_NONE)):
        w = _bf(torch.randn(cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
        b = torch.randn(cout, generator=g) * 0.1
        y = F.conv2d(x, w, b, padding=1) * 0.7
        if post == L.POST_DEPTH:
            ref = (y.mean(1, keepdim=True).clamp(-1, 1) + 1) / 2
        elif post == L.POST_NORMALS:
            yc = y.clamp(-1, 1)
            ref = yc / yc.norm(dim=1, keepdim=True).clamp(min=1e-6)
        else:
            ref = y
        w8 = torch.zeros(8, 9 * Cin)
        w8[:cout] = Wm.pack_conv3x3(w)
        b8 = torch.zeros(8)
        b8[:cout] = b
        tmp = torch.full((B * H * W, 8), float("nan"), device=dev)
        out = torch.full(ref.shape, float("nan"), device=dev)
        _run(ops.igemm(_nhwc(x).to(dev, OP16), w8.to(dev, OP16), tmp, B=B, H=H, W=W, Cin=Cin,
                       Ho=H, Wo=W, N=8, taps=9, stride=1, pad=1, bias=b8.to(dev), epi=L.EPI_F32, ldo=8))
        _run(ops.post_nchw(tmp, out, B=B, HW=H * W, Cout=cout, ldi=8, post=post, scale=0.7))
        _close(f"small_cout_mfma/c{cout}p{post}", out, ref, tol=2e-2)


def test_small_cin_conv_on_mfma_path(dev):
    """conv3x3 from <= 8 fp32 NCHW channels (two sources, first one broadcast) = MG_OP_IM2COL_SMALL +
    GEMM with K padded to 64 / 128."""
    from marigold_amd import ops, weights as Wm
    g = torch.Generator().manual_seed(9)
    for (B, H, W, C0, C1, Cout, bcast) in ((3, 10, 14, 4, 4, 64, True), (2, 9, 7, 4, 4, 320, False),
                                           (1, 12, 20, 3, 0, 128, False), (2, 8, 16, 4, 0, 64, False)):
        a = torch.randn(1 if bcast else B, C0, H, W, generator=g)
        b = torch.randn(B, C1, H, W, generator=g) if C1 else None
        w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))
        bias = torch.randn(Cout, generator=g) * 0.1
        xin = a.expand(B, -1, -1, -1) if bcast else a
        if C1:
            xin = torch.cat([xin, b], dim=1)
        ref = F.conv2d(_bf(xin), _bf(w), bias, padding=1)
        k = 9 * (C0 + C1)
        kp = 64 if k <= 64 else 128
        wp = torch.zeros(Cout, kp)
        wp[:, :k] = Wm.pack_conv3x3(_bf(w))
        col = torch.full((B * H * W, kp), float("nan"), device=dev, dtype=OP16)
        out = torch.full((B, H, W, Cout), float("nan"), device=dev, dtype=OP16)
        _run(ops.im2col_small(a.to(dev), b.to(dev) if C1 else None, col, B=B, H=H, W=W, C0=C0, C1=C1, Kp=kp,
                              bcast0=bcast))
        _run(ops.linear(col, wp.to(dev, OP16), out, M=B * H * W, K=kp, N=Cout, bias=bias.to(dev)))
        _close(f"small_cin_mfma/{C0}+{C1}->{Cout}", out.float().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("mode", ["bilinear", "bicubic", "nearest-exact"])
def test_resize_vs_torch_cpu(dev, mode):
    """MG_OP_RESIZE vs torch's CPU F.interpolate(antialias=True) (what torchvision's resize runs): down-
    and up-sampling, one axis only, uint8 round trip (reference: image_util.py:90-120, :306-312)."""
    from marigold_amd.util.image_util import InterpolationMode, resize
    g = torch.Generator().manual_seed(4)
    im = {m.value: m for m in InterpolationMode}[mode]
    cases = [((1, 3, 96, 128), (72, 96)), ((1, 3, 50, 70), (100, 140)), ((1, 1, 64, 48), (64, 31)),
             ((2, 3, 33, 40), (77, 40)), ((1, 3, 480, 640), (576, 768)), ((1, 1, 768, 576), (480, 360))]
    for shape, size in cases:
        xf = torch.rand(shape, generator=g) * 2 - 1
        xu = (torch.rand(shape, generator=g) * 255).round().to(torch.uint8)
        for x in (xf, xu):
            ref = resize(x, size, im)            # host path = torch CPU
            got = resize(x.to(dev), size, im).cpu()
            assert got.shape == ref.shape and got.dtype == ref.dtype
            if mode == "nearest-exact":
                assert torch.equal(got, ref)
            elif x.dtype == torch.uint8:
                d = (got.int() - ref.int()).abs()
                assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3, (shape, size, int(d.max()))
            else:
                assert float((got - ref).abs().max()) < 2e-5, (shape, size, float((got - ref).abs().max()))


@pytest.mark.parametrize("cmap", ["Spectral", "viridis"])
def test_colorize_lut_matches_reference_chain(dev, cmap):
    """MG_OP_COLORIZE vs the reference's colorize_depth_maps + (x * 255).astype(uint8) (marigold/util/image_util.py:38-76,
    marigold_depth_pipeline.py:318-327) restated on matplotlib: bit-exact uint8 image."""
    from marigold_amd.util import image_util as iu
    g = torch.Generator().manual_seed(3)
    d = torch.rand(37, 53, generator=g)
    d[0, 0], d[0, 1], d[0, 2] = 1.0, 0.0, 1.5    # ends of the table and a value the clip must catch
    ref = (iu.colorize_depth_maps(d.numpy(), 0, 1, cmap=cmap).squeeze() * 255).astype(np.uint8)   # [3,H,W]
    got = iu.colorize_depth_device(d.to(dev), 0.0, 1.0, cmap=cmap).cpu().numpy()                  # [H,W,3]
    assert got.shape == (37, 53, 3) and np.array_equal(np.moveaxis(got, -1, 0), ref)
