"""GPU parity tests of EXPERIMENTAL kernels that no product path selects (run with MG_EXPERIMENTAL=1).
Currently: the halo-shared 3x3 convolution tile (csrc/igemm3.hip, MG_OP_IGEMM tile variants 70-73)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.environ.get("MG_EXPERIMENTAL"), reason="experimental kernels: set MG_EXPERIMENTAL=1")]


def _bf(t):
    return t.to(torch.bfloat16).float()


HALO_CASES = [
    # name, B, H, W, Cin, Cout, residual, rowvec, fp32
    ("one_tile", 1, 12, 20, 64, 128, False, False, False),
    ("spans_images_medge", 3, 9, 31, 128, 192, True, True, False),     # M = 837: tiles cross images, partial last tile
    ("n_edge_320", 2, 16, 24, 192, 320, True, False, False),
    ("narrow_rows", 2, 40, 3, 64, 64, False, True, False),              # W = 3: most pixels touch a row end
    ("single_column", 2, 300, 1, 64, 64, False, False, False),          # W = 1: kx = 0 and 2 are always padding
    ("fp32_out", 1, 24, 24, 256, 128, False, False, True),
    ("deep_k", 1, 13, 12, 1280, 256, False, False, False),
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[c[0] for c in HALO_CASES])
def test_halo_conv3x3(case):
    from marigold_amd import _lib as L, ops, weights as Wm
    name, B, H, W, Cin, Cout, use_res, use_rv, f32 = case
    dev = torch.device("cuda:0")
    L.init(0)
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    rv = torch.randn(B, Cout, generator=g) * 0.1 if use_rv else None
    res = _bf(torch.randn(B, Cout, H, W, generator=g)) if use_res else None
    ref = F.conv2d(x, w, b, padding=1)
    if rv is not None:
        ref = ref + rv[:, :, None, None]
    if res is not None:
        ref = ref + res
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()   # noqa: E731
    M = B * H * W
    outs = {}
    for variant in (70, 71, 72, 73, 23):   # the experimental tiles (256x128 / 256x256, burst / split issue) and a validated one
        out = torch.full((M, Cout), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        op = ops.igemm(nhwc(x).to(dev, torch.bfloat16), Wm.pack_conv3x3(w).to(dev, torch.bfloat16), out, B=B, H=H, W=W,
                       Cin=Cin, Ho=H, Wo=W, N=Cout, taps=9, stride=1, pad=1, bias=b.to(dev),
                       rowvec=None if rv is None else rv.to(dev).contiguous(),
                       residual=None if res is None else nhwc(res).reshape(M, Cout).to(dev, torch.bfloat16),
                       epi=L.EPI_F32 if f32 else L.EPI_BF16, variant=variant)
        for _ in range(3 if variant >= 70 else 1):
            ops.launch(op)
        torch.cuda.synchronize()
        outs[variant] = out.float().cpu()
    want = nhwc(ref).reshape(M, Cout)
    for variant, got in outs.items():
        assert torch.isfinite(got).all(), f"{name}/v{variant}: non-finite"
        err = (got - want).abs().max().item()
        scale = want.abs().max().item()
        print(f"[parity] halo/{name}/v{variant}: max|err| {err:.3e} (scale {scale:.2f})")
        assert err <= (2e-3 if f32 else 1.5e-2) * scale, f"{name}/v{variant}"


def test_igemm_128x320_bk32_two_workgroups_per_cu():
    """Tile variant 49 (128x320, 32-deep K tiles, 4 waves): a new instantiation of the generation-2 template for the
    K = C linears of the 320-channel level - against the validated full-width tile (46) on the same buffers."""
    from marigold_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    L.init(0)
    g = torch.Generator().manual_seed(3)
    for M, K, N in ((300, 320, 320), (4096, 1280, 320), (1000, 64, 320)):
        x = _bf(torch.randn(M, K, generator=g))
        w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
        b = torch.randn(N, generator=g) * 0.1
        res = _bf(torch.randn(M, N, generator=g))
        ref = x @ w.t() + b + res
        outs = {}
        for v in (49, 46):
            out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            ops.launch(ops.linear(x.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16), out, M=M, K=K, N=N, bias=b.to(dev),
                                  residual=res.to(dev, torch.bfloat16), variant=v))
            torch.cuda.synchronize()
            outs[v] = out.float().cpu()
            err = (outs[v] - ref).abs().max().item()
            print(f"[parity] linear {M}x{K}x{N}/v{v}: max|err| {err:.3e}")
            assert torch.isfinite(outs[v]).all() and err <= 1.5e-2 * ref.abs().max().item()
        assert torch.equal(outs[49], outs[46])   # same K order per accumulator -> identical bits
