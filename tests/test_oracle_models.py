"""Structural + golden checks of the oracle's UNet / VAE / scheduler restatements."""
import math
import os

import numpy as np
import pytest
import torch

from marigold_amd import synthetic as syn
from marigold_amd.arch import (TINY_UNET, TINY_VAE, UNetConfig, unet_param_shapes,
                               unet_up_resnet_channels, vae_param_shapes)
from oracle import pipeline as opipe
from oracle.schedulers import DDIMScheduler, LCMScheduler
from oracle.sd2_unet import UNet2DConditionModel
from oracle.sd2_vae import AutoencoderKL


def _count(d, *prefixes):
    return sum(math.prod(s) for k, s in d.items() if k.startswith(prefixes))


def test_param_counts_match_published_sd2():
    u, v = unet_param_shapes(), vae_param_shapes()
    assert round(_count(u, "") / 1e6, 2) == 865.92
    assert round(_count(v, "encoder", "quant_conv") / 1e6, 2) == 34.16
    assert round(_count(v, "decoder", "post_quant_conv") / 1e6, 2) == 49.49


def test_state_dict_keys_match_arch_tables():
    with torch.device("meta"):
        m, a = UNet2DConditionModel(), AutoencoderKL()
    assert {k: tuple(p.shape) for k, p in m.state_dict().items()} == dict(unet_param_shapes())
    assert {k: tuple(p.shape) for k, p in a.state_dict().items()} == dict(vae_param_shapes())


def test_skip_stack_widths():
    rows = unet_up_resnet_channels(UNetConfig())
    assert [r[2] + r[3] for r in rows] == [2560, 2560, 2560, 2560, 2560, 1920,
                                            1920, 1280, 960, 960, 640, 640]


def test_scheduler_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "scheduler_tables.npz"))
    for n in (10, 4, 1):
        s = DDIMScheduler()
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"ddim_trailing_{n}"].tolist()
    for n in (10, 1):
        s = DDIMScheduler(timestep_spacing="leading", rescale_betas_zero_snr=False)
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"ddim_leading_{n}"].tolist()
    for n in (4, 1):
        s = LCMScheduler()
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"lcm_{n}"].tolist()
    z = DDIMScheduler()
    assert float(z.alphas_cumprod[999]) == 0.0          # zero terminal SNR
    assert abs(float(z.alphas_cumprod[0]) - 0.99915) < 1e-4


def test_zero_snr_first_step_is_minus_v():
    s = DDIMScheduler()
    s.set_timesteps(1)
    x, v = torch.randn(1, 4, 4, 4), torch.randn(1, 4, 4, 4)
    out = s.step(v, 999, x)
    torch.testing.assert_close(out.pred_original_sample, -v)


@pytest.fixture(scope="module")
def tiny():
    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(min(16, usable_cores()))
    unet = UNet2DConditionModel(block_out_channels=TINY_UNET.block_out_channels,
                                attention_head_dim=TINY_UNET.heads,
                                cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict(TINY_UNET))
    vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels).eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict(TINY_VAE))
    return unet, vae


def test_tiny_pipeline_golden(tiny, golden_dir):
    unet, vae = tiny
    g = np.load(os.path.join(golden_dir, "tiny_pipeline.npz"))
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    img = syn.synthetic_image(64, 128, seed=0)
    lat0 = syn.synthetic_latents(3, 8, 16, seed=2024)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    with torch.no_grad():
        rl = opipe.encode_rgb(vae, rgb)
        np.testing.assert_allclose(rl.numpy(), g["rgb_latent"], atol=1e-4)
        d = opipe.single_infer("depth", unet, vae, DDIMScheduler(), rgb.expand(3, -1, -1, -1),
                               lat0, ctx, 4)
        np.testing.assert_allclose(d.numpy(), g["depth_ddim4"], atol=1e-3)
        gen = torch.Generator("cpu").manual_seed(99)
        dl = opipe.single_infer("depth", unet, vae, LCMScheduler(), rgb, lat0[:1], ctx, 3,
                                generator=gen)
        np.testing.assert_allclose(dl.numpy(), g["depth_lcm3"], atol=1e-3)


def test_unet_odd_latent_size_uses_upsample_size(tiny):
    unet, _ = tiny
    x = torch.randn(1, 8, 12, 20)
    y = unet(x, torch.tensor(500), torch.randn(1, 2, TINY_UNET.cross_attention_dim)).sample
    assert y.shape == (1, 4, 12, 20)


def test_oracle_pinned_against_diffusers(golden_dir):
    """The UNet / VAE / scheduler restatements against diffusers' own outputs (tests/golden/diffusers_pin.npz, written by
    ``python oracle/pin_against_diffusers.py`` wherever diffusers>=0.25.0 imports - reference requirement,
    /root/reference/requirements.txt:2, call sites marigold_depth_pipeline.py:461-468, 491-492, 512-513).  Until that file
    exists the oracle is PARITY UNPINNED for these modules (structure and parameter counts only): said loudly, not failed."""
    import warnings
    path = os.path.join(golden_dir, "diffusers_pin.npz")
    if not os.path.exists(path):
        msg = ("PARITY UNPINNED: oracle/sd2_unet.py, sd2_vae.py and schedulers.py have not been checked against diffusers "
               "(tests/golden/diffusers_pin.npz is absent; one command where diffusers imports: python oracle/pin_against_diffusers.py)")
        warnings.warn(msg)
        pytest.skip(msg)
    pin = np.load(path)
    assert bool(pin["all_within_tolerance"]), {k: float(pin[k]) for k in pin.files if k.startswith("err/")}
    tol = float(pin["tolerance"])
    usd, vsd = syn.synthetic_unet_state_dict(TINY_UNET), syn.synthetic_vae_state_dict(TINY_VAE)
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    g = torch.Generator().manual_seed(4242)
    with torch.no_grad():
        unet = UNet2DConditionModel(in_channels=TINY_UNET.in_channels, out_channels=TINY_UNET.out_channels,
                                    block_out_channels=TINY_UNET.block_out_channels, attention_head_dim=TINY_UNET.heads,
                                    cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
        unet.load_state_dict(usd)
        x = torch.randn(2, TINY_UNET.in_channels, 8, 16, generator=g)
        for t in (999, 500, 1):
            ref = torch.from_numpy(pin[f"ref/tiny/unet/t{t}"])
            got = unet(x, torch.tensor(t), ctx.expand(2, -1, -1)).sample
            assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), t
        vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels, layers_per_block=TINY_VAE.layers_per_block,
                            latent_channels=TINY_VAE.latent_channels).eval()
        vae.load_state_dict(vsd)
        img = torch.rand(1, 3, 64, 128, generator=g) * 2 - 1
        ref = torch.from_numpy(pin["ref/tiny/vae/encode_moments"])
        got = vae.quant_conv(vae.encoder(img))
        assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())
        ref_dec = torch.from_numpy(pin["ref/tiny/vae/decode"])
        got = vae.decoder(vae.post_quant_conv(ref[:, :TINY_VAE.latent_channels]))
        assert float((got - ref_dec).abs().max()) <= tol * float(ref_dec.abs().max())


# ---- per-block arithmetic of the restatement against independent torch implementations (round-3 verdict #9) ----------
# diffusers cannot be imported here, so the TOPOLOGY of sd2_unet.py / sd2_vae.py stays unpinned (test above); what these
# tests remove from the unpinned set is the arithmetic inside a block: the oracle's hand-written attention against
# F.scaled_dot_product_attention and nn.MultiheadAttention loaded with the same weights in the diffusers key layout
# (to_q / to_k / to_v / to_out.0; /root/reference/marigold/marigold_depth_pipeline.py:35-42 imports them through
# diffusers.models.attention_processor), GEGLU against the erf formula written out, the sinusoidal timestep table against
# a float64 double loop, the asymmetric down-sampling pad and the nearest up-sampling against explicit index arithmetic.
def test_oracle_attention_equals_sdpa_and_multihead_attention():
    from oracle.sd2_unet import Attention
    torch.manual_seed(3)
    B, N, C, heads = 2, 37, 64, 4
    att = Attention(C, heads, C // heads).double()
    x = torch.randn(B, N, C, dtype=torch.float64)
    got = att(x)
    q, k, v = (m(x).view(B, N, heads, -1).transpose(1, 2) for m in (att.to_q, att.to_k, att.to_v))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C)
    ref = att.to_out[0](ref)
    assert torch.allclose(got, ref, atol=1e-6), float((got - ref).abs().max())
    # the same weights in torch's own multi-head attention module (packed in_proj = [q; k; v] rows, no q/k/v bias)
    mha = torch.nn.MultiheadAttention(C, heads, bias=True, batch_first=True).double()
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([att.to_q.weight, att.to_k.weight, att.to_v.weight]))
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(att.to_out[0].weight)
        mha.out_proj.bias.copy_(att.to_out[0].bias)
    ref2, _ = mha(x, x, x, need_weights=False)
    assert torch.allclose(got, ref2, atol=1e-6), float((got - ref2).abs().max())
    # cross-attention over a 2-token context of another width (the empty-text embedding: [bos, eos] x 1024)
    catt = Attention(C, heads, C // heads, cross_dim=48).double()
    ctx = torch.randn(B, 2, 48, dtype=torch.float64)
    got = catt(x, ctx)
    q = catt.to_q(x).view(B, N, heads, -1).transpose(1, 2)
    k, v = (m(ctx).view(B, 2, heads, -1).transpose(1, 2) for m in (catt.to_k, catt.to_v))
    ref = catt.to_out[0](torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))
    assert torch.allclose(got, ref, atol=1e-6)


def test_oracle_vae_attention_equals_sdpa_single_head():
    from oracle.sd2_vae import VaeAttention
    torch.manual_seed(4)
    C = 64
    va = VaeAttention(C).double()
    x = torch.randn(2, C, 5, 7, dtype=torch.float64)
    got = va(x)
    h = torch.nn.functional.group_norm(x, 32, va.group_norm.weight, va.group_norm.bias, eps=1e-6)
    t = h.flatten(2).transpose(1, 2)                       # [B, HW, C]: one head of width C, scale C^-1/2
    q, k, v = (m(t)[:, None] for m in (va.to_q, va.to_k, va.to_v))
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)[:, 0]
    ref = va.to_out[0](o).transpose(1, 2).reshape(x.shape) + x
    assert torch.allclose(got, ref, atol=1e-6), float((got - ref).abs().max())


def test_oracle_geglu_layernorm_groupnorm_blocks():
    from oracle.sd2_unet import GEGLU, ResnetBlock2D
    torch.manual_seed(5)
    g = GEGLU(16, 24).double()
    x = torch.randn(3, 11, 16, dtype=torch.float64)
    y = x @ g.proj.weight.t() + g.proj.bias
    u, gate = y[..., :24], y[..., 24:]                      # diffusers: hidden_states, gate = proj(x).chunk(2, dim=-1)
    ref = u * (0.5 * gate * (1.0 + torch.erf(gate / math.sqrt(2.0))))
    assert torch.allclose(g(x), ref, atol=1e-12)
    # ResnetBlock2D: GroupNorm written out (per-group mean / biased variance over channels-in-group x pixels)
    rb = ResnetBlock2D(32, 64, 24, 1e-5).double()
    xx = torch.randn(2, 32, 6, 5, dtype=torch.float64)
    temb = torch.randn(2, 24, dtype=torch.float64)

    def gn(t, w, b, eps):
        B, C, H, W = t.shape
        tg = t.reshape(B, 32, C // 32 * H * W)
        tn = (tg - tg.mean(-1, keepdim=True)) / torch.sqrt(tg.var(-1, unbiased=False, keepdim=True) + eps)
        return tn.reshape(B, C, H, W) * w[None, :, None, None] + b[None, :, None, None]

    def silu(t):
        return t / (1.0 + torch.exp(-t))
    h = torch.nn.functional.conv2d(silu(gn(xx, rb.norm1.weight, rb.norm1.bias, 1e-5)), rb.conv1.weight, rb.conv1.bias, padding=1)
    h = h + (silu(temb) @ rb.time_emb_proj.weight.t() + rb.time_emb_proj.bias)[:, :, None, None]
    h = torch.nn.functional.conv2d(silu(gn(h, rb.norm2.weight, rb.norm2.bias, 1e-5)), rb.conv2.weight, rb.conv2.bias, padding=1)
    ref = torch.nn.functional.conv2d(xx, rb.conv_shortcut.weight, rb.conv_shortcut.bias) + h
    assert torch.allclose(rb(xx, temb), ref, atol=1e-10)


def test_oracle_timestep_table_and_resampling_index_arithmetic():
    from oracle.sd2_unet import Downsample2D, Upsample2D, sinusoidal_timestep_embedding
    t = torch.tensor([999, 500, 1])
    got = sinusoidal_timestep_embedding(t, 320).double()
    ref = torch.empty(3, 320, dtype=torch.float64)
    for i, tv in enumerate((999.0, 500.0, 1.0)):           # Timesteps(320, flip_sin_to_cos=True, downscale_freq_shift=0)
        for j in range(160):
            f = math.exp(-math.log(10000.0) * j / 160.0)
            ref[i, j], ref[i, 160 + j] = math.cos(tv * f), math.sin(tv * f)
    assert torch.allclose(got, ref, atol=2e-4)             # fp32 table, arguments up to 999 rad
    torch.manual_seed(6)
    x = torch.randn(1, 4, 7, 9, dtype=torch.float64)
    up = Upsample2D(4).double()
    big = torch.empty(1, 4, 14, 18, dtype=torch.float64)
    for y in range(14):
        for xx in range(18):
            big[:, :, y, xx] = x[:, :, y // 2, xx // 2]    # nearest, exact factor 2
    assert torch.allclose(up(x), torch.nn.functional.conv2d(big, up.conv.weight, up.conv.bias, padding=1), atol=1e-12)
    odd = torch.empty(1, 4, 13, 17, dtype=torch.float64)   # forward_upsample_size: nearest to an explicit size
    for y in range(13):
        for xx in range(17):
            odd[:, :, y, xx] = x[:, :, min(int(y * 7 / 13), 6), min(int(xx * 9 / 17), 8)]
    assert torch.allclose(up(x, (13, 17)), torch.nn.functional.conv2d(odd, up.conv.weight, up.conv.bias, padding=1), atol=1e-12)
    dn = Downsample2D(4, padding=0).double()               # VAE encoder: zero pad right / bottom by one, stride 2, no padding
    xp = torch.zeros(1, 4, 8, 10, dtype=torch.float64)
    xp[:, :, :7, :9] = x
    assert torch.allclose(dn(x), torch.nn.functional.conv2d(xp, dn.conv.weight, dn.conv.bias, stride=2), atol=1e-12)
