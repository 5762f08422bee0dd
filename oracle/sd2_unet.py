"""Oracle: SD-v2 ``UNet2DConditionModel`` as Marigold configures it.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **parity unpinned**: diffusers is
an un-vendored dependency of the reference (requirements.txt: diffusers>=0.25.0)
and cannot be imported here; this file restates its published architecture.

Reference call site: /root/reference/marigold/marigold_depth_pipeline.py:461-463
(``self.unet(unet_input, t, encoder_hidden_states=...).sample``); the 8-channel
``conv_in`` comes from /root/reference/src/trainer/marigold_depth_trainer.py:187-206.
Module / parameter names equal the diffusers state-dict keys
so a real ``unet/diffusion_pytorch_model.safetensors`` loads with
``load_state_dict``.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    """GN->SiLU->conv3x3->(+temb)->GN->SiLU->conv3x3, + (1x1) shortcut."""

    def __init__(self, cin, cout, temb_channels, eps, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        if temb_channels:
            self.time_emb_proj = nn.Linear(temb_channels, cout)
        else:
            self.time_emb_proj = None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers ``Attention``: q/k/v Linear (bias optional), to_out.0 Linear (bias)."""

    def __init__(self, query_dim, heads, dim_head, cross_dim=None, qkv_bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        cross_dim = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=qkv_bias)
        self.to_k = nn.Linear(cross_dim, inner, bias=qkv_bias)
        self.to_v = nn.Linear(cross_dim, inner, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def forward(self, x, context=None):
        ctx = x if context is None else context
        B, N, _ = x.shape
        h = self.heads
        q = self.to_q(x).view(B, N, h, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        p = torch.softmax(s.float(), dim=-1).to(v.dtype)
        o = torch.matmul(p, v).transpose(1, 2).reshape(B, N, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        u, g = self.proj(x).chunk(2, dim=-1)
        return u * F.gelu(g)  # erf gelu


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    """``use_linear_projection=True`` variant: GN(eps 1e-6) -> Linear -> block -> Linear -> +res."""

    def __init__(self, channels, heads, cross_dim, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, channels // heads, cross_dim)]
        )
        self.proj_out = nn.Linear(channels, channels)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        h = self.proj_out(h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, ch, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:  # VAE encoder: asymmetric right/bottom zero pad
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:  # diffusers ``forward_upsample_size`` path (latent dims not divisible by 8)
            x = F.interpolate(x, size=tuple(output_size), mode="nearest")
        return self.conv(x)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, heads, cross_dim, n_layers, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, 1e-5) for i in range(n_layers)]
        )
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, cross_dim) for _ in range(n_layers)]
            )
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, 1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, cross_dim)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin_skip, prev, cout, temb, heads, cross_dim, n_layers, has_attn, add_up):
        super().__init__()
        rs = []
        for j in range(n_layers):
            skip = cin_skip if j == n_layers - 1 else cout
            rin = prev if j == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, 1e-5))
        self.resnets = nn.ModuleList(rs)
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, cross_dim) for _ in range(n_layers)]
            )
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx, force_size=False):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            size = skips[-1].shape[2:] if force_size else None
            x = self.upsamplers[0](x, size)
        return x


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


def sinusoidal_timestep_embedding(t, dim=320):
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos, sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freqs[None].to(t.device)
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


SD2_UNET_CONFIG = dict(
    in_channels=8,
    out_channels=4,
    block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20),  # diffusers quirk: these are head COUNTS
    cross_attention_dim=1024,
)


class UNet2DConditionModel(nn.Module):
    def __init__(self, in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024):
        super().__init__()
        boc = list(block_out_channels)
        heads = list(attention_head_dim)
        temb = boc[0] * 4
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(boc),
            layers_per_block=layers_per_block, attention_head_dim=tuple(heads),
            cross_attention_dim=cross_attention_dim)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        n = len(boc)
        downs = []
        ch = boc[0]
        for i in range(n):
            downs.append(DownBlock(ch, boc[i], temb, heads[i], cross_attention_dim, layers_per_block,
                                   has_attn=(i < n - 1), add_down=(i < n - 1)))
            ch = boc[i]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb, heads[-1], cross_attention_dim)
        rev = boc[::-1]
        rheads = heads[::-1]
        ups = []
        out_ch = rev[0]
        for i in range(n):
            prev = out_ch
            out_ch = rev[i]
            in_ch = rev[min(i + 1, n - 1)]
            ups.append(UpBlock(in_ch, prev, out_ch, temb, rheads[i], cross_attention_dim,
                               layers_per_block + 1, has_attn=(i > 0), add_up=(i < n - 1)))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], device=sample.device)
        t = t.reshape(-1).expand(sample.shape[0])
        temb = sinusoidal_timestep_embedding(t, self.config.block_out_channels[0]).to(sample.dtype)
        temb = self.time_embedding(temb)
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb, encoder_hidden_states)
        n_up = len(self.up_blocks) - 1
        force = any(d % (2 ** n_up) != 0 for d in sample.shape[-2:])
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states, force)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return SimpleNamespace(sample=x)
