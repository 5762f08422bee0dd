"""Oracle: SD-v2 ``AutoencoderKL`` encoder / decoder.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **parity unpinned** (diffusers is an
un-vendored dependency; restated from its published architecture).

Reference call sites: /root/reference/marigold/marigold_depth_pipeline.py:491-492
(``vae.encoder`` + ``vae.quant_conv``) and :512-513 (``vae.post_quant_conv`` +
``vae.decoder``).  Names equal the diffusers state-dict keys.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sd2_unet import Attention, Downsample2D, ResnetBlock2D, Upsample2D


class VaeAttention(nn.Module):
    """Mid-block attention: GN -> 1-head attention (q/k/v with bias) -> +residual."""

    def __init__(self, ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, ch, eps=1e-6)
        attn = Attention(ch, 1, ch, qkv_bias=True)
        self.to_q, self.to_k, self.to_v, self.to_out = attn.to_q, attn.to_k, attn.to_v, attn.to_out
        self._attn = [attn]  # not registered twice: keys stay to_q/to_k/to_v/to_out.0

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        h = self._attn[0](h)
        h = h.transpose(1, 2).reshape(B, C, H, W)
        return h + x


class VaeMidBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, 0, 1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(ch)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n_layers, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, 0, 1e-6) for i in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n_layers, add_up):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, 0, 1e-6) for i in range(n_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, in_ch=3, latent=4, boc=(128, 256, 512, 512), layers=2):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, c in enumerate(boc):
            blocks.append(DownEncoderBlock2D(ch, c, layers, add_down=(i < len(boc) - 1)))
            ch = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = VaeMidBlock(ch)
        self.conv_norm_out = nn.GroupNorm(32, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, out_ch=3, latent=4, boc=(128, 256, 512, 512), layers=2):
        super().__init__()
        rev = list(boc)[::-1]
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0])
        blocks, ch = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(UpDecoderBlock2D(ch, c, layers + 1, add_up=(i < len(rev) - 1)))
            ch = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, out_ch, 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block,
                                      latent_channels=latent_channels)
        self.encoder = Encoder(3, latent_channels, block_out_channels, layers_per_block)
        self.decoder = Decoder(3, latent_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype
