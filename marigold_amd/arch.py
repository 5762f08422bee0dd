"""Architecture tables for the checkpoints Marigold ships: SD-v2 UNet (8-ch conv_in) and
SD AutoencoderKL, as (state-dict key -> shape) enumerations in the diffusers key scheme.

The reference never spells these out - it imports them from diffusers
(/root/reference/marigold/marigold_depth_pipeline.py:35-41) and loads
``unet/diffusion_pytorch_model.safetensors`` / ``vae/...`` through ``from_pretrained``
(/root/reference/script/depth/run.py:213-215).  The engine's program builders
(``unet_program.py`` / ``vae_program.py``) walk the same structure; ``synthetic.py`` fills it
with seeded weights when no checkpoint is available.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: Tuple[int, ...] = (5, 10, 20, 20)   # diffusers' "attention_head_dim" (head counts)
    cross_attention_dim: int = 1024
    norm_groups: int = 32

    @property
    def temb_dim(self):
        return self.block_out_channels[0] * 4


@dataclass(frozen=True)
class VAEConfig:
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    norm_groups: int = 32


def _conv(d, name, cout, cin, k):
    d[f"{name}.weight"] = (cout, cin, k, k)
    d[f"{name}.bias"] = (cout,)


def _lin(d, name, cout, cin, bias=True):
    d[f"{name}.weight"] = (cout, cin)
    if bias:
        d[f"{name}.bias"] = (cout,)


def _norm(d, name, c):
    d[f"{name}.weight"] = (c,)
    d[f"{name}.bias"] = (c,)


def _resnet(d, name, cin, cout, temb):
    _norm(d, f"{name}.norm1", cin)
    _conv(d, f"{name}.conv1", cout, cin, 3)
    if temb:
        _lin(d, f"{name}.time_emb_proj", cout, temb)
    _norm(d, f"{name}.norm2", cout)
    _conv(d, f"{name}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, f"{name}.conv_shortcut", cout, cin, 1)


def _transformer(d, name, c, cross):
    _norm(d, f"{name}.norm", c)
    _lin(d, f"{name}.proj_in", c, c)
    b = f"{name}.transformer_blocks.0"
    _norm(d, f"{b}.norm1", c)
    for q in ("to_q", "to_k", "to_v"):
        _lin(d, f"{b}.attn1.{q}", c, c, bias=False)
    _lin(d, f"{b}.attn1.to_out.0", c, c)
    _norm(d, f"{b}.norm2", c)
    _lin(d, f"{b}.attn2.to_q", c, c, bias=False)
    _lin(d, f"{b}.attn2.to_k", c, cross, bias=False)
    _lin(d, f"{b}.attn2.to_v", c, cross, bias=False)
    _lin(d, f"{b}.attn2.to_out.0", c, c)
    _norm(d, f"{b}.norm3", c)
    _lin(d, f"{b}.ff.net.0.proj", 8 * c, c)
    _lin(d, f"{b}.ff.net.2", c, 4 * c)
    _lin(d, f"{name}.proj_out", c, c)


def unet_up_resnet_channels(cfg: UNetConfig):
    """[(block, j, in_from_below, skip, out)] for the up path (SURVEY.md App. B skip stack)."""
    boc = list(cfg.block_out_channels)
    rev = boc[::-1]
    n = len(boc)
    rows = []
    out_ch = rev[0]
    for i in range(n):
        prev, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_ch if j == cfg.layers_per_block else out_ch
            rin = prev if j == 0 else out_ch
            rows.append((i, j, rin, skip, out_ch))
    return rows


def unet_param_shapes(cfg: UNetConfig = UNetConfig()):
    d = OrderedDict()
    boc = list(cfg.block_out_channels)
    n = len(boc)
    temb = cfg.temb_dim
    _conv(d, "conv_in", boc[0], cfg.in_channels, 3)
    _lin(d, "time_embedding.linear_1", temb, boc[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            _resnet(d, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i], temb)
            if i < n - 1:
                _transformer(d, f"down_blocks.{i}.attentions.{j}", boc[i], cfg.cross_attention_dim)
        if i < n - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", boc[i], boc[i], 3)
        ch = boc[i]
    _resnet(d, "mid_block.resnets.0", ch, ch, temb)
    _transformer(d, "mid_block.attentions.0", ch, cfg.cross_attention_dim)
    _resnet(d, "mid_block.resnets.1", ch, ch, temb)
    for (i, j, rin, skip, out) in unet_up_resnet_channels(cfg):
        _resnet(d, f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
        if i > 0:
            _transformer(d, f"up_blocks.{i}.attentions.{j}", out, cfg.cross_attention_dim)
        if j == cfg.layers_per_block and i < n - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(d, "conv_norm_out", boc[0])
    _conv(d, "conv_out", cfg.out_channels, boc[0], 3)
    return d


def _vae_mid(d, name, c):
    _resnet(d, f"{name}.resnets.0", c, c, 0)
    a = f"{name}.attentions.0"
    _norm(d, f"{a}.group_norm", c)
    for q in ("to_q", "to_k", "to_v"):
        _lin(d, f"{a}.{q}", c, c)
    _lin(d, f"{a}.to_out.0", c, c)
    _resnet(d, f"{name}.resnets.1", c, c, 0)


def vae_param_shapes(cfg: VAEConfig = VAEConfig()):
    d = OrderedDict()
    boc = list(cfg.block_out_channels)
    n = len(boc)
    L = cfg.latent_channels
    _conv(d, "encoder.conv_in", boc[0], 3, 3)
    ch = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else c, c, 0)
        if i < n - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        ch = c
    _vae_mid(d, "encoder.mid_block", ch)
    _norm(d, "encoder.conv_norm_out", ch)
    _conv(d, "encoder.conv_out", 2 * L, ch, 3)
    rev = boc[::-1]
    _conv(d, "decoder.conv_in", rev[0], L, 3)
    _vae_mid(d, "decoder.mid_block", rev[0])
    ch = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet(d, f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else c, c, 0)
        if i < n - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        ch = c
    _norm(d, "decoder.conv_norm_out", ch)
    _conv(d, "decoder.conv_out", 3, ch, 3)
    _conv(d, "quant_conv", 2 * L, 2 * L, 1)
    _conv(d, "post_quant_conv", L, L, 1)
    return d


# Small configurations used by CPU tests / smoke (same topology, fewer channels).
TINY_UNET = UNetConfig(block_out_channels=(64, 128, 128, 128), heads=(1, 2, 2, 2),
                       cross_attention_dim=64)
TINY_VAE = VAEConfig(block_out_channels=(64, 64, 128, 128))
