"""Strict validation of the diffusers ``config.json`` files of a checkpoint folder.

The reference hands these files to diffusers (``MarigoldDepthPipeline.from_pretrained``,
/root/reference/script/depth/run.py:203-222), which builds whatever architecture they describe.  The HIP engine
implements ONE architecture family (SD-v2 UNet2DConditionModel with cross-attention transformer blocks, SD
AutoencoderKL); a field it cannot honour must stop the load, never be ignored, or a checkpoint would load and produce
wrong numbers.  Every key of the published configs is listed here with the values the engine implements
(SURVEY.md App. C.1 / C.5 / C.6); unknown keys are errors too.
"""
from .arch import UNetConfig, VAEConfig

_ANY = object()          # informational: does not change the arithmetic of the inference path
_INFO_KEYS = ("_class_name", "_diffusers_version", "_name_or_path", "_use_default_values")


class UnsupportedConfigError(ValueError):
    pass


def _none_or(*vals):
    return (None,) + vals


# key -> allowed values (tuple), _ANY, or a callable(value) -> bool
_UNET_RULES = {
    "act_fn": ("silu", "swish"),
    "addition_embed_type": (None,), "addition_embed_type_num_heads": _ANY, "addition_time_embed_dim": (None,),
    "attention_head_dim": lambda v: isinstance(v, (list, tuple)) and all(isinstance(h, int) and h > 0 for h in v),
    "attention_type": ("default",),
    "block_out_channels": lambda v: isinstance(v, (list, tuple)) and len(v) == 4 and all(c % 64 == 0 for c in v),
    "center_input_sample": (False,),
    "class_embed_type": (None,), "class_embeddings_concat": (False,), "num_class_embeds": (None,),
    "projection_class_embeddings_input_dim": (None,),
    "conv_in_kernel": (3,), "conv_out_kernel": (3,),
    "cross_attention_dim": lambda v: isinstance(v, int) and v > 0,
    "cross_attention_norm": (None,),
    "down_block_types": (["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],),
    "up_block_types": (["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],),
    "mid_block_type": ("UNetMidBlock2DCrossAttn",),
    "downsample_padding": (1,),
    "dropout": _ANY,                       # inactive at inference
    "dual_cross_attention": (False,),
    "encoder_hid_dim": (None,), "encoder_hid_dim_type": (None,),
    "flip_sin_to_cos": (True,), "freq_shift": (0,),
    "in_channels": lambda v: isinstance(v, int) and v in (8, 12, 16),      # 4 image + 4 per predicted modality
    "out_channels": lambda v: isinstance(v, int) and v in (4, 8, 12),
    "layers_per_block": (2,),
    "mid_block_only_cross_attention": (None,), "mid_block_scale_factor": (1, 1.0),
    "norm_eps": (1e-5,), "norm_num_groups": (32,),
    "num_attention_heads": (None,),
    "only_cross_attention": (False,),
    "resnet_out_scale_factor": (1, 1.0), "resnet_skip_time_act": (False,), "resnet_time_scale_shift": ("default",),
    "reverse_transformer_layers_per_block": (None,), "transformer_layers_per_block": (1,),
    "sample_size": _ANY,                   # default latent size only
    "time_cond_proj_dim": (None,), "time_embedding_act_fn": (None,), "time_embedding_dim": (None,),
    "time_embedding_type": ("positional",), "timestep_post_act": (None,),
    "upcast_attention": _ANY,              # the attention kernels always take the softmax in fp32
    "use_linear_projection": (True,),      # proj_in / proj_out are Linear layers on the token-major tensor
}

_VAE_RULES = {
    "act_fn": ("silu", "swish"),
    "block_out_channels": lambda v: isinstance(v, (list, tuple)) and len(v) == 4 and all(c % 64 == 0 for c in v),
    "down_block_types": (["DownEncoderBlock2D"] * 4,), "up_block_types": (["UpDecoderBlock2D"] * 4,),
    "force_upcast": _ANY,                  # activations are bf16 with fp32 accumulation either way
    "in_channels": (3,), "out_channels": (3,),
    "latent_channels": (4,), "layers_per_block": (2,), "norm_num_groups": (32,),
    "sample_size": _ANY,
    "scaling_factor": (0.18215,),          # the reference hard-codes it (marigold_depth_pipeline.py:118)
    "shift_factor": (None,), "latents_mean": (None,), "latents_std": (None,),
    "use_quant_conv": (True,), "use_post_quant_conv": (True,), "mid_block_add_attention": (True,),
}


def _check(cfg, rules, what):
    for k, v in cfg.items():
        if k in _INFO_KEYS:
            continue
        if k not in rules:
            raise UnsupportedConfigError(f"{what}/config.json: unknown field '{k}' = {v!r}; the HIP engine does not know "
                                         f"whether it can honour it")
        rule = rules[k]
        if rule is _ANY:
            continue
        ok = rule(v) if callable(rule) else any(v == a for a in rule)
        if not ok:
            allowed = "see marigold_amd/config_check.py" if callable(rule) else f"supported: {list(rule)}"
            raise UnsupportedConfigError(f"{what}/config.json: {k} = {v!r} is not implemented by the HIP engine ({allowed})")


def unet_config_from_json(cfg: dict) -> UNetConfig:
    _check(cfg, _UNET_RULES, "unet")
    boc = tuple(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    heads = tuple(cfg.get("attention_head_dim", (5, 10, 20, 20)))
    if len(heads) != len(boc) or any(c % h != 0 or c // h != 64 for c, h in zip(boc, heads)):
        raise UnsupportedConfigError(f"unet/config.json: attention_head_dim {list(heads)} with block_out_channels {list(boc)}: "
                                     f"the flash-attention kernel is built for head dimension 64 (channels / heads)")
    return UNetConfig(in_channels=cfg.get("in_channels", 8), out_channels=cfg.get("out_channels", 4),
                      block_out_channels=boc, layers_per_block=cfg.get("layers_per_block", 2), heads=heads,
                      cross_attention_dim=cfg.get("cross_attention_dim", 1024), norm_groups=cfg.get("norm_num_groups", 32))


def vae_config_from_json(cfg: dict) -> VAEConfig:
    _check(cfg, _VAE_RULES, "vae")
    return VAEConfig(block_out_channels=tuple(cfg.get("block_out_channels", (128, 256, 512, 512))),
                     layers_per_block=cfg.get("layers_per_block", 2), latent_channels=cfg.get("latent_channels", 4),
                     norm_groups=cfg.get("norm_num_groups", 32))


# The published key sets (diffusers 0.25, stabilityai/stable-diffusion-2 as shipped inside the Marigold checkpoints):
# used by save_synthetic_checkpoint so that test folders carry every field a real one does.
SD2_UNET_CONFIG = {
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.25.0", "act_fn": "silu", "addition_embed_type": None,
    "addition_embed_type_num_heads": 64, "addition_time_embed_dim": None, "attention_head_dim": [5, 10, 20, 20],
    "attention_type": "default", "block_out_channels": [320, 640, 1280, 1280], "center_input_sample": False,
    "class_embed_type": None, "class_embeddings_concat": False, "conv_in_kernel": 3, "conv_out_kernel": 3,
    "cross_attention_dim": 1024, "cross_attention_norm": None,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "dropout": 0.0, "dual_cross_attention": False, "encoder_hid_dim": None,
    "encoder_hid_dim_type": None, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 8, "layers_per_block": 2,
    "mid_block_only_cross_attention": None, "mid_block_scale_factor": 1, "mid_block_type": "UNetMidBlock2DCrossAttn",
    "norm_eps": 1e-05, "norm_num_groups": 32, "num_attention_heads": None, "num_class_embeds": None,
    "only_cross_attention": False, "out_channels": 4, "projection_class_embeddings_input_dim": None,
    "resnet_out_scale_factor": 1.0, "resnet_skip_time_act": False, "resnet_time_scale_shift": "default",
    "reverse_transformer_layers_per_block": None, "sample_size": 96, "time_cond_proj_dim": None,
    "time_embedding_act_fn": None, "time_embedding_dim": None, "time_embedding_type": "positional",
    "timestep_post_act": None, "transformer_layers_per_block": 1,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "upcast_attention": True, "use_linear_projection": True,
}
SD2_VAE_CONFIG = {
    "_class_name": "AutoencoderKL", "_diffusers_version": "0.25.0", "act_fn": "silu",
    "block_out_channels": [128, 256, 512, 512], "down_block_types": ["DownEncoderBlock2D"] * 4, "force_upcast": True,
    "in_channels": 3, "latent_channels": 4, "layers_per_block": 2, "norm_num_groups": 32, "out_channels": 3,
    "sample_size": 768, "scaling_factor": 0.18215, "up_block_types": ["UpDecoderBlock2D"] * 4,
}
