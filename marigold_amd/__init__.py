"""marigold_amd - MI355X-native engine for the Marigold inference hot path, behind the
reference's pipeline API (exports mirror /root/reference/marigold/__init__.py:31-41)."""
from .pipeline import (IIDEntry, MarigoldDepthOutput, MarigoldDepthPipeline,  # noqa: F401
                       MarigoldIIDOutput, MarigoldIIDPipeline, MarigoldNormalsOutput, MarigoldNormalsPipeline)

MarigoldPipeline = MarigoldDepthPipeline  # for backward compatibility


def build_synthetic_pipeline(kind="depth", unet_cfg=None, vae_cfg=None, scheduler=None, seed=1234, compute_dtype=None, **kw):
    """A pipeline on seeded synthetic weights in the real architecture (no checkpoints exist in
    this environment - BASELINE.md §4)."""
    from . import synthetic as syn
    from .arch import UNetConfig, VAEConfig
    from .modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from .schedulers import DDIMScheduler
    if kind == "iid":   # appearance model: albedo + material
        kw.setdefault("target_properties", {"target_names": ["albedo", "material"],
                                            "albedo": {"prediction_space": "srgb"},
                                            "material": {"prediction_space": "stack"}})
        n = len(kw["target_properties"]["target_names"])
        unet_cfg = unet_cfg or UNetConfig(in_channels=4 + 4 * n, out_channels=4 * n)
    unet_cfg = unet_cfg or UNetConfig()
    vae_cfg = vae_cfg or VAEConfig()
    import torch
    compute_dtype = compute_dtype or torch.bfloat16   # torch.float16: the fp16-operand build of the engine
    unet = UNet2DConditionModelHIP(syn.synthetic_unet_state_dict(unet_cfg, seed), unet_cfg, compute_dtype=compute_dtype)
    vae = AutoencoderKLHIP(syn.synthetic_vae_state_dict(vae_cfg, seed), vae_cfg, compute_dtype=compute_dtype)
    emb = syn.synthetic_text_embedding(unet_cfg.cross_attention_dim)
    scheduler = scheduler or DDIMScheduler()
    cls = {"depth": MarigoldDepthPipeline, "normals": MarigoldNormalsPipeline, "iid": MarigoldIIDPipeline}[kind]
    kw.setdefault("default_denoising_steps", 4)
    kw.setdefault("default_processing_resolution", 768)
    return cls(unet=unet, vae=vae, scheduler=scheduler, empty_text_embed=emb, **kw)
