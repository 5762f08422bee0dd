"""Checkpoint IO in the diffusers folder layout the reference loads with ``from_pretrained``
(script/depth/run.py:203-215): ``model_index.json``, ``unet/config.json`` +
``diffusion_pytorch_model[.fp16].safetensors``, ``vae/...``, ``scheduler/scheduler_config.json``,
optional ``text_encoder/`` + ``tokenizer/`` (or a precomputed ``empty_text_embed.safetensors``).
"""
import json
import os

import torch
from safetensors.torch import load_file, save_file

from . import config_check as CC
from .modules import AutoencoderKLHIP, UNet2DConditionModelHIP
from .schedulers import DDIMScheduler, LCMScheduler

_SCHEDULERS = {"DDIMScheduler": DDIMScheduler, "LCMScheduler": LCMScheduler}


def _weights(folder, variant):
    names = ([f"diffusion_pytorch_model.{variant}.safetensors"] if variant else []) + \
        ["diffusion_pytorch_model.safetensors"]
    for n in names:
        p = os.path.join(folder, n)
        if os.path.exists(p):
            return load_file(p)
    raise FileNotFoundError(f"no safetensors weights in {folder} (tried {names})")


def _json(path):
    with open(path) as f:
        return json.load(f)


def _rename_legacy_vae_keys(sd):
    """Older AutoencoderKL files use query/key/value/proj_attn."""
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts:
            parts = [ren.get(p, p) for p in parts]
            if v.dim() == 4 and parts[-1] == "weight" and parts[-2] in ("to_q", "to_k", "to_v", "0"):
                v = v.reshape(v.shape[0], v.shape[1])
        out[".".join(parts)] = v
    return out


def _resolve(path):
    """A local checkpoint folder, or a hub id already present in the local Hugging Face cache (the
    reference passes ids like ``prs-eth/marigold-depth-v1-1``; nothing is downloaded here)."""
    if os.path.isdir(path):
        return path
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(path, local_files_only=True)
    except Exception as e:  # noqa: BLE001
        raise FileNotFoundError(
            f"checkpoint '{path}' is neither a local folder nor in the local Hugging Face cache ({type(e).__name__}); "
            f"download it first (script/download_weights.sh in the reference) and pass the folder") from e


# scheduler_config.json fields that do not enter the arithmetic of this path, with the only value accepted
_SCHED_INERT = {"trained_betas": (None,), "thresholding": (False,), "clip_sample_range": None, "dynamic_thresholding_ratio": None,
                "sample_max_value": None, "skip_prk_steps": None}


def _scheduler_kwargs(sname, scfg):
    allowed = set(_SCHEDULERS[sname]._defaults)
    kw = {}
    for k, v in scfg.items():
        if k.startswith("_"):
            continue
        if k in allowed:
            kw[k] = v
        elif k in _SCHED_INERT:
            ok = _SCHED_INERT[k]
            if ok is not None and v not in ok:
                raise CC.UnsupportedConfigError(f"scheduler/scheduler_config.json: {k} = {v!r} is not implemented")
        else:
            raise CC.UnsupportedConfigError(f"scheduler/scheduler_config.json: unknown field '{k}' = {v!r}")
    return kw


def load_pipeline(cls, path, variant=None, torch_dtype=None, **kw):
    path = _resolve(path)
    index = _json(os.path.join(path, "model_index.json"))
    # every field of the two config files is checked against what the engine implements: a value it cannot honour
    # (or a field it does not know) raises instead of loading a model that would compute something else
    unet_cfg = CC.unet_config_from_json(_json(os.path.join(path, "unet", "config.json")))
    vae_cfg = CC.vae_config_from_json(_json(os.path.join(path, "vae", "config.json")))
    if torch_dtype is not None and torch_dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise ValueError(f"torch_dtype={torch_dtype}: expected torch.float32, torch.float16 or torch.bfloat16")
    # torch_dtype=torch.float16 (the reference's --fp16, script/depth/run.py:203-211) runs the engine's fp16-operand build: the same
    # kernels on fp16 operands with fp32 accumulation; torch.bfloat16 the product (bf16) build.  The reference's DEFAULT (fp32, or no
    # torch_dtype) has no engine counterpart - the matrix cores take 16-bit operands - and runs the bf16 build, said once, loudly.
    compute = torch.float16 if torch_dtype is torch.float16 else torch.bfloat16
    if torch_dtype in (None, torch.float32):
        import logging
        logging.warning("torch_dtype=float32 (or none): the HIP engine has no fp32-operand mode; it computes on bf16 operands with fp32 "
                        "accumulation and keeps latents / predictions in fp32 (torch_dtype=torch.float16 selects fp16 operands)")
    unet = UNet2DConditionModelHIP(_weights(os.path.join(path, "unet"), variant), unet_cfg, compute_dtype=compute)
    vae = AutoencoderKLHIP(_rename_legacy_vae_keys(_weights(os.path.join(path, "vae"), variant)), vae_cfg, compute_dtype=compute)
    scfg = _json(os.path.join(path, "scheduler", "scheduler_config.json"))
    sname = scfg.get("_class_name", "DDIMScheduler")
    if sname not in _SCHEDULERS:
        raise RuntimeError(f"Unsupported scheduler type: {sname}")
    scheduler = _SCHEDULERS[sname](**_scheduler_kwargs(sname, scfg))
    text_encoder = tokenizer = empty = None
    emb = os.path.join(path, "empty_text_embed.safetensors")
    if os.path.exists(emb):
        empty = load_file(emb)["empty_text_embed"]
    elif os.path.isdir(os.path.join(path, "text_encoder")):
        from transformers import CLIPTextModel, CLIPTokenizer
        text_encoder = CLIPTextModel.from_pretrained(os.path.join(path, "text_encoder"))
        tokenizer = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
    extra = {k: index[k] for k in ("scale_invariant", "shift_invariant", "default_denoising_steps",
                                   "default_processing_resolution", "target_properties") if k in index}
    if cls.__name__ != "MarigoldDepthPipeline":
        extra.pop("scale_invariant", None)
        extra.pop("shift_invariant", None)
    if cls.__name__ != "MarigoldIIDPipeline":
        extra.pop("target_properties", None)
    pipe = cls(unet=unet, vae=vae, scheduler=scheduler, text_encoder=text_encoder, tokenizer=tokenizer,
               empty_text_embed=empty, **extra)
    # the reference draws its latents / LCM noise in the dtype the pipeline was loaded with (marigold_depth_pipeline.py:430-435)
    if torch_dtype in (torch.float16, torch.bfloat16):
        pipe.noise_dtype = torch_dtype
    return pipe


def save_synthetic_checkpoint(path, cls_name, unet_sd, vae_sd, unet_cfg, vae_cfg, scheduler, empty_text_embed,
                              **model_index):
    """Write a checkpoint folder in the layout above (used by tests; weights as safetensors)."""
    for sub in ("unet", "vae", "scheduler"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    save_file({k: v.contiguous() for k, v in unet_sd.items()},
              os.path.join(path, "unet", "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in vae_sd.items()},
              os.path.join(path, "vae", "diffusion_pytorch_model.safetensors"))
    # the full published key sets (config_check.SD2_*), with the fields this architecture instance changes
    with open(os.path.join(path, "unet", "config.json"), "w") as f:
        json.dump(dict(CC.SD2_UNET_CONFIG, in_channels=unet_cfg.in_channels,
                       out_channels=unet_cfg.out_channels, block_out_channels=list(unet_cfg.block_out_channels),
                       layers_per_block=unet_cfg.layers_per_block, attention_head_dim=list(unet_cfg.heads),
                       cross_attention_dim=unet_cfg.cross_attention_dim, norm_num_groups=unet_cfg.norm_groups), f)
    with open(os.path.join(path, "vae", "config.json"), "w") as f:
        json.dump(dict(CC.SD2_VAE_CONFIG, block_out_channels=list(vae_cfg.block_out_channels),
                       layers_per_block=vae_cfg.layers_per_block, latent_channels=vae_cfg.latent_channels,
                       norm_num_groups=vae_cfg.norm_groups), f)
    with open(os.path.join(path, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name=type(scheduler).__name__, **vars(scheduler.config)), f)
    save_file({"empty_text_embed": empty_text_embed.contiguous()}, os.path.join(path, "empty_text_embed.safetensors"))
    with open(os.path.join(path, "model_index.json"), "w") as f:
        json.dump(dict(_class_name=cls_name, **model_index), f)
