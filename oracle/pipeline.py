"""Oracle: the reference's per-batch prediction (``single_infer``) on the oracle modules.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows
/root/reference/marigold/marigold_depth_pipeline.py:396-516 (depth: encode :479-496, noise
:430-435, loop :455-468, decode :498-516, clip/shift :473-475) and
/root/reference/marigold/marigold_normals_pipeline.py:361-479 (normals: clip + L2 normalise
:437-440, 3-channel decode :463-479).  For parity runs the initial latents (and LCM per-step
noise, consumed through ``generator``) are passed in explicitly.
"""
import torch

from . import ensemble as ens

LATENT_SCALE = 0.18215  # marigold_depth_pipeline.py:118


@torch.no_grad()
def encode_rgb(vae, rgb_norm):
    h = vae.encoder(rgb_norm)
    mean, _logvar = torch.chunk(vae.quant_conv(h), 2, dim=1)
    return mean * LATENT_SCALE


@torch.no_grad()
def decode_latent(vae, latent):
    return vae.decoder(vae.post_quant_conv(latent / LATENT_SCALE))


@torch.no_grad()
def denoise(unet, scheduler, rgb_latent, target_latent, text_embed, n_steps, generator=None,
            trace=None):
    scheduler.set_timesteps(n_steps)
    ctx = text_embed.repeat(rgb_latent.shape[0], 1, 1).to(rgb_latent.dtype)
    for t in scheduler.timesteps:
        x = torch.cat([rgb_latent, target_latent], dim=1)  # order matters (:456-458)
        pred = unet(x, t, encoder_hidden_states=ctx).sample
        target_latent = scheduler.step(pred, t, target_latent, generator=generator).prev_sample
        if trace is not None:
            trace.append((int(t), pred.clone(), target_latent.clone()))
    return target_latent


@torch.no_grad()
def single_infer(kind, unet, vae, scheduler, rgb_norm, init_latent, text_embed, n_steps,
                 generator=None, trace=None):
    """rgb_norm [B,3,H,W] in [-1,1]; init_latent [B,4,H/8,W/8] ([B,4n,...] for 'iid').
    kind: 'depth' | 'normals' | 'iid'."""
    rgb_latent = encode_rgb(vae, rgb_norm)
    lat = denoise(unet, scheduler, rgb_latent, init_latent.to(rgb_latent.dtype), text_embed,
                  n_steps, generator, trace)
    if kind == "iid":   # marigold_iid_pipeline.py:556-585, 523-526: every modality decoded separately
        n = lat.shape[1] // 4
        dec = torch.cat([decode_latent(vae, lat[:, 4 * i:4 * i + 4]) for i in range(n)], dim=1)
        return (torch.clip(dec, -1.0, 1.0) + 1.0) / 2.0
    dec = decode_latent(vae, lat)
    if kind == "depth":
        d = dec.mean(dim=1, keepdim=True)
        return (torch.clip(d, -1.0, 1.0) + 1.0) / 2.0
    n = torch.clip(dec, -1.0, 1.0)
    return n / torch.norm(n, dim=1, keepdim=True).clamp(min=1e-6)


@torch.no_grad()
def predict(kind, unet, vae, scheduler, image_u8, init_latents, text_embed, n_steps,
            ensemble_kwargs=None, generator=None):
    """Whole-call oracle at processing_res == input res (no resize): the E members of
    ``init_latents`` [E,4,h,w] are predicted one by one and aggregated like
    marigold_depth_pipeline.py:258-303 / marigold_normals_pipeline.py:272-280."""
    rgb = image_u8.float() / 255.0 * 2.0 - 1.0
    preds = [single_infer(kind, unet, vae, scheduler, rgb, init_latents[e:e + 1], text_embed,
                          n_steps, generator) for e in range(init_latents.shape[0])]
    preds = torch.cat(preds, dim=0)
    if preds.shape[0] == 1:
        return preds, None, preds
    kw = ensemble_kwargs or {}
    if kind == "depth":
        out, unc = ens.ensemble_depth(preds, True, True, **kw)
    elif kind == "iid":
        out, unc = ens.ensemble_iid(preds, **kw)
    else:
        out, unc = ens.ensemble_normals(preds, **kw)
    return out, unc, preds
