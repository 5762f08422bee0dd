"""__graft_entry__.smoke(): one tiny depth prediction on the GPU through the public pipeline,
checked against the CPU oracle (the oracle is only the checker here)."""
import numpy as np
import torch


def run(device="cuda:0"):
    import marigold_amd as M
    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(min(16, usable_cores()))
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE
    from oracle import metrics as omet, pipeline as opipe
    from oracle.schedulers import DDIMScheduler as OracleDDIM
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL

    pipe = M.build_synthetic_pipeline("depth", TINY_UNET, TINY_VAE, default_processing_resolution=0).to(device)
    img = syn.synthetic_image(64, 128, seed=0)
    lat0 = syn.synthetic_latents(2, 8, 16, seed=2024)
    out = pipe(img, denoising_steps=2, ensemble_size=2, processing_res=0, match_input_res=False,
               color_map=None, show_progress_bar=False, init_latents=lat0)
    assert out.depth_np.shape == (64, 128) and np.isfinite(out.depth_np).all()

    unet = UNet2DConditionModel(block_out_channels=TINY_UNET.block_out_channels,
                                attention_head_dim=TINY_UNET.heads,
                                cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict(TINY_UNET))
    vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels).eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict(TINY_VAE))
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    ref, _, members = opipe.predict("depth", unet, vae, OracleDDIM(), img, lat0, ctx, 2)
    # single member (no ensembling): tight, in the reference's affine-invariant metric
    out1 = pipe(img, denoising_steps=2, ensemble_size=1, processing_res=0, match_input_res=False,
                color_map=None, show_progress_bar=False, init_latents=lat0[:1])
    e1 = omet.affine_invariant_depth_errors(members[0, 0].numpy(), out1.depth_np)
    print(f"[smoke] depth 64x128 E=1 T=2 vs CPU oracle: {e1}")
    assert e1["rmse"] < 8.5e-3 and e1["delta1"] > 0.99, e1   # measured 2.76e-3 / 0.9988 (GPUTEST_r04): 3x
    # E=2 ensemble: the reference's shift optimiser stops where its fp32 noise does (ensemble.py
    # docstring), so the ensembled maps agree loosely on these random-weight members
    err = omet.affine_invariant_depth_errors(ref.squeeze().numpy(), out.depth_np)
    print(f"[smoke] depth 64x128 E=2 T=2 vs CPU oracle: {err}")
    assert err["rmse"] < 1.7e-2 and err["delta1"] > 0.98 and abs(err["scale"] - 1.0) < 0.02, err   # measured 5.7e-3 / 0.9973 / 0.9989: 3x
    return err
