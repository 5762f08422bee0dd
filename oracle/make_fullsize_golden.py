"""Full-architecture goldens at the benchmark resolution (768x768) for BASELINE.json's C2 / C4 / C5.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Run in the BUILD container (about 25 minutes on 8 cores):

    python -m oracle.make_fullsize_golden

The fp32 CPU oracle (``oracle/sd2_unet.py``, ``sd2_vae.py``, ``schedulers.py``, ``pipeline.py`` - the restatement of
what /root/reference/marigold/marigold_depth_pipeline.py:396-516 and marigold_normals_pipeline.py:361-479 call into)
is run on the SD-v2 architecture with the seeded synthetic weights of ``marigold_amd/synthetic.py`` (every tensor is
drawn from its own seeded CPU generator, so the GPU box rebuilds bit-identical weights), and its outputs are stored
as ``tests/golden/fullsize_768.npz``.  The -m gpu tests (tests/test_gpu_fullsize.py) compare the engine against
them without running the oracle on the GPU box, which would take 20 minutes of its CPU.

Members and decodes are shared between the configurations - depth and normals differ only in the pointwise tail
after the decoder (depth :473-475,515; normals :437-440):

* ``rgb_latent``            encode_rgb of synthetic_image(768, 768, seed 0)                        [1,4,96,96]
* DDIM trailing / zero-SNR / v-prediction, T = 10 (C2, C5): init latents synthetic_latents(4, 96, 96, seed 7)
    ``ddim10_trace_m0``     latent of member 0 after every step                                    [10,4,96,96]
    ``ddim10_pred_m0``      UNet output of member 0 at every step                                  [10,4,96,96]
    ``ddim10_final``        x_0 latents of all 4 members                                           [4,4,96,96]
    ``ddim10_depth_m0``     C2: single_infer depth of member 0 (fp16)                              [768,768]
    ``ddim10_normals_m0``   single_infer normals of member 0 (fp16)                                [3,768,768]
    ``ddim10_normals_e4``   C5: ensemble_normals over the 4 members (fp16)                         [3,768,768]
* LCM, T = 4 (C4): member 0, per-step noise = torch.randn((1,4,96,96)) from a CPU generator seeded 99
    ``lcm4_trace``, ``lcm4_depth`` (fp16)
Stored fp16 maps carry <= 5e-4 relative rounding - two orders below the bf16 engine tolerance.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

IMG_SEED, LAT_SEED, LCM_NOISE_SEED, MEMBERS = 0, 7, 99, 4


def main_heavy():
    """C2 once more on weights with SD-like activation statistics planted (synthetic.plant_heavy_tails: outlier channels in
    the residual stream, large norm gains, GEGLU gates beyond +-4): member 0, DDIM T = 10, latent after every step and the
    depth map -> tests/golden/fullsize_768_heavy.npz (about 4 minutes on 8 cores)."""
    from marigold_amd import synthetic as syn
    from oracle import pipeline as opipe
    from oracle.schedulers import DDIMScheduler
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL

    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "8")))
    t0 = time.time()
    unet = UNet2DConditionModel().eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict(heavy_tail=True))
    vae = AutoencoderKL().eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict())
    ctx = syn.synthetic_text_embedding()
    rgb = syn.synthetic_image(768, 768, seed=IMG_SEED).float() / 255.0 * 2.0 - 1.0
    lat0 = syn.synthetic_latents(MEMBERS, 96, 96, seed=LAT_SEED)
    out = {}
    with torch.no_grad():
        rgb_latent = opipe.encode_rgb(vae, rgb)
        tr = []
        x0 = opipe.denoise(unet, DDIMScheduler(), rgb_latent, lat0[:1], ctx, 10, trace=tr)
        out["ddim10_trace_m0"] = np.concatenate([t[2].numpy() for t in tr])
        print(f"[{time.time() - t0:7.1f}s] denoise done; |latent| rms per step "
              + " ".join(f"{float((t[2] ** 2).mean().sqrt()):.3f}" for t in tr), flush=True)
        d = opipe.decode_latent(vae, x0).mean(dim=1, keepdim=True)
        out["ddim10_depth_m0"] = ((torch.clip(d, -1.0, 1.0) + 1.0) / 2.0)[0, 0].numpy().astype(np.float16)
    path = os.path.join(GOLD, "fullsize_768_heavy.npz")
    np.savez_compressed(path, **out)
    print("fullsize_768_heavy.npz:", {k: (v.shape, str(v.dtype)) for k, v in out.items()}, os.path.getsize(path) >> 10, "KiB",
          f"{time.time() - t0:.0f}s")


def main():
    from marigold_amd import synthetic as syn
    from oracle import ensemble as oens, pipeline as opipe
    from oracle.schedulers import DDIMScheduler, LCMScheduler
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL

    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "8")))
    t0 = time.time()
    unet = UNet2DConditionModel().eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict())
    vae = AutoencoderKL().eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict())
    ctx = syn.synthetic_text_embedding()
    img = syn.synthetic_image(768, 768, seed=IMG_SEED)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    lat0 = syn.synthetic_latents(MEMBERS, 96, 96, seed=LAT_SEED)
    out = {}

    def log(msg):
        print(f"[{time.time() - t0:7.1f}s] {msg}", flush=True)

    with torch.no_grad():
        rgb_latent = opipe.encode_rgb(vae, rgb)
        out["rgb_latent"] = rgb_latent.numpy()
        log("encode done")
        finals = []
        for e in range(MEMBERS):
            tr = []
            x0 = opipe.denoise(unet, DDIMScheduler(), rgb_latent, lat0[e:e + 1], ctx, 10, trace=tr)
            finals.append(x0)
            if e == 0:
                out["ddim10_pred_m0"] = np.concatenate([t[1].numpy() for t in tr])
                out["ddim10_trace_m0"] = np.concatenate([t[2].numpy() for t in tr])
            log(f"ddim member {e} done")
        finals = torch.cat(finals)
        out["ddim10_final"] = finals.numpy()
        normals = []
        for e in range(MEMBERS):
            dec = opipe.decode_latent(vae, finals[e:e + 1])
            if e == 0:
                d = dec.mean(dim=1, keepdim=True)
                out["ddim10_depth_m0"] = ((torch.clip(d, -1.0, 1.0) + 1.0) / 2.0)[0, 0].numpy().astype(np.float16)
            n = torch.clip(dec, -1.0, 1.0)
            normals.append(n / torch.norm(n, dim=1, keepdim=True).clamp(min=1e-6))
            log(f"decode member {e} done")
        normals = torch.cat(normals)
        out["ddim10_normals_m0"] = normals[0].numpy().astype(np.float16)
        ens, _ = oens.ensemble_normals(normals)
        out["ddim10_normals_e4"] = ens[0].numpy().astype(np.float16)
        # C4: LCM, 4 steps
        g = torch.Generator("cpu").manual_seed(LCM_NOISE_SEED)
        tr = []
        xl = opipe.denoise(unet, LCMScheduler(), rgb_latent, lat0[:1], ctx, 4, generator=g, trace=tr)
        out["lcm4_trace"] = np.concatenate([t[2].numpy() for t in tr])
        d = opipe.decode_latent(vae, xl).mean(dim=1, keepdim=True)
        out["lcm4_depth"] = ((torch.clip(d, -1.0, 1.0) + 1.0) / 2.0)[0, 0].numpy().astype(np.float16)
        log("lcm done")
    out["meta"] = np.array([IMG_SEED, LAT_SEED, LCM_NOISE_SEED, MEMBERS])
    path = os.path.join(GOLD, "fullsize_768.npz")
    np.savez_compressed(path, **out)
    print("fullsize_768.npz:", {k: (v.shape, str(v.dtype)) for k, v in out.items()}, os.path.getsize(path) >> 10, "KiB")


if __name__ == "__main__":
    main_heavy() if "--heavy" in sys.argv else main()
