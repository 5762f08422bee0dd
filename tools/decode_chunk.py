#!/usr/bin/env python
"""Times the VAE decode of E=10 members at 768^2 for different members-per-launch-group values
(activations of a 10-member batch are 1.5-3 GB per tensor, far beyond the 256 MB Infinity Cache; smaller
groups keep a layer's working set closer to it).  Tuning tool, run on the MI355X."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, synthetic as syn  # noqa: E402
from marigold_amd.arch import VAEConfig  # noqa: E402
from marigold_amd.modules import AutoencoderKLHIP  # noqa: E402

vae = AutoencoderKLHIP(syn.synthetic_vae_state_dict(VAEConfig()), VAEConfig()).to("cuda:0")
lat = syn.synthetic_latents(10, 96, 96, seed=3).cuda()
for chunk in (10, 5, 2, 1):
    vae.decode_chunk = chunk
    vae.decode(lat, post=L.POST_DEPTH)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = vae.decode(lat, post=L.POST_DEPTH)
    torch.cuda.synchronize()
    print(f"decode E=10 @768^2, {chunk:2d} members per group: {(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms", flush=True)
