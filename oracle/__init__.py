"""CPU oracle for the Marigold inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``marigold_amd``) never imports this package and fails loudly when its HIP
library is missing.

What it restates (plain PyTorch fp32 on CPU):

* ``sd2_unet``    - diffusers ``UNet2DConditionModel`` in the SD-v2 / Marigold
                    configuration (8-ch ``conv_in``), called at
                    /root/reference/marigold/marigold_depth_pipeline.py:461-463
* ``sd2_vae``     - diffusers ``AutoencoderKL`` encoder / decoder, called at
                    marigold_depth_pipeline.py:491-492, 512-513
* ``schedulers``  - diffusers ``DDIMScheduler`` / ``LCMScheduler``
                    (marigold_depth_pipeline.py:423-424, 466-468)
* ``ensemble``    - marigold/util/ensemble.py:39-249
* ``pipeline``    - control flow of marigold_depth_pipeline.py:396-516 and
                    marigold_normals_pipeline.py:361-479

Pinning status
--------------
* ``ensemble``: PINNED - checked against outputs of the reference's own
  ``marigold/util/ensemble.py`` executed in the build container
  (``oracle/make_golden.py`` -> ``tests/golden/ensemble_*.npz``).
* ``schedulers``: pinned against the closed-form timestep tables the reference
  relies on - the diffusers package itself is absent.
* ``sd2_unet`` / ``sd2_vae``: **parity unpinned**.  diffusers (un-vendored
  dependency, ``diffusers>=0.25.0`` in /root/reference/requirements.txt) is not
  installed in this environment and no checkpoint exists on disk; the modules
  are restated from the published architecture and checked structurally
  (parameter counts 865.92 M / 34.16 M / 49.49 M, state-dict key scheme).
"""
