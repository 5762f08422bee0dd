"""Engine handles that stand where the reference's ``unet`` / ``vae`` diffusers modules stand
(``register_modules`` at marigold_depth_pipeline.py:133-139).  They own the device-resident
weights and a cache of native programs keyed by problem shape; calling them runs HIP kernels
only (there is no torch fallback: without the HIP library construction fails).
"""
from types import SimpleNamespace

import torch

from . import _lib as L
from . import engine as E
from . import ops as O
from .arch import UNetConfig, VAEConfig, unet_param_shapes, vae_param_shapes


def _check_state_dict(sd, shapes, what):
    missing = [k for k in shapes if k not in sd]
    if missing:
        raise KeyError(f"{what}: state dict lacks {len(missing)} tensors, e.g. {missing[:4]}")
    for k, s in shapes.items():
        if tuple(sd[k].shape) != tuple(s):
            raise ValueError(f"{what}: {k} has shape {tuple(sd[k].shape)}, expected {tuple(s)}")


def _device_index(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the Marigold HIP engine runs on an MI355X only (device must be 'cuda[:i]')")
    return device.index if device.index is not None else torch.cuda.current_device()


class _EngineModule:
    # The 16-bit operand type of the engine (fp32 accumulation either way): bf16 = libmarigold_hip.so, the product build;
    # torch.float16 = libmarigold_hip_f16.so, the same kernels on fp16 operands - the arithmetic of the reference's
    # `--fp16` / from_pretrained(torch_dtype=torch.float16) (script/depth/run.py:203-211, marigold_depth_pipeline.py:253, 433).
    compute_dtype = torch.bfloat16

    def __init__(self, compute_dtype=torch.bfloat16):
        if compute_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(f"compute_dtype={compute_dtype}: the engine's operands are torch.bfloat16 or torch.float16")
        self.compute_dtype = compute_dtype
        self.device = torch.device("cpu")
        self.ws = None
        self.pool = None
        self._programs = {}

    @property
    def dtype(self):
        return self.compute_dtype

    @property
    def f16(self):
        return self.compute_dtype == torch.float16

    def _seq(self, name):
        return O.OpSeq(name, f16=self.f16)

    def to(self, device):
        idx = _device_index(device)
        L.init(idx)             # dtype-independent ops (ensembling, resampling, colour table) launch from the product library
        if self.f16:
            L.init(idx, True)
        self.device = torch.device("cuda", idx)
        self.ws = E.WeightStore(self.sd, self.device, self.compute_dtype)
        self.pool = E.Pool(self.device)
        self._programs = {}
        return self

    def dry(self):
        """Host-only mode for contract checks (tests): programs are built against CPU buffers and
        can be ``validate()``d, never run."""
        self.device = torch.device("cpu")
        self.ws = E.WeightStore(self.sd, self.device, self.compute_dtype)
        self.pool = E.Pool(self.device)
        self._programs = {}
        return self

    def replica(self):
        """A second engine over the SAME device-resident weights: its own workspace pool, programs and every piece of
        launch-private state a program carries (tickets, split-K and key-split workspaces) - what a map needs to run on another
        HIP stream while this engine's map is in flight (pipeline.map_images)."""
        self._require_device()
        if self.device.type != "cuda":
            raise RuntimeError("replica(): the engine is not on a GPU")
        r = object.__new__(type(self))
        r.__dict__.update(self.__dict__)
        r.pool = E.Pool(self.device)
        r._programs = {}
        return r

    def _require_device(self):
        if self.ws is None:
            raise RuntimeError(f"{type(self).__name__}: call .to('cuda') first")

    def workspace_bytes(self):
        return 0 if self.pool is None else self.pool.bytes


class DenoiseProgram:
    """One native program = T x (UNet forward + scheduler update) for fixed (B, h, w, timesteps)."""

    def __init__(self, seq, rgb_latent, x, eps, noises, n_fwd_ops, n_prologue_ops):
        self.seq, self.rgb_latent, self.x, self.eps, self.noises = seq, rgb_latent, x, eps, noises
        self.n_fwd_ops, self.n_prologue_ops = n_fwd_ops, n_prologue_ops

    def run(self):
        self.seq.run()


class UNet2DConditionModelHIP(_EngineModule):
    """SD-v2 UNet (8-channel conv_in).  ``__call__`` mirrors
    ``unet(sample, t, encoder_hidden_states=ctx).sample`` (marigold_depth_pipeline.py:461-463);
    ``denoise_program`` builds the whole T-step loop (:455-468) as one program."""

    def __init__(self, state_dict, config: UNetConfig = UNetConfig(), compute_dtype=torch.bfloat16):
        super().__init__(compute_dtype)
        _check_state_dict(state_dict, unet_param_shapes(config), "UNet")
        self.sd = state_dict
        self.config = config
        self._ctx = None

    def set_context(self, ctx):
        """ctx: the empty-prompt embedding [1,2,D] (or [2,D]); constant per checkpoint."""
        ctx = ctx.detach().float().cpu().reshape(-1, ctx.shape[-1])
        if ctx.shape != (2, self.config.cross_attention_dim):
            raise ValueError(f"expected a 2-token context of width {self.config.cross_attention_dim}, "
                             f"got {tuple(ctx.shape)}")
        if self._ctx is None or not torch.equal(self._ctx, ctx):
            self._ctx = ctx
            self._programs = {}
            if self.ws is not None:
                for k in [k for k in self.ws.cache if k[0] == "x"]:
                    del self.ws.cache[k]

    def denoise_program(self, B, h, w, scheduler, n_steps, rgb_broadcast=True, with_scheduler=True,
                        timesteps=None):
        self._require_device()
        if self._ctx is None:
            raise RuntimeError("UNet context not set (set_context)")
        if timesteps is None:
            scheduler.set_timesteps(n_steps)
            timesteps = [int(t) for t in scheduler.timesteps]
        key = (B, h, w, tuple(timesteps), rgb_broadcast, with_scheduler,
               scheduler.signature() if with_scheduler else None)
        if key in self._programs:
            return self._programs[key]
        dev = self.device
        seq = self._seq(f"denoise[B={B},{h}x{w},T={len(timesteps)}]")
        bld = E.Builder(seq, self.pool, self.ws, self.config.norm_groups)
        rgb_latent = torch.zeros(1 if rgb_broadcast else B, 4, h, w, device=dev)
        x = torch.zeros(B, self.config.out_channels, h, w, device=dev)
        eps = torch.zeros(B, self.config.out_channels, h, w, device=dev)
        seq.hold(rgb_latent, x, eps)
        table = E.emit_time_embeddings(bld, self.config, timesteps)
        n_pro = len(seq)
        noises = []
        n_fwd = 0
        for i in range(len(timesteps)):
            n0 = len(seq)
            sched = None
            if with_scheduler:   # the scheduler update (reference :466-468) is the tail of conv_out's pointwise pass
                cx, cm, cn = scheduler.step_coefficients(i)
                nz = None
                if scheduler.needs_noise(i):
                    nz = seq.hold(torch.zeros(B, self.config.out_channels, h, w, device=dev))
                    noises.append(nz)
                sched = (cx, cm, cn, nz)
            E.emit_unet_forward(bld, self.config, self._ctx, rgb_latent, x, eps, table, i, B, h, w, sched=sched)
            n_fwd = len(seq) - n0
        seq.keep.extend(bld.persist.values())
        seq.zero_state = {t.data_ptr() for t in bld.persist.values()}   # Builder.zeros_persistent: state the kernels expect zeroed
        prog = DenoiseProgram(seq, rgb_latent, x, eps, noises, n_fwd, n_pro)
        self._programs[key] = prog
        return prog

    def __call__(self, sample, timestep, encoder_hidden_states=None):
        """Step-wise mirror (used by parity tests): sample [B,8,h,w] -> .sample [B,4,h,w] fp32."""
        if encoder_hidden_states is not None:
            self.set_context(encoder_hidden_states[:1])
        B, _, h, w = sample.shape
        prog = self.denoise_program(B, h, w, None, 1, rgb_broadcast=False, with_scheduler=False,
                                    timesteps=[int(timestep)])
        s = sample.to(self.device, torch.float32)
        prog.rgb_latent.copy_(s[:, :4])
        prog.x.copy_(s[:, 4:])   # 4 channels per predicted modality
        prog.run()
        return SimpleNamespace(sample=prog.eps.clone())


class AutoencoderKLHIP(_EngineModule):
    """SD AutoencoderKL.  ``encode_rgb_latent`` = 0.18215 * mean(quant_conv(encoder(x)))
    (marigold_depth_pipeline.py:491-495); ``decode`` = decoder(post_quant_conv(z / 0.18215)) with
    the pipeline's pointwise tail fused (:510-515, :473-475 / normals :437-440)."""

    def __init__(self, state_dict, config: VAEConfig = VAEConfig(), compute_dtype=torch.bfloat16):
        super().__init__(compute_dtype)
        _check_state_dict(state_dict, vae_param_shapes(config), "VAE")
        self.sd = state_dict
        self.config = config

    def _program(self, kind, B, H, W, post=0):
        key = (kind, B, H, W, post)
        if key in self._programs:
            return self._programs[key]
        dev = self.device
        seq = self._seq(f"vae.{kind}[B={B},{H}x{W}]")
        bld = E.Builder(seq, self.pool, self.ws, self.config.norm_groups)
        if kind == "encode":
            n_down = len(self.config.block_out_channels) - 1
            h, w = H, W
            for _ in range(n_down):
                h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
            inp = torch.zeros(B, 3, H, W, device=dev)
            out = torch.zeros(B, self.config.latent_channels, h, w, device=dev)
            E.emit_vae_encode(bld, self.config, inp, out, B, H, W)
        else:
            f = 2 ** (len(self.config.block_out_channels) - 1)
            inp = torch.zeros(B, self.config.latent_channels, H, W, device=dev)
            cout = 1 if post == L.POST_DEPTH else 3
            out = torch.zeros(B, cout, H * f, W * f, device=dev)
            E.emit_vae_decode(bld, self.config, inp, out, B, H, W, post)
        seq.hold(inp, out)
        seq.keep.extend(bld.persist.values())
        seq.zero_state = {t.data_ptr() for t in bld.persist.values()}   # Builder.zeros_persistent: state the kernels expect zeroed
        self._programs[key] = (seq, inp, out)
        return self._programs[key]

    def encode_rgb_latent(self, rgb):
        """rgb [B,3,H,W] in [-1,1] (any float dtype, on device) -> scaled latent mean [B,4,h,w] fp32."""
        self._require_device()
        B, _, H, W = rgb.shape
        seq, inp, out = self._program("encode", B, H, W)
        inp.copy_(rgb)
        seq.run()
        return out.clone()

    decode_chunk = 0   # members per decode launch group; 0 = all at once (VAE batching)

    def decode(self, latent, post=L.POST_NONE):
        """latent [B,4,h,w] fp32 -> decoded map fp32 ([B,3,H,W], or [B,1,H,W] for POST_DEPTH)."""
        self._require_device()
        B, _, h, w = latent.shape
        chunk = self.decode_chunk if 0 < self.decode_chunk < B else B
        outs = []
        for i in range(0, B, chunk):
            part = latent[i:i + chunk]
            seq, inp, out = self._program("decode", part.shape[0], h, w, post)
            inp.copy_(part)
            seq.run()
            outs.append(out.clone())
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
