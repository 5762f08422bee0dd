"""``MarigoldDepthPipeline`` / ``MarigoldNormalsPipeline`` with the reference's call surface
(marigold/marigold_depth_pipeline.py:154-338, marigold/marigold_normals_pipeline.py:139-308):
same arguments, defaults, asserts / exceptions / warnings and output containers, on top of the
MI355X engine (HIP programs for VAE encode, the whole T-step denoising loop, VAE decode and the
ensembling).  Differences that are deliberate and documented in DESIGN.md:

* the image is VAE-encoded once per call, not once per ensemble member (the reference encodes E
  identical copies, :258 / :427);
* all members of a batch run through one native denoising program (no per-step Python);
* optional member parallelism over the GPUs of a node (``enable_member_parallel``): members are
  sharded over ranks and collected with ONE gather (RCCL over xGMI) before aggregation;
* ``init_latents`` (extension) lets callers supply the initial noise for parity runs.
"""
import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch
from PIL import Image

from . import _lib as L
from . import dist as mdist
from .ensemble import ensemble_depth, ensemble_iid, ensemble_normals
from .modules import AutoencoderKLHIP, UNet2DConditionModelHIP
from .schedulers import DDIMScheduler, LCMScheduler
from .util.batchsize import find_batch_size
from .util.image_util import (chw2hwc, colorize_depth_device, colorize_depth_maps, get_tv_resample_method, pil_to_tensor,
                              resize, resize_max_res)


@dataclass
class MarigoldDepthOutput:
    """depth_np [H,W] in [0,1]; depth_colored PIL RGB | None; uncertainty [H,W] | None
    (reference :60-75)."""
    depth_np: np.ndarray
    depth_colored: Union[None, Image.Image]
    uncertainty: Union[None, np.ndarray]


@dataclass
class MarigoldNormalsOutput:
    """normals_np [3,H,W] unit vectors in [-1,1]; normals_img PIL; uncertainty | None
    (normals reference :59-74)."""
    normals_np: np.ndarray
    normals_img: Image.Image
    uncertainty: Union[None, np.ndarray]


class _Turnstile:
    """Calls pass in index order, one at a time: ``wait(k)`` returns once 0 ... k - 1 are ``done``."""

    def __init__(self):
        import threading
        self._cv = threading.Condition()
        self._next = 0
        self._error = None

    def wait(self, k):
        with self._cv:
            self._cv.wait_for(lambda: self._next == k or self._error is not None)
            if self._error is not None:
                raise RuntimeError("map_images: another lane failed before its gather") from self._error

    def done(self, k):
        with self._cv:
            if self._next == k:
                self._next = k + 1
            self._cv.notify_all()

    def abort(self, error):
        with self._cv:
            self._error = error
            self._cv.notify_all()


class _MarigoldPipelineBase:
    latent_scale_factor = 0.18215
    _kind = "depth"
    _ckpt_hint = "prs-eth/marigold-depth-v1-1"
    _target_latent_channels = 4    # latent channels the UNet predicts (4 per modality)
    _pred_channels = 1             # channels of one decoded prediction

    def __init__(self, unet: UNet2DConditionModelHIP, vae: AutoencoderKLHIP,
                 scheduler: Union[DDIMScheduler, LCMScheduler], text_encoder=None, tokenizer=None,
                 scale_invariant: Optional[bool] = True, shift_invariant: Optional[bool] = True,
                 default_denoising_steps: Optional[int] = None,
                 default_processing_resolution: Optional[int] = None, empty_text_embed=None):
        self.unet, self.vae, self.scheduler = unet, vae, scheduler
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.scale_invariant = scale_invariant
        self.shift_invariant = shift_invariant
        self.default_denoising_steps = default_denoising_steps
        self.default_processing_resolution = default_processing_resolution
        self.config = dict(scale_invariant=scale_invariant, shift_invariant=shift_invariant,
                           default_denoising_steps=default_denoising_steps,
                           default_processing_resolution=default_processing_resolution)
        self.empty_text_embed = empty_text_embed
        self._member_group = None
        self._member_parallel = False

    # ---- diffusers.DiffusionPipeline surface the callers use ---------------------------------
    @property
    def device(self):
        return self.unet.device

    @property
    def dtype(self):
        """Compute dtype of the engine (bf16 operands, fp32 accumulation)."""
        return self.unet.dtype

    # Tensors at the pipeline boundary - the normalised image, the initial latents, the LCM per-step noise, the
    # predictions - are fp32 like the reference's default pipeline (script/depth/run.py:203-215 loads fp32 unless
    # --fp16); the engine's boundary convolutions convert them on the fly, so nothing is rounded to bf16 on the way in.
    io_dtype = torch.float32
    # The reference draws the initial latents and the LCM per-step noise in the MODEL dtype (:430-435:
    # ``torch.randn(..., dtype=self.dtype, generator=generator)``): a pipeline loaded with
    # ``from_pretrained(torch_dtype=torch.float16 | torch.bfloat16)`` (script/depth/run.py:203-214, ``--fp16``) consumes the
    # generator as 16-bit draws - a different random stream from the fp32 one.  ``noise_dtype`` reproduces that choice:
    # fp32 by default (the reference's default load), the caller's ``torch_dtype`` when one was given; the draws are
    # widened to fp32 on their way into the engine's latent buffers (exact for both 16-bit types).
    noise_dtype = torch.float32

    def _randn(self, shape, generator):
        return torch.randn(tuple(shape), device=self.device, dtype=self.noise_dtype, generator=generator).to(self.io_dtype)

    def to(self, device):
        self.unet.to(device)
        self.vae.to(device)
        self._lanes = None   # engine replicas of map_images belong to the device they were made on
        return self

    @classmethod
    def from_pretrained(cls, path, variant=None, torch_dtype=None, **kw):
        from .checkpoint import load_pipeline
        return load_pipeline(cls, path, variant=variant, torch_dtype=torch_dtype, **kw)

    def enable_xformers_memory_efficient_attention(self):
        """No-op: the engine's attention is already a fused flash kernel (run.py:217-220)."""

    def set_progress_bar_config(self, **kw):
        pass

    def enable_member_parallel(self, group=None, root=None, force_collective=False):
        """Shard ensemble members over the ranks of ``group`` (torch.distributed; RCCL on GPUs).  ``force_collective``:
        take the sharded path - full-E noise draw sliced by member, ONE gather - even in a group of one rank (the RCCL
        path on a single GPU; results are those of the plain path bit for bit)."""
        self._member_group = group
        self._member_parallel = True
        self._member_root = root
        self._member_force = bool(force_collective) and mdist.is_on(group)

    def _sharded(self):
        return self._member_parallel and (mdist.world_size(self._member_group) > 1 or getattr(self, "_member_force", False))

    # ---- maps in flight ----------------------------------------------------------------------
    # The reference's scripts call the pipeline image by image (script/depth/run.py:231-262, script/depth/infer.py): every map waits
    # for the one before it.  One map alone leaves the MI355X partly idle wherever a launch is a single lockstep round of workgroups
    # whose HBM-bound epilogues follow their K loops, has a part-filled last round, or is a host-driven chain (the alignment
    # optimiser's ~80 evaluations per map): a SECOND, independent map on another HIP stream fills those holes (DESIGN.md section 6b;
    # tools/inflight_bench.py: +9 % maps/s at E = 10, every map bit-identical to the one-at-a-time result).  288 GB of HBM hold the
    # second set of workspaces (23 GB at 768 x 768, E = 10) many times over; the weights are shared.
    # A GPU that holds eight members or fewer (small ensembles, the shards of the member-parallel path) has more and longer holes
    # per map - its launches are single part-filled rounds - and takes a THIRD lane: same box, interleaved
    # (profiles/r6_inflight_by_ensemble.log) E = 1 / 2 / 3 / 5 / 6 / 8: 51.4 / 77.5 / 100.6 / 146.5 / 172.6 / 211.8 ms per map with two
    # lanes, 45.0 / 70.3 / 95.0 / 143.1 / 165.7 / 208.2 with three (a fourth: 48.4 / 70.7 / 94.2 / 142.4); at E = 10 the third lane
    # buys 0.4-0.5 %.
    default_maps_in_flight = 2
    small_ensemble_maps_in_flight = 3
    small_ensemble_members = 8

    def maps_in_flight_for(self, ensemble_size: int = 1) -> int:
        """The lane count ``map_images`` uses when the caller names none: by the members THIS GPU runs per map (the ensemble
        divided over the ranks of a member-parallel pipeline)."""
        world = mdist.world_size(self._member_group) if self._sharded() else 1
        local = -(-max(1, int(ensemble_size)) // max(1, world))
        return self.small_ensemble_maps_in_flight if local <= self.small_ensemble_members else self.default_maps_in_flight

    def replicate(self):
        """Another pipeline over the same device-resident weights: engine replicas (own workspaces, programs, launch-private
        state) and its own scheduler object - what ``map_images`` runs a second map on."""
        import copy
        r = copy.copy(self)
        r.unet, r.vae = self.unet.replica(), self.vae.replica()
        r.scheduler = copy.deepcopy(self.scheduler)
        r._lanes = None
        return r

    def _lane_pipelines(self, n):
        lanes = getattr(self, "_lanes", None)
        if lanes is None:
            lanes = self._lanes = [(self, torch.cuda.Stream(device=self.device))]
        while len(lanes) < n:
            lanes.append((self.replicate(), torch.cuda.Stream(device=self.device)))
        return lanes[:n]

    def map_images(self, images, in_flight: Optional[int] = None, generators=None, **call_kwargs):
        """``(pipe(image, **call_kwargs) for image in images)`` with up to ``in_flight`` maps on the GPU at a time (default
        ``maps_in_flight_for(ensemble_size)``; 1 = one after the other on the caller's stream).  A generator: outputs come in input order
        as they complete, and ``images`` (any iterable) is consumed as lanes become free.  ``generators``: one
        ``torch.Generator`` (or None) per image - with several maps in flight a single shared generator would be consumed in
        completion order, so ``generator=`` is refused; every map is then bit-identical to what ``pipe(image, generator=g)``
        returns on its own.  Member-parallel pipelines (several ranks): every rank must call this with the same images and
        ``in_flight``; the lanes then issue their gathers strictly in map order, one at a time (``_Turnstile``), so the collective
        sequence is the same on every rank whichever lane finishes first."""
        n = self.maps_in_flight_for(call_kwargs.get("ensemble_size", 1)) if in_flight is None else int(in_flight)
        if n < 1:
            raise ValueError(f"in_flight must be >= 1 (got {in_flight})")
        if generators is not None and hasattr(images, "__len__") and hasattr(generators, "__len__") and len(images) != len(generators):
            raise ValueError(f"{len(generators)} generators for {len(images)} images")
        if hasattr(images, "__len__"):
            n = min(n, max(1, len(images)))
        if self.device.type != "cuda":
            n = 1
        if n > 1 and call_kwargs.get("generator") is not None:
            raise ValueError("map_images: pass `generators` (one per image) instead of a shared `generator` when in_flight > 1")
        return self._map_images(iter(images), None if generators is None else iter(generators), n, call_kwargs)

    def _map_images(self, images, generators, n, call_kwargs):
        import threading
        lock = threading.Lock()
        count = [0]

        def take():
            """-> (index, image, generator) | None; under the lock: the two iterables advance together"""
            with lock:
                try:
                    image = next(images)
                except StopIteration:
                    return None
                g = None
                if generators is not None:
                    try:
                        g = next(generators)
                    except StopIteration:
                        raise ValueError("map_images: fewer generators than images") from None
                k = count[0]
                count[0] += 1
                return k, image, g

        turnstile = _Turnstile() if (n > 1 and self._sharded()) else None

        def one(pipe, image, g, k=0):
            kw = dict(call_kwargs)
            if generators is not None:
                kw["generator"] = g
            pipe._gather_turn = None if turnstile is None else (turnstile, k)
            try:
                return pipe(image, **kw)
            finally:
                pipe._gather_turn = None

        if n == 1:
            while (item := take()) is not None:
                yield one(self, item[1], item[2])
            return
        lanes = self._lane_pipelines(n)
        caller = torch.cuda.current_stream(self.device)
        done = {}
        cv = threading.Condition()
        stop = threading.Event()
        live = [len(lanes)]

        def work(pipe, stream):
            try:
                torch.cuda.set_device(self.device)
                stream.wait_stream(caller)   # inputs the caller produced on its stream
                with torch.cuda.stream(stream):
                    while not stop.is_set():
                        item = take()
                        if item is None:
                            break
                        out = one(pipe, item[1], item[2], item[0])
                        with cv:
                            done[item[0]] = out
                            cv.notify_all()
            except BaseException as e:  # noqa: BLE001 - handed to the caller's thread
                if turnstile is not None:
                    turnstile.abort(e)   # lanes waiting for their turn must not wait for a gather that will never be issued
                with cv:
                    done.setdefault("error", e)
                    cv.notify_all()
            finally:
                with cv:
                    live[0] -= 1
                    cv.notify_all()

        threads = [threading.Thread(target=work, args=lane, daemon=True) for lane in lanes]
        for t in threads:
            t.start()
        try:
            k = 0
            while True:
                with cv:
                    cv.wait_for(lambda: k in done or "error" in done or live[0] == 0)
                    if "error" in done:
                        raise done["error"]
                    if k not in done:
                        break   # every lane has finished and map k was never started: the input is exhausted
                    out = done.pop(k)
                yield out
                k += 1
        finally:
            stop.set()
            for t in threads:
                t.join()
            for _, stream in lanes:
                caller.wait_stream(stream)

    # ---- reference methods -------------------------------------------------------------------
    def _check_inference_step(self, n_step: int) -> None:
        assert n_step >= 1
        if isinstance(self.scheduler, DDIMScheduler):
            if "trailing" != self.scheduler.config.timestep_spacing:
                logging.warning(
                    f"The loaded `DDIMScheduler` is configured with `timestep_spacing="
                    f'"{self.scheduler.config.timestep_spacing}"`; the recommended setting is `"trailing"`. '
                    f"This change is backward-compatible and yields better results. "
                    f"Consider using `{self._ckpt_hint}` for the best experience.")
            else:
                if n_step > 10:
                    logging.warning(
                        f"Setting too many denoising steps ({n_step}) may degrade the prediction; consider "
                        f"relying on the default values.")
            if not self.scheduler.config.rescale_betas_zero_snr:
                logging.warning(
                    f"The loaded `DDIMScheduler` is configured with `rescale_betas_zero_snr="
                    f"{self.scheduler.config.rescale_betas_zero_snr}`; the recommended setting is True. "
                    f"Consider using `{self._ckpt_hint}` for the best experience.")
        elif isinstance(self.scheduler, LCMScheduler):
            self._lcm_policy(n_step)
        else:
            raise RuntimeError(f"Unsupported scheduler type: {type(self.scheduler)}")

    def encode_empty_text(self):
        """CLIP("") with padding="do_not_pad" -> [1,2,D]; constant per checkpoint, computed once on
        the host (reference :381-394)."""
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("no text encoder/tokenizer and no precomputed empty_text_embed")
        text_inputs = self.tokenizer("", padding="do_not_pad", max_length=self.tokenizer.model_max_length,
                                     truncation=True, return_tensors="pt")
        with torch.no_grad():
            self.empty_text_embed = self.text_encoder(text_inputs.input_ids)[0].to(self.dtype)

    def encode_rgb(self, rgb_in: torch.Tensor) -> torch.Tensor:
        return self.vae.encode_rgb_latent(rgb_in)

    @torch.no_grad()
    def single_infer(self, rgb_in: torch.Tensor, num_inference_steps: int,
                     generator: Union[torch.Generator, None], show_pbar: bool = False,
                     init_latents: Optional[torch.Tensor] = None,
                     step_noises: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One batched prediction (reference :396-477).  rgb_in [B,3,h,w] in [-1,1]; identical
        (expanded) rows are encoded once."""
        device = self.device
        B = rgb_in.shape[0]
        shared = B == 1 or rgb_in.stride(0) == 0
        rgb_in = (rgb_in[:1] if shared else rgb_in).to(device)
        rgb_latent = self.encode_rgb(rgb_in)                       # [1|B,4,h,w] fp32
        h, w = rgb_latent.shape[-2:]
        if init_latents is None:
            target_latent = self._randn((B, self._target_latent_channels, h, w), generator)
        else:
            target_latent = init_latents.to(device)
        if self.empty_text_embed is None:
            self.encode_empty_text()
        self.unet.set_context(self.empty_text_embed)
        prog = self.unet.denoise_program(B, h, w, self.scheduler, num_inference_steps,
                                         rgb_broadcast=shared)
        prog.rgb_latent.copy_(rgb_latent)
        prog.x.copy_(target_latent)
        for k, nz in enumerate(prog.noises):  # LCM consumes the generator once per non-final step (:466-468)
            if step_noises is not None:
                nz.copy_(step_noises[k])
            else:
                nz.copy_(self._randn(nz.shape, generator))
        prog.run()
        return self._decode(prog.x)

    def _predict_members(self, rgb_norm, ensemble_size, denoising_steps, batch_size, generator,
                         init_latents):
        """All E members of one image -> [E,C,h,w] (on every rank when member-parallel)."""
        _bs = batch_size if batch_size > 0 else find_batch_size(
            ensemble_size=ensemble_size, input_res=max(rgb_norm.shape[1:]), dtype=self.dtype)
        E = ensemble_size
        members = list(range(E))
        step_noises_all = None
        if self._sharded():
            # every rank draws the full [E,4,h,w] noise (same generator state) and keeps its slice, so results do not depend on
            # the number of GPUs: they are what ONE process draws when its batch holds all E members (batch_size >= E, the
            # default on this hardware).  With a smaller batch_size the reference draws batch by batch (:281-289), and so
            # does the single-process path below - the reference's own results depend on the batch size in that case
            if init_latents is None:
                hh, ww = self._latent_hw(rgb_norm.shape[-2:])
                init_latents = self._randn((E, self._target_latent_channels, hh, ww), generator)
            # the LCM scheduler consumes the generator once per non-final step (:466-468): those draws are made for
            # all E members on every rank too, in the order a single process holding the E members in one batch makes
            # them (initial latents, then one [E,...] draw per step), and sliced by member
            self.scheduler.set_timesteps(denoising_steps)
            n_noise = sum(bool(self.scheduler.needs_noise(i)) for i in range(denoising_steps))
            if n_noise:
                step_noises_all = [self._randn(init_latents.shape, generator) for _ in range(n_noise)]
            members = mdist.shard_members(E, mdist.world_size(self._member_group),
                                          mdist.rank(self._member_group))
        preds = []
        for i in range(0, len(members), _bs):
            idx = members[i:i + _bs]
            lat = None if init_latents is None else init_latents[idx]
            rgb = rgb_norm.expand(len(idx), -1, -1, -1)
            nzs = None if step_noises_all is None else [nz[idx] for nz in step_noises_all]
            preds.append(self.single_infer(rgb, denoising_steps, generator, False, lat, step_noises=nzs))
        local = torch.cat(preds, dim=0) if preds else None
        if self._sharded():
            C = self._pred_channels
            # decoded maps are latent size x 2^(levels-1), which is smaller than the image when its size is not a
            # multiple of 8 (KITTI 1242x375 -> 768x231 -> latent 96x28 -> decoded 768x224); ranks without members
            # need the shape too, so it is computed, not taken from `local`
            f = 2 ** (len(self.vae.config.block_out_channels) - 1)
            hh, ww = (f * d for d in self._latent_hw(rgb_norm.shape[-2:]))
            # maps in flight: the gather is a collective on ONE process group - every rank issues the gathers of maps 0, 1, 2 ...
            # in that order, one at a time, whichever lane (thread, stream) predicted them (map_images hands each call its turn)
            turn = getattr(self, "_gather_turn", None)
            if turn is not None:
                turn[0].wait(turn[1])
            try:
                return mdist.gather_members(local, E, (C, hh, ww), self.device, self._member_group,
                                            getattr(self, "_member_root", None), force=getattr(self, "_member_force", False))
            finally:
                if turn is not None:
                    turn[0].done(turn[1])
        return local

    def _latent_hw(self, hw):
        h, w = hw
        for _ in range(len(self.vae.config.block_out_channels) - 1):
            h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
        return h, w

    def _preprocess(self, input_image, processing_res, resample_method):
        if isinstance(input_image, Image.Image):
            input_image = input_image.convert("RGB")
            rgb = pil_to_tensor(input_image).unsqueeze(0)
        elif isinstance(input_image, torch.Tensor):
            rgb = input_image
        else:
            raise TypeError(f"Unknown input type: {type(input_image) = }")
        input_size = rgb.shape
        assert 4 == rgb.dim() and 3 == input_size[-3], f"Wrong input shape {input_size}, expected [1, rgb, H, W]"
        if processing_res > 0:
            if max(input_size[-2:]) != processing_res and self.device.type == "cuda":
                rgb = rgb.to(self.device)   # resample on the device (csrc/resize.hip)
            rgb = resize_max_res(rgb, max_edge_resolution=processing_res, resample_method=resample_method)
        rgb_norm = rgb / 255.0 * 2.0 - 1.0
        rgb_norm = rgb_norm.to(self.io_dtype)
        assert rgb_norm.min() >= -1.0 and rgb_norm.max() <= 1.0
        return rgb_norm, input_size


class MarigoldDepthPipeline(_MarigoldPipelineBase):
    """Affine-invariant monocular depth (reference marigold/marigold_depth_pipeline.py:78-516)."""
    _kind = "depth"
    _ckpt_hint = "prs-eth/marigold-depth-v1-1"

    def _lcm_policy(self, n_step):
        logging.warning("DeprecationWarning: LCMScheduler will not be supported in the future. "
                        "Consider using `prs-eth/marigold-depth-v1-1` for the best experience.")
        if n_step > 10:
            logging.warning(f"Setting too many denoising steps ({n_step}) may degrade the prediction; "
                            f"consider relying on the default values.")

    def _decode(self, latent):
        return self.decode_depth(latent)

    def decode_depth(self, depth_latent: torch.Tensor) -> torch.Tensor:
        """latent -> depth in [0,1], [B,1,H,W] (decode, channel mean, clip, shift fused on device;
        reference :498-516 + :473-475)."""
        return self.vae.decode(depth_latent, post=L.POST_DEPTH)

    @torch.no_grad()
    def __call__(self, input_image: Union[Image.Image, torch.Tensor], denoising_steps: Optional[int] = None,
                 ensemble_size: int = 1, processing_res: Optional[int] = None, match_input_res: bool = True,
                 resample_method: str = "bilinear", batch_size: int = 0,
                 generator: Union[torch.Generator, None] = None, color_map: str = "Spectral",
                 show_progress_bar: bool = True, ensemble_kwargs: Dict = None,
                 init_latents: Optional[torch.Tensor] = None) -> MarigoldDepthOutput:
        if denoising_steps is None:
            denoising_steps = self.default_denoising_steps
        if processing_res is None:
            processing_res = self.default_processing_resolution
        assert processing_res >= 0
        assert ensemble_size >= 1
        self._check_inference_step(denoising_steps)
        resample = get_tv_resample_method(resample_method)
        rgb_norm, input_size = self._preprocess(input_image, processing_res, resample)

        target_preds = self._predict_members(rgb_norm, ensemble_size, denoising_steps, batch_size,
                                             generator, init_latents)
        if target_preds is None:  # member-parallel non-root rank with a rooted gather
            return MarigoldDepthOutput(depth_np=None, depth_colored=None, uncertainty=None)
        if ensemble_size > 1:
            final_pred, pred_uncert = ensemble_depth(target_preds, scale_invariant=self.scale_invariant,
                                                     shift_invariant=self.shift_invariant,
                                                     **(ensemble_kwargs or {}))
        else:
            final_pred, pred_uncert = target_preds, None
        if match_input_res:
            final_pred = resize(final_pred, input_size[-2:], interpolation=resample, antialias=True)
        colored_dev = None
        if color_map is not None and final_pred.is_cuda:
            # colour table look-up on the device (clip(0, 1) is part of the kernel); the host gets a uint8 HWC image
            colored_dev = colorize_depth_device(final_pred.squeeze().float(), 0.0, 1.0, cmap=color_map)
        final_pred = final_pred.squeeze().cpu().numpy()
        if pred_uncert is not None:
            pred_uncert = pred_uncert.squeeze().cpu().numpy()
        final_pred = final_pred.clip(0, 1)
        if colored_dev is not None:
            depth_colored_img = Image.fromarray(colored_dev.cpu().numpy())
        elif color_map is not None:
            colored = colorize_depth_maps(final_pred, 0, 1, cmap=color_map).squeeze()
            colored = (colored * 255).astype(np.uint8)
            depth_colored_img = Image.fromarray(chw2hwc(colored))
        else:
            depth_colored_img = None
        return MarigoldDepthOutput(depth_np=final_pred, depth_colored=depth_colored_img, uncertainty=pred_uncert)


class MarigoldNormalsPipeline(_MarigoldPipelineBase):
    """Surface normals (reference marigold/marigold_normals_pipeline.py:77-479)."""
    _kind = "normals"
    _ckpt_hint = "prs-eth/marigold-normals-v1-1"
    _pred_channels = 3

    def __init__(self, unet, vae, scheduler, text_encoder=None, tokenizer=None,
                 default_denoising_steps: Optional[int] = None,
                 default_processing_resolution: Optional[int] = None, empty_text_embed=None):
        super().__init__(unet, vae, scheduler, text_encoder, tokenizer, None, None, default_denoising_steps,
                         default_processing_resolution, empty_text_embed)
        self.config = dict(default_denoising_steps=default_denoising_steps,
                           default_processing_resolution=default_processing_resolution)

    def _lcm_policy(self, n_step):
        raise RuntimeError("This pipeline implementation does not support the LCMScheduler. Please refer to "
                           "the project README.md for instructions about using LCM.")

    def _decode(self, latent):
        return self.decode_normals(latent)

    def decode_normals(self, normals_latent: torch.Tensor) -> torch.Tensor:
        """latent -> unit normals [B,3,H,W] (decode, clip, L2 normalise fused; reference :463-479,
        :437-440)."""
        return self.vae.decode(normals_latent, post=L.POST_NORMALS)

    @torch.no_grad()
    def __call__(self, input_image: Union[Image.Image, torch.Tensor], denoising_steps: Optional[int] = None,
                 ensemble_size: int = 1, processing_res: Optional[int] = None, match_input_res: bool = True,
                 resample_method: str = "bilinear", batch_size: int = 0,
                 generator: Union[torch.Generator, None] = None, show_progress_bar: bool = True,
                 ensemble_kwargs: Dict = None,
                 init_latents: Optional[torch.Tensor] = None) -> MarigoldNormalsOutput:
        if denoising_steps is None:
            denoising_steps = self.default_denoising_steps
        if processing_res is None:
            processing_res = self.default_processing_resolution
        assert processing_res >= 0
        assert ensemble_size >= 1
        self._check_inference_step(denoising_steps)
        resample = get_tv_resample_method(resample_method)
        rgb_norm, input_size = self._preprocess(input_image, processing_res, resample)

        target_preds = self._predict_members(rgb_norm, ensemble_size, denoising_steps, batch_size,
                                             generator, init_latents)
        if target_preds is None:
            return MarigoldNormalsOutput(normals_np=None, normals_img=None, uncertainty=None)
        if ensemble_size > 1:
            final_pred, pred_uncert = ensemble_normals(target_preds, **(ensemble_kwargs or {}))
        else:
            final_pred, pred_uncert = target_preds, None
        if match_input_res:
            final_pred = resize(final_pred, input_size[-2:], interpolation=resample, antialias=True)
        final_pred = final_pred.squeeze().cpu().numpy()
        if pred_uncert is not None:
            pred_uncert = pred_uncert.squeeze().cpu().numpy()
        final_pred = final_pred.clip(-1, 1)
        normals_img = ((final_pred + 1) * 127.5).astype(np.uint8)
        normals_img = Image.fromarray(chw2hwc(normals_img))
        return MarigoldNormalsOutput(normals_np=final_pred, normals_img=normals_img, uncertainty=pred_uncert)


# ------------------------------------------------------------------------------------------ IID

@dataclass
class IIDEntry:
    """One decomposed component (reference marigold/marigold_iid_pipeline.py:59-77): ``array`` [3,H,W] in
    [0,1], ``image`` PIL RGB, ``uncertainty`` [3,H,W] | None."""
    name: str
    array: Optional[np.ndarray] = None
    image: Optional[Image.Image] = None
    uncertainty: Optional[np.ndarray] = None


class MarigoldIIDOutput:
    """Named container of the predicted modalities (reference :80-161)."""

    def __init__(self, target_names: List[str]):
        self.n_targets = len(target_names)
        self.target_names = target_names
        self.entries: List[IIDEntry] = [IIDEntry(name=n) for n in target_names]
        self._by_name = {e.name: e for e in self.entries}
        self._filled = set()

    def fill_entry(self, name: str, prediction: torch.Tensor, uncertainty: Optional[torch.Tensor] = None,
                   target_properties: Optional[Dict[str, Any]] = None) -> None:
        if name not in self._by_name:
            raise KeyError(f"Unknown entry name: {name}")
        if name in self._filled:
            raise RuntimeError(f"Entry {name} already filled")
        array = prediction.squeeze().cpu().numpy()
        vis = array
        space = target_properties[name].get("prediction_space", "srgb")
        if space == "linear":   # linear radiometric space -> display gamma, optionally normalised to its maximum
            if target_properties[name].get("up_to_scale", False):
                vis = vis / max(vis.max(), 1e-6)
            vis = vis ** (1 / 2.2)
        entry = self._by_name[name]
        entry.array = array
        entry.image = Image.fromarray(chw2hwc((vis * 255).astype(np.uint8)))
        entry.uncertainty = None if uncertainty is None else uncertainty.squeeze().cpu().numpy()
        self._filled.add(name)

    @property
    def is_complete(self) -> bool:
        return len(self._filled) == self.n_targets

    def __getitem__(self, key: str) -> IIDEntry:
        return self._by_name[key]

    def __iter__(self):
        return iter(self.entries)


class MarigoldIIDPipeline(_MarigoldPipelineBase):
    """Intrinsic image decomposition (reference marigold/marigold_iid_pipeline.py:164-585): the UNet
    predicts 4 latent channels per modality in ``target_properties["target_names"]``
    (8 + ... input channels = image latent + all modality latents), every modality is decoded by the
    VAE separately (here: as extra batch entries of ONE decode program) and mapped to [0,1]."""
    _kind = "iid"
    _ckpt_hint = "prs-eth/marigold-iid-appearance-v1-1` or `prs-eth/marigold-iid-lighting-v1-1"

    def __init__(self, unet, vae, scheduler, text_encoder=None, tokenizer=None,
                 target_properties: Optional[Dict[str, Any]] = None, default_denoising_steps: Optional[int] = None,
                 default_processing_resolution: Optional[int] = None, empty_text_embed=None):
        super().__init__(unet, vae, scheduler, text_encoder, tokenizer, None, None, default_denoising_steps,
                         default_processing_resolution, empty_text_embed)
        self.target_properties = target_properties
        self.target_names = target_properties["target_names"]
        self.n_targets = len(self.target_names)
        self._target_latent_channels = 4 * self.n_targets
        self._pred_channels = 3 * self.n_targets
        if unet.config.out_channels != self._target_latent_channels or \
                unet.config.in_channels != 4 + self._target_latent_channels:
            raise ValueError(f"UNet with {unet.config.in_channels}->{unet.config.out_channels} channels does not "
                             f"match {self.n_targets} target(s) {self.target_names}")
        self.config = dict(target_properties=target_properties, default_denoising_steps=default_denoising_steps,
                           default_processing_resolution=default_processing_resolution)

    def _lcm_policy(self, n_step):
        raise RuntimeError("This pipeline implementation does not support the LCMScheduler. Please refer to the "
                           "project README.md for instructions about using LCM.")

    def _decode(self, latent):
        return self.decode_targets(latent)

    def decode_targets(self, target_latent: torch.Tensor) -> torch.Tensor:
        """[B,4n,h,w] -> [B,3n,H,W] in [0,1]: every modality through post_quant_conv + decoder
        (reference :556-585) with the clip / shift of :523-526 fused; modalities ride in the batch."""
        B, _, h, w = target_latent.shape
        dec = self.vae.decode(target_latent.reshape(B * self.n_targets, 4, h, w), post=L.POST_UNIT)
        return dec.reshape(B, 3 * self.n_targets, dec.shape[-2], dec.shape[-1])

    def fill_outputs(self, output: MarigoldIIDOutput, final_pred: torch.Tensor,
                     pred_uncert: Optional[torch.Tensor] = None):
        for i, name in enumerate(self.target_names):
            output.fill_entry(name=name, prediction=final_pred[:, 3 * i:3 * i + 3],
                              uncertainty=None if pred_uncert is None else pred_uncert[:, 3 * i:3 * i + 3],
                              target_properties=self.target_properties)

    @torch.no_grad()
    def __call__(self, input_image: Union[Image.Image, torch.Tensor], denoising_steps: Optional[int] = None,
                 ensemble_size: int = 1, processing_res: Optional[int] = None, match_input_res: bool = True,
                 resample_method: str = "bilinear", batch_size: int = 0,
                 generator: Union[torch.Generator, None] = None, show_progress_bar: bool = True,
                 ensemble_kwargs: Dict = None, init_latents: Optional[torch.Tensor] = None) -> MarigoldIIDOutput:
        if denoising_steps is None:
            denoising_steps = self.default_denoising_steps
        if processing_res is None:
            processing_res = self.default_processing_resolution
        assert processing_res >= 0
        assert ensemble_size >= 1
        self._check_inference_step(denoising_steps)
        resample = get_tv_resample_method(resample_method)
        rgb_norm, input_size = self._preprocess(input_image, processing_res, resample)
        target_preds = self._predict_members(rgb_norm, ensemble_size, denoising_steps, batch_size, generator,
                                             init_latents)
        output = MarigoldIIDOutput(target_names=self.target_names)
        if target_preds is None:   # member-parallel non-root rank with a rooted gather
            return output
        assert target_preds.dim() == 4 and target_preds.shape[1] == 3 * self.n_targets
        if ensemble_size > 1:
            final_pred, pred_uncert = ensemble_iid(target_preds, **(ensemble_kwargs or {}))
        else:
            final_pred, pred_uncert = target_preds, None
        if match_input_res:
            final_pred = resize(final_pred, input_size[-2:], interpolation=resample, antialias=True)
        self.fill_outputs(output, final_pred, pred_uncert)
        assert output.is_complete
        return output
