"""Multi-image inference CLI with the reference's flags and on-disk formats
(script/depth/run.py:54-135 flags, :165-171 output folders, :270-292 files; script/normals/run.py is
the same minus ``--color_map``):

  depth   : <out>/depth_npy/<name>_depth.npy      float32 [H,W] in [0,1]
            <out>/depth_bw/<name>_depth.png        16-bit PNG, value * 65535
            <out>/depth_colored/<name>_depth_colored.png
  normals : <out>/normals_npy/<name>_normals.npy  float32 [3,H,W] in [-1,1]
            <out>/normals_vis/<name>_normals.png   (n + 1) * 127.5
  iid     : <out>/iid_{appearance|lighting}_npy/<name>_<target>.npy  float32 [H,W,3] in [0,1]
            <out>/iid_{appearance|lighting}_vis/<name>_<target>.png  (script/iid/run.py:160-166, :260-270;
            the folder flavour is picked from the checkpoint name like the reference does)

Differences: the engine only runs on an MI355X (there is no CPU / MPS path: ``--apple_silicon`` is
accepted and refused); ``--half_precision`` / ``--fp16`` selects the ``fp16`` weight variant of the checkpoint AND the
fp16-operand build of the engine (fp32 accumulation), like the reference; without it the engine computes on bf16 operands.
"""
import argparse
import logging
import os
from glob import glob

import numpy as np
from PIL import Image

EXTENSION_LIST = (".jpg", ".jpeg", ".png")
_DEFAULT_CKPT = {"depth": "prs-eth/marigold-depth-v1-1", "normals": "prs-eth/marigold-normals-v1-1",
                 "iid": "prs-eth/marigold-iid-appearance-v1-1"}
_TITLE = {"depth": "Marigold : Monocular Depth Estimation : Multi-image Inference",
          "normals": "Marigold : Surface Normals Estimation : Multi-image Inference",
          "iid": "Marigold : Intrinsic Image Decomposition : Multi-image Inference"}


def build_parser(kind: str) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=_TITLE[kind])
    p.add_argument("--checkpoint", type=str, default=_DEFAULT_CKPT[kind], help="Checkpoint path or hub name.")
    p.add_argument("--input_rgb_dir", type=str, required=True, help="Path to the input image folder.")
    p.add_argument("--output_dir", type=str, required=True, help="Output directory.")
    p.add_argument("--denoise_steps", type=int, default=None,
                   help="Diffusion denoising steps; `None` reads the default from the checkpoint.")
    p.add_argument("--processing_res", type=int, default=None,
                   help="Resolution the input is resized to before estimation; 0 = native, None = checkpoint default.")
    p.add_argument("--ensemble_size", type=int, default=1, help="Number of predictions to be ensembled.")
    p.add_argument("--half_precision", "--fp16", action="store_true",
                   help="Load the 16-bit weight variant of the checkpoint.")
    p.add_argument("--output_processing_res", action="store_true",
                   help="Output at the processing resolution instead of resizing back to the input resolution.")
    p.add_argument("--resample_method", choices=["bilinear", "bicubic", "nearest"], default="bilinear")
    if kind == "depth":
        p.add_argument("--color_map", type=str, default="Spectral", help="Colormap of the depth visualisation.")
    p.add_argument("--seed", type=int, default=None, help="Reproducibility seed; None = randomised inference.")
    p.add_argument("--batch_size", type=int, default=0, help="Inference batch size; 0 = automatic.")
    p.add_argument("--apple_silicon", action="store_true", help="(reference flag; not available on this engine)")
    p.add_argument("--maps_in_flight", type=int, default=0,
                   help="Images on the GPU at a time (independent maps on concurrent HIP streams; results do not depend on it); "
                        "0 = the engine's default (2).")
    return p


def list_images(folder):
    files = sorted(f for f in glob(os.path.join(folder, "*")) if os.path.splitext(f)[1].lower() in EXTENSION_LIST)
    return files


def output_dirs(kind, output_dir, checkpoint=""):
    if kind == "iid":
        flavour = "appearance" if "appearance" in checkpoint else "lighting"
        dirs = {"iid_vis": os.path.join(output_dir, f"iid_{flavour}_vis"),
                "iid_npy": os.path.join(output_dir, f"iid_{flavour}_npy")}
    else:
        sub = ("depth_colored", "depth_bw", "depth_npy") if kind == "depth" else ("normals_vis", "normals_npy")
        dirs = {s: os.path.join(output_dir, s) for s in sub}
    for d in (output_dir, *dirs.values()):
        os.makedirs(d, exist_ok=True)
    return dirs


def _save(path, writer):
    if os.path.exists(path):
        logging.warning(f"Existing file: '{path}' will be overwritten")
    writer(path)


def save_prediction(kind, dirs, rgb_path, out):
    """Write one prediction in the reference's formats; returns the written paths."""
    base = os.path.splitext(os.path.basename(rgb_path))[0]
    written = []
    if kind == "depth":
        name = base + "_depth"
        depth = out.depth_np
        p = os.path.join(dirs["depth_npy"], f"{name}.npy")
        _save(p, lambda q: np.save(q, depth))
        written.append(p)
        p = os.path.join(dirs["depth_bw"], f"{name}.png")
        _save(p, lambda q: Image.fromarray((depth * 65535.0).astype(np.uint16)).save(q, mode="I;16"))
        written.append(p)
        if out.depth_colored is not None:
            p = os.path.join(dirs["depth_colored"], f"{name}_colored.png")
            _save(p, out.depth_colored.save)
            written.append(p)
    elif kind == "iid":
        for entry in out:
            arr = np.moveaxis(entry.array, 0, -1)   # chw2hwc
            p = os.path.join(dirs["iid_npy"], f"{base}_{entry.name}.npy")
            _save(p, lambda q: np.save(q, arr))
            written.append(p)
            p = os.path.join(dirs["iid_vis"], f"{base}_{entry.name}.png")
            _save(p, entry.image.save)
            written.append(p)
    else:
        name = base + "_normals"
        p = os.path.join(dirs["normals_npy"], f"{name}.npy")
        _save(p, lambda q: np.save(q, out.normals_np))
        written.append(p)
        p = os.path.join(dirs["normals_vis"], f"{name}.png")
        _save(p, out.normals_img.save)
        written.append(p)
    return written


def main(kind: str, argv=None, pipeline=None) -> int:
    """``pipeline`` lets tests inject a ready pipeline object; otherwise the checkpoint is loaded."""
    import torch
    logging.basicConfig(level=logging.INFO)
    args = build_parser(kind).parse_args(argv)
    if args.ensemble_size > 15:
        logging.warning("Running with large ensemble size will be slow.")
    match_input_res = not args.output_processing_res
    if 0 == args.processing_res and match_input_res is False:
        logging.warning("Processing at native resolution without resizing output might NOT lead to exactly the "
                        "same resolution, due to the padding and pooling properties of conv layers.")
    if args.apple_silicon:
        raise RuntimeError("--apple_silicon: this engine runs on an AMD MI355X only")
    dirs = output_dirs(kind, args.output_dir, args.checkpoint)
    logging.info(f"output dir = {args.output_dir}")
    files = list_images(args.input_rgb_dir)
    if not files:
        logging.error(f"No image found in '{args.input_rgb_dir}'")
        return 1
    logging.info(f"Found {len(files)} images")
    if pipeline is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the Marigold HIP engine has no CPU fallback")
        import marigold_amd as M
        cls = {"depth": M.MarigoldDepthPipeline, "normals": M.MarigoldNormalsPipeline,
               "iid": M.MarigoldIIDPipeline}[kind]
        pipeline = cls.from_pretrained(args.checkpoint, variant="fp16" if args.half_precision else None,
                                       torch_dtype=torch.float16 if args.half_precision else torch.float32)
        pipeline.enable_xformers_memory_efficient_attention()   # no-op: attention is always the fused kernel
        pipeline = pipeline.to("cuda")
    if kind == "depth":
        logging.info(f"Loaded depth pipeline: scale_invariant={pipeline.scale_invariant}, "
                     f"shift_invariant={pipeline.shift_invariant}")
    elif kind == "iid":
        logging.info(f"Loaded IID pipeline with predicted target names: {pipeline.target_names}")
    else:
        logging.info("Loaded normals pipeline")
    logging.info(f"Inference settings: checkpoint = `{args.checkpoint}`, with denoise_steps = "
                 f"{args.denoise_steps or pipeline.default_denoising_steps}, ensemble_size = {args.ensemble_size}, "
                 f"processing resolution = {args.processing_res or pipeline.default_processing_resolution}, "
                 f"seed = {args.seed}" + (f"; color_map = {args.color_map}." if kind == "depth" else ""))
    device = getattr(pipeline, "device", "cpu")
    kw = dict(denoising_steps=args.denoise_steps, ensemble_size=args.ensemble_size,
              processing_res=args.processing_res, match_input_res=match_input_res,
              batch_size=args.batch_size, show_progress_bar=True, resample_method=args.resample_method)
    if kind == "depth":
        kw["color_map"] = args.color_map

    def generator_of(_path):   # a fresh generator per image, like the reference's loop (script/depth/run.py:240-244)
        if args.seed is None:
            return None
        g = torch.Generator(device=device)
        g.manual_seed(args.seed)
        return g

    if hasattr(pipeline, "map_images"):
        # the engine's multi-image form: up to --maps_in_flight images on the GPU at a time, outputs in input order
        outs = pipeline.map_images((Image.open(f) for f in files), in_flight=args.maps_in_flight or None,
                                   generators=(generator_of(f) for f in files), **kw)
        for rgb_path, out in zip(files, outs):
            save_prediction(kind, dirs, rgb_path, out)
    else:   # an object with the reference pipeline's call surface only
        for rgb_path in files:
            out = pipeline(Image.open(rgb_path), generator=generator_of(rgb_path), **kw)
            save_prediction(kind, dirs, rgb_path, out)
    return 0
