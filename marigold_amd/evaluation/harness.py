"""The two benchmark programs of the reference, per task (depth / normals / iid):

* ``infer_main``  - script/<task>/infer.py: run the pipeline over a dataset split and write one ``.npy``
  prediction per image under ``<output_dir>/<scene dirs of the rgb path>/``;
* ``eval_main``   - script/<task>/eval.py: read those predictions back, align (depth), score against
  the ground truth, write ``per_sample_metrics.csv`` and ``eval_metrics[-<alignment>].txt``.

Flags, file names and the text formats are the reference's.  What differs is the plumbing: samples are
decoded one image ahead on a host thread and predictions are written behind the GPU (the reference's
DataLoader(batch_size=1, num_workers=0) serialises decode -> predict -> save), and the scores are numpy
(see metrics.py).  Inference only runs on an MI355X - there is no CPU path.
"""
import argparse
import logging
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image
from tabulate import tabulate

from . import metrics as M
from .alignment import align_depth_least_square, depth2disparity, disparity2depth
from .datasets import DatasetMode, get_dataset, get_pred_name, load_dataset_config

_DEFAULT_CKPT = {"depth": "prs-eth/marigold-depth-v1-1", "normals": "prs-eth/marigold-normals-v1-1",
                 "iid": "prs-eth/marigold-iid-appearance-v1-1"}
_TASK = {"depth": "Monocular Depth Estimation", "normals": "Surface Normals Estimation",
         "iid": "Intrinsic Image Decomposition"}


def seed_all(seed=0):
    """src/util/seeding.py:31-39 (the engine itself draws from the per-image generator only)."""
    import random
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ---- inference over a dataset ------------------------------------------------------------------------


def infer_parser(kind):
    p = argparse.ArgumentParser(description=f"Marigold : {_TASK[kind]} : Dataset Inference")
    p.add_argument("--checkpoint", type=str, default=_DEFAULT_CKPT[kind], help="Checkpoint path or hub name.")
    p.add_argument("--dataset_config", type=str, required=True, help="Path to the config file of the evaluation dataset.")
    p.add_argument("--base_data_dir", type=str, required=True, help="Base path to the datasets.")
    p.add_argument("--output_dir", type=str, required=True, help="Output directory.")
    p.add_argument("--denoise_steps", type=int, required=True, help="Diffusion denoising steps.")
    p.add_argument("--processing_res", type=int, required=True,
                   help="Resolution the input is resized to before estimation; 0 = native.")
    p.add_argument("--ensemble_size", type=int, required=True, help="Number of predictions to be ensembled.")
    p.add_argument("--half_precision", "--fp16", action="store_true", help="Load the 16-bit weight variant.")
    p.add_argument("--output_processing_res", action="store_true",
                   help="Output at the processing resolution instead of resizing back to the input resolution.")
    p.add_argument("--resample_method", choices=["bilinear", "bicubic", "nearest"], default="bilinear")
    p.add_argument("--seed", type=int, default=None, help="Reproducibility seed; None = time-seeded.")
    p.add_argument("--yes", action="store_true", help="Do not ask before writing into an existing output dir.")
    p.add_argument("--maps_in_flight", type=int, default=0,
                   help="Images on the GPU at a time (independent maps on concurrent HIP streams; results do not depend on it); "
                        "0 = the engine's default (2).")
    return p


def _confirm_existing(directory, assume_yes):
    """script/depth/infer.py:165-183: ask before re-using an output folder."""
    while os.path.exists(directory) and not assume_yes:
        answer = input(f"The directory '{directory}' already exists. Are you sure to continue? (y/n): ").strip().lower()
        if answer == "y":
            return True
        if answer == "n":
            print("Exiting...")
            return False
        print("Invalid input. Please enter 'y' (for Yes) or 'n' (for No).")
    return True


def _pipeline_input(kind, sample):
    if kind == "iid":   # float [0,1] -> uint8 by truncation (marigold/util/image_util.py:137-141)
        return (sample["rgb"] * 255.0).astype(np.uint8)
    return sample["rgb_int"].astype(np.uint8)


def _prediction_files(kind, dataset, pipeline, rgb_rel, out):
    """[(relative file name, array)] of one pipeline output (depth infer.py:268-281, normals infer.py:258-270,
    iid infer.py:270-287)."""
    folder, base = os.path.dirname(rgb_rel), os.path.basename(rgb_rel)
    stem = os.path.splitext(base)[0]
    if kind == "depth":
        return [(os.path.join(folder, get_pred_name(base, dataset.name_mode, suffix=".npy")), out.depth_np)]
    if kind == "normals":
        return [(os.path.join(folder, stem + ".npy"), out.normals_np)]
    return [(os.path.join(folder, f"{stem}_{t}.npy"), out[t].array) for t in pipeline.target_names]


def _save_npy(path, arr):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if os.path.exists(path):
        logging.warning(f"Existing file: '{path}' will be overwritten")
    np.save(path, arr)


def infer_main(kind, argv=None, pipeline=None) -> int:
    """``pipeline`` lets tests inject a ready pipeline object; otherwise the checkpoint is loaded onto the GPU."""
    import torch
    logging.basicConfig(level=logging.INFO)
    args = infer_parser(kind).parse_args(argv)
    if args.ensemble_size > 15:
        logging.warning("Running with large ensemble size will be slow.")
    match_input_res = not args.output_processing_res
    if 0 == args.processing_res and match_input_res is False:
        logging.warning("Processing at native resolution without resizing output might NOT lead to exactly the "
                        "same resolution, due to the padding and pooling properties of conv layers.")
    logging.info(f"Inference settings: checkpoint = `{args.checkpoint}`, with denoise_steps = {args.denoise_steps}, "
                 f"ensemble_size = {args.ensemble_size}, processing resolution = {args.processing_res}, "
                 f"seed = {args.seed}; dataset config = `{args.dataset_config}`.")
    seed = int(time.time()) if args.seed is None else args.seed
    seed_all(seed)
    if not _confirm_existing(args.output_dir, args.yes):
        return 0
    os.makedirs(args.output_dir, exist_ok=True)
    logging.info(f"output dir = {args.output_dir}")
    dataset = get_dataset(load_dataset_config(args.dataset_config), args.base_data_dir, DatasetMode.RGB_ONLY)
    if dataset.spec.kind != kind:
        raise AssertionError(f"'{dataset.name}' is a {dataset.spec.kind} dataset, not {kind}")
    if pipeline is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the Marigold HIP engine has no CPU fallback")
        import marigold_amd as MA
        cls = {"depth": MA.MarigoldDepthPipeline, "normals": MA.MarigoldNormalsPipeline,
               "iid": MA.MarigoldIIDPipeline}[kind]
        pipeline = cls.from_pretrained(args.checkpoint, variant="fp16" if args.half_precision else None,
                                       torch_dtype=torch.float16 if args.half_precision else torch.float32)
        pipeline = pipeline.to("cuda")
    device = getattr(pipeline, "device", "cpu")
    n = len(dataset)
    t0 = time.perf_counter()
    kw = dict(denoising_steps=args.denoise_steps, ensemble_size=args.ensemble_size,
              processing_res=args.processing_res, match_input_res=match_input_res, batch_size=0,
              show_progress_bar=False, resample_method=args.resample_method)
    if kind == "depth":
        kw["color_map"] = None

    def generator_of():   # a fresh generator per image, seeded alike (script/depth/infer.py:186-190)
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        return g
    with ThreadPoolExecutor(max_workers=1) as reader, ThreadPoolExecutor(max_workers=2) as writer:
        samples = []   # in input order; the engine asks for the next image when a lane is free, results come back in order

        def images():
            pending = reader.submit(dataset.__getitem__, 0) if n else None
            for i in range(n):
                sample = pending.result()
                pending = reader.submit(dataset.__getitem__, i + 1) if i + 1 < n else None
                samples.append(sample)
                yield Image.fromarray(np.moveaxis(_pipeline_input(kind, sample), 0, -1))
        if hasattr(pipeline, "map_images"):   # the engine: up to --maps_in_flight images on the GPU at a time
            outs = pipeline.map_images(images(), in_flight=args.maps_in_flight or None, generators=(generator_of() for _ in range(n)), **kw)
        else:                                 # an object with the reference pipeline's call surface only
            outs = (pipeline(im, generator=generator_of(), **kw) for im in images())
        writes = []
        for i, out in enumerate(outs):
            for rel, arr in _prediction_files(kind, dataset, pipeline, samples[i]["rgb_relative_path"], out):
                writes.append(writer.submit(_save_npy, os.path.join(args.output_dir, rel), arr))
            samples[i] = None
        for w in writes:
            w.result()
    dt = time.perf_counter() - t0
    logging.info(f"{_TASK[kind]} inference on {dataset.disp_name}: {n} images in {dt:.1f} s "
                 f"({n / max(dt, 1e-9):.2f} img/s)")
    return 0


# ---- evaluation ------------------------------------------------------------------------------------------


def eval_parser(kind):
    p = argparse.ArgumentParser(description=f"Marigold : {_TASK[kind]} : Metrics Evaluation")
    p.add_argument("--prediction_dir", type=str, required=True, help="Directory with predictions obtained from inference.")
    p.add_argument("--dataset_config", type=str, required=True, help="Path to the config file of the evaluation dataset.")
    p.add_argument("--base_data_dir", type=str, required=True, help="Base path to the datasets.")
    p.add_argument("--output_dir", type=str, required=True, help="Output directory.")
    if kind == "depth":
        p.add_argument("--alignment", choices=[None, "least_square", "least_square_disparity"], default=None,
                       help="Method to estimate scale and shift between predictions and ground truth.")
        p.add_argument("--alignment_max_res", type=int, default=None, help="Max operating resolution used for LS alignment")
    else:
        p.add_argument("--use_mask", action="store_true", help="Evaluate only in the masked region.")
    if kind == "iid":
        p.add_argument("--target_names", nargs="+", default=["albedo", "material"], type=str,
                       help="A list of predicted targets to evaluate.")
        p.add_argument("--targets_to_eval_in_linear_space", nargs="*", default=[None], type=str,
                       help="Targets to evaluate in linear space (as opposed to sRGB by default).")
        p.add_argument("--metrics", nargs="+", default=["psnr", "ssim"], choices=["psnr", "ssim"],
                       help="(LPIPS of the reference needs pretrained network weights; not provided)")
    p.add_argument("--no_cuda", action="store_true", help="(reference flag; scoring runs on the host here)")
    return p


def align_and_clip_depth(depth_pred, depth_raw, valid_mask, dataset, alignment=None, alignment_max_res=None):
    """The per-sample preparation of script/depth/eval.py:176-212."""
    if alignment == "least_square":
        depth_pred, _, _ = align_depth_least_square(depth_raw, depth_pred, valid_mask, True, alignment_max_res)
    elif alignment == "least_square_disparity":
        gt_disp, gt_pos = depth2disparity(depth_raw, return_mask=True)
        ok = valid_mask & gt_pos & (depth_pred > 0)
        disp, _, _ = align_depth_least_square(gt_disp, depth_pred, ok, True, alignment_max_res)
        depth_pred = disparity2depth(np.clip(disp, a_min=1e-3, a_max=None))   # avoid 0 disparity
    depth_pred = np.clip(depth_pred, a_min=dataset.min_depth, a_max=dataset.max_depth)
    return np.clip(depth_pred, a_min=1e-6, a_max=None)


def _score_depth(args, dataset, data, names):
    rgb_name = data["rgb_relative_path"]
    pred_name = os.path.join(os.path.dirname(rgb_name),
                             get_pred_name(os.path.basename(rgb_name), dataset.name_mode, suffix=".npy"))
    path = os.path.join(args.prediction_dir, pred_name)
    if not os.path.exists(path):
        logging.warning(f"Can't find prediction: {path}")
        return None
    gt, valid = data["depth_raw_linear"].squeeze(), data["valid_mask_raw"].squeeze()
    pred = align_and_clip_depth(np.load(path).astype(np.float32), gt, valid, dataset, args.alignment,
                                args.alignment_max_res)
    return pred_name, [getattr(M, n)(pred, gt, valid) for n in names]


def _score_normals(args, dataset, data, names):
    rgb_name = data["rgb_relative_path"]
    path = os.path.join(args.prediction_dir, os.path.splitext(rgb_name)[0] + ".npy")
    if not os.path.exists(path):
        logging.warning(f"Can't find prediction: {path}")
        return None
    err = M.compute_cosine_error(np.load(path).astype(np.float32), data["normals"], masked=True)
    return rgb_name, [getattr(M, n)(err) for n in names]


def _score_iid(args, dataset, data, names):
    rgb_name = data["rgb_relative_path"]
    stem = os.path.join(args.prediction_dir, os.path.splitext(rgb_name)[0])
    values = []
    for target in args.target_names:
        path = f"{stem}_{target}.npy"
        if not os.path.exists(path):
            # keep the columns aligned with `names` (metric-major per target): a missing target leaves empty cells
            # and is not counted in the averages (the reference updates its tracker by metric name too)
            logging.warning(f"Can't find prediction: {path}")
            values += [None] * len(args.metrics)
            continue
        pred, gt = np.load(path)[None].astype(np.float32), data[target][None].astype(np.float32)
        if target in args.targets_to_eval_in_linear_space:
            pred, gt = pred ** 2.2, gt ** 2.2
        if "hypersim" in dataset.name and len(args.target_names) == 3 and target == "albedo":
            pred, gt = pred ** (1.0 / 2.2), gt ** (1.0 / 2.2)
        mask = data["mask_" + target] if args.use_mask else None
        values += [M.compute_iid_metric(pred.copy(), gt.copy(), target, m, mask) for m in args.metrics]
    return rgb_name, values


def eval_main(kind, argv=None) -> int:
    logging.basicConfig(level=logging.INFO)
    args = eval_parser(kind).parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    dataset = get_dataset(load_dataset_config(args.dataset_config), args.base_data_dir, DatasetMode.EVAL)
    if kind == "depth":
        names, score = list(M.DEPTH_METRICS), _score_depth
    elif kind == "normals":
        names, score = list(M.NORMALS_METRICS), _score_normals
    else:
        for t in args.targets_to_eval_in_linear_space:
            if t is not None and t not in args.target_names:
                raise ValueError(f"'{t}' specified in targets_to_eval_in_linear_space does not belong to the "
                                 f"predicted targets: target_names={args.target_names}")
        names, score = [f"{m}_{t}" for t in args.target_names for m in args.metrics], _score_iid
    tracker = M.MetricTracker(*names)
    per_sample = os.path.join(args.output_dir, "per_sample_metrics.csv")
    with open(per_sample, "w+") as f:
        f.write("filename," + ",".join(names) + "\n")
        for data in dataset:
            scored = score(args, dataset, data, names)
            if scored is None:
                continue
            label, values = scored
            assert len(values) == len(names)
            for n, v in zip(names, values):
                if v is not None:
                    tracker.update(n, v)
            f.write(label + "," + ",".join("" if v is None else str(v) for v in values) + "\n")
    text = (f"Evaluation metrics:\n    of predictions: {args.prediction_dir}\n    on dataset: {dataset.disp_name}\n"
            f"    with samples in: {dataset.filename_ls_path}\n")
    if kind == "depth":
        text += f"min_depth = {dataset.min_depth}\nmax_depth = {dataset.max_depth}\n"
    result = tracker.result()
    text += tabulate([list(result.keys()), list(result.values())])
    name = "eval_metrics" + (f"-{args.alignment}" if kind == "depth" and args.alignment else "") + ".txt"
    with open(os.path.join(args.output_dir, name), "w+") as f:
        f.write(text)
    logging.info(f"Evaluation metrics saved to {os.path.join(args.output_dir, name)}")
    return 0
