#!/usr/bin/env python
"""TEST INFRASTRUCTURE - generates tests/golden/eval_ref.npz by running the REFERENCE's own evaluation
code (imported unmodified from /root/reference: src/util/metric.py, src/util/alignment.py and the dataset
readers in src/dataset/) on seeded synthetic inputs.  /root/reference only exists in the build container,
so the vectors are committed; tests/test_evaluation.py rebuilds the same inputs from ``eval_inputs`` /
``write_synthetic_datasets`` below and compares marigold_amd.evaluation against them.

torchvision, cv2 and omegaconf are not installed here: the few symbols the reference's dataset modules
import at module level (never called on the inference / evaluation path) are stubbed.

    python oracle/make_eval_golden.py
"""
import io
import os
import sys
import tarfile
import tempfile
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "eval_ref.npz")


# ---- seeded inputs shared with the tests -----------------------------------------------------------

def eval_inputs():
    """Deterministic depth / normals / iid prediction + ground-truth pairs."""
    r = np.random.default_rng(20240607)
    cases = {}
    for tag, (h, w) in (("a", (48, 64)), ("b", (37, 53)), ("big", (120, 200))):
        gt = (r.uniform(0.5, 9.5, (h, w)) * (1 + 0.3 * np.sin(np.arange(w) / 7.0))).astype(np.float32)
        pred = (gt * r.uniform(0.8, 1.25, (h, w)) + r.normal(0, 0.05, (h, w))).astype(np.float32)
        pred = np.clip(pred, 1e-3, None)
        mask = r.uniform(size=(h, w)) > 0.2
        gt_holes = gt.copy()
        gt_holes[~mask] = 0.0
        rel = ((gt.max() - gt) / (gt.max() - gt.min()) * 0.9 + 0.05 + r.normal(0, 0.01, (h, w))).astype(np.float32)
        cases[f"depth_{tag}"] = dict(gt=gt, gt_holes=gt_holes, pred=pred, mask=mask, rel=rel)
    n_gt = r.normal(size=(3, 40, 56)).astype(np.float32)
    n_gt /= np.linalg.norm(n_gt, axis=0, keepdims=True)
    n_gt[:, :5, :7] = 0
    n_pred = n_gt + r.normal(0, 0.25, n_gt.shape).astype(np.float32)
    n_pred /= np.maximum(np.linalg.norm(n_pred, axis=0, keepdims=True), 1e-6)
    cases["normals"] = dict(gt=n_gt, pred=n_pred.astype(np.float32))
    i_gt = r.uniform(0, 0.7, (3, 32, 40)).astype(np.float32)
    i_pred = np.clip(i_gt * 0.6 + r.normal(0, 0.03, i_gt.shape), 0, 1).astype(np.float32)
    i_mask = np.broadcast_to(r.uniform(size=(1, 32, 40)) > 0.15, i_gt.shape).copy()
    cases["iid"] = dict(gt=i_gt, pred=i_pred, mask=i_mask)
    return cases


def _png16(a):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(a.astype(np.uint16)).save(b, format="PNG")
    return b.getvalue()


def _png8(a):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(a.astype(np.uint8)).save(b, format="PNG")
    return b.getvalue()


def _npy(a):
    b = io.BytesIO()
    np.save(b, a)
    return b.getvalue()


def synthetic_dataset_files():
    """{dataset key: (config dict, split lines, {relative path: bytes})} - small stand-ins in every on-disk
    format the reference's readers decode."""
    r = np.random.default_rng(77)
    out = {}

    def rgb(h, w):
        return r.integers(0, 256, (h, w, 3))

    # NYU: 16-bit PNG millimetres, raw + filled, Eigen window needs 480x640
    files, lines = {}, []
    for i in (3, 12):
        d = r.integers(0, 11000, (480, 640))
        files[f"test/kitchen/rgb_{i:04d}.png"] = _png8(rgb(480, 640))
        files[f"test/kitchen/depth_{i:04d}.png"] = _png16(d)
        files[f"test/kitchen/filled_{i:04d}.png"] = _png16(np.maximum(d, 400))
        lines.append(f"test/kitchen/rgb_{i:04d}.png test/kitchen/depth_{i:04d}.png test/kitchen/filled_{i:04d}.png")
    out["nyu"] = (dict(name="nyu_depth", disp_name="nyu_synth", dir="nyu", eigen_valid_mask=True), lines, files)
    # KITTI: 16-bit PNG /256, benchmark crop + eigen window, a row without ground truth
    files, lines = {}, []
    for i in range(2):
        files[f"2011/image_02/{i:010d}.png"] = _png8(rgb(375, 1242))
        files[f"2011/gt/{i:010d}.png"] = _png16(r.integers(0, 22000, (375, 1242)) * (r.uniform(size=(375, 1242)) > 0.6))
        lines.append(f"2011/image_02/{i:010d}.png 2011/gt/{i:010d}.png 721.5")
    lines.insert(1, "2011/image_02/0000000009.png None 721.5")
    out["kitti"] = (dict(name="kitti_depth", disp_name="kitti_synth", dir="kitti", kitti_bm_crop=True,
                         valid_mask_crop="eigen"), lines, files)
    out["kitti_garg"] = (dict(name="kitti_depth", disp_name="kitti_synth_garg", dir="kitti", kitti_bm_crop=False,
                              valid_mask_crop="garg"), lines, files)
    # DIODE: float .npy [H,W,1] + mask .npy
    files, lines = {}, []
    d = r.uniform(0, 400, (60, 80, 1)).astype(np.float32)
    files["indoors/s0/a.png"], files["indoors/s0/a_depth.npy"] = _png8(rgb(60, 80)), _npy(d)
    files["indoors/s0/a_depth_mask.npy"] = _npy((r.uniform(size=(60, 80)) > 0.3).astype(np.float32))
    lines.append("indoors/s0/a.png indoors/s0/a_depth.npy indoors/s0/a_depth_mask.npy")
    out["diode"] = (dict(name="diode_depth", disp_name="diode_synth", dir="diode"), lines, files)
    # ScanNet / Hypersim: 16-bit PNG millimetres
    files = {"s/color_1.png": _png8(rgb(30, 40)), "s/depth_1.png": _png16(r.integers(0, 12000, (30, 40)))}
    out["scannet"] = (dict(name="scannet_depth", disp_name="scannet_synth", dir="scannet"),
                      ["s/color_1.png s/depth_1.png"], files)
    files = {"ai/rgb_cam_00_fr0001.png": _png8(rgb(30, 40)),
             "ai/depth_plane_cam_00_fr0001.png": _png16(r.integers(0, 65535, (30, 40)))}
    out["hypersim"] = (dict(name="hypersim_depth", disp_name="hypersim_synth", dir="hypersim"),
                       ["ai/rgb_cam_00_fr0001.png ai/depth_plane_cam_00_fr0001.png"], files)
    # normals: [H,W,3] .npy
    n = r.normal(size=(30, 40, 3)).astype(np.float32)
    files = {"x/img.png": _png8(rgb(30, 40)), "x/normal.npy": _npy(n)}
    out["nyu_normals"] = (dict(name="nyu_normals", disp_name="nyu_normals_synth", dir="nyun"),
                          ["x/img.png x/normal.npy"], files)
    n = r.normal(size=(436, 1024, 3)).astype(np.float32) * (r.uniform(size=(436, 1024, 1)) > 0.3)
    files = {"alley/frame_0001.png": _png8(rgb(436, 1024)), "alley/normal_0001.npy": _npy(n.astype(np.float32))}
    out["sintel"] = (dict(name="sintel_normals", disp_name="sintel_synth", dir="sintel"),
                     ["alley/frame_0001.png alley/normal_0001.npy"], files)
    # Hypersim IID: linear-space [H,W,3] .npy rasters
    alb = r.uniform(0, 1, (24, 32, 3)).astype(np.float32)
    alb[:3, :4] = 0
    sh = r.gamma(2.0, 0.5, (24, 32, 3)).astype(np.float32)
    res = r.gamma(1.0, 0.2, (24, 32, 3)).astype(np.float32)
    files = {"ai/rgb_cam_00_fr0000.png": _png8(rgb(24, 32)), "ai/albedo_cam_00_fr0000.npy": _npy(alb),
             "ai/shading_cam_00_fr0000.npy": _npy(sh), "ai/residual_cam_00_fr0000.npy": _npy(res)}
    out["hypersim_iid"] = (dict(name="hypersim_iid", disp_name="hypersim_iid_synth", dir="hyperiid"),
                           ["ai/rgb_cam_00_fr0000.png ai/albedo_cam_00_fr0000.npy ai/shading_cam_00_fr0000.npy "
                            "ai/residual_cam_00_fr0000.npy"], files)
    return out


def write_synthetic_datasets(base_dir, as_tar=()):
    """Materialise ``synthetic_dataset_files`` under ``base_dir`` (folders, or ``<dir>.tar`` archives with
    ``./``-prefixed members for the keys in ``as_tar``) -> {key: config dict with absolute split path}."""
    cfgs = {}
    for key, (cfg, lines, files) in synthetic_dataset_files().items():
        cfg = dict(cfg)
        split = os.path.join(base_dir, f"split_{key}.txt")
        with open(split, "w") as f:
            f.write("\n".join(lines) + "\n")
        cfg["filenames"] = split
        if key in as_tar:
            cfg["dir"] = cfg["dir"] + ".tar"
            with tarfile.open(os.path.join(base_dir, cfg["dir"]), "w") as t:
                for rel, data in files.items():
                    info = tarfile.TarInfo("./" + rel)
                    info.size = len(data)
                    t.addfile(info, io.BytesIO(data))
        else:
            for rel, data in files.items():
                p = os.path.join(base_dir, cfg["dir"], rel)
                os.makedirs(os.path.dirname(p), exist_ok=True)
                with open(p, "wb") as f:
                    f.write(data)
        cfgs[key] = cfg
    return cfgs


# ---- the reference, imported where it lies -----------------------------------------------------------

def _import_reference():
    import torch
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class _Unused:
        def __init__(self, *a, **k):
            raise RuntimeError("training-only symbol stubbed for golden generation")

    tvt.InterpolationMode = types.SimpleNamespace(NEAREST_EXACT="nearest-exact", BILINEAR="bilinear")
    tvt.Resize = tvt.ColorJitter = _Unused
    tvt.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvt.functional, "cv2": types.ModuleType("cv2")})
    sys.path.insert(0, REF)
    from src.util import alignment, metric   # noqa: E402
    import src.dataset as dataset            # noqa: E402
    return torch, alignment, metric, dataset


def main():
    torch, ralign, rmetric, rdata = _import_reference()
    gold = {}
    depth_names = ["abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10",
                   "delta1_acc", "delta2_acc", "delta3_acc", "i_rmse", "silog_rmse"]
    cases = eval_inputs()
    for key, c in cases.items():
        if not key.startswith("depth_"):
            continue
        gt, pred, mask = (torch.from_numpy(c[k].copy()) for k in ("gt", "pred", "mask"))
        gold[f"{key}/metrics_masked"] = np.array([getattr(rmetric, n)(pred.clone(), gt.clone(), mask).item()
                                                  for n in depth_names], np.float64)
        gold[f"{key}/metrics_nomask"] = np.array(
            [getattr(rmetric, n)(pred.clone(), gt.clone(), None).item() for n in depth_names
             if n not in ("delta1_acc", "delta2_acc", "delta3_acc")], np.float64)   # n.cpu() needs a mask there
        for res in (None, 64):
            a, s, t = ralign.align_depth_least_square(c["gt"], c["rel"], c["mask"], True, res)
            gold[f"{key}/ls_{res}"] = np.array([float(s), float(t)], np.float64)
            gold[f"{key}/ls_{res}_aligned"] = a.astype(np.float32)
        disp, pos = ralign.depth2disparity(c["gt_holes"], return_mask=True)
        gold[f"{key}/disparity"] = disp.astype(np.float32)
        ok = c["mask"] & pos & (c["rel"] > 0)
        a, s, t = ralign.align_depth_least_square(disp, c["rel"], ok, True, None)
        gold[f"{key}/ls_disp"] = np.array([float(s), float(t)], np.float64)
    c = cases["normals"]
    for masked in (False, True):
        err = rmetric.compute_cosine_error(torch.from_numpy(c["pred"])[None], torch.from_numpy(c["gt"])[None], masked)
        gold[f"normals/err_masked{int(masked)}"] = err.astype(np.float32)
        gold[f"normals/metrics_masked{int(masked)}"] = np.array(
            [getattr(rmetric, n)(err) for n in ("mean_angular_error", "median_angular_error", "rmse_angular_error",
                                                "sub5_error", "sub7_5_error", "sub11_25_error", "sub22_5_error",
                                                "sub30_error")], np.float64)
    c = cases["iid"]
    p, g, m = (torch.from_numpy(c[k].copy()) for k in ("pred", "gt", "mask"))
    gold["iid/scale_nomask"] = np.array(float(rmetric.compute_alignment_scale(p, g, None)))
    gold["iid/scale_masked"] = np.array(float(rmetric.compute_alignment_scale(p, g, m)))
    for tag, mm in (("nomask", None), ("masked", m)):
        pm, gm = rmetric.quantile_map(p.clone(), g.clone(), mm)
        gold[f"iid/qmap_pred_{tag}"], gold[f"iid/qmap_gt_{tag}"] = pm.numpy(), gm.numpy()

    def psnr(a, b):   # what torchmetrics' PeakSignalNoiseRatio(data_range=1.0) evaluates
        return 10 * torch.log10(1.0 / torch.mean((a - b) ** 2))

    for target in ("albedo", "shading"):
        for tag, mm in (("nomask", None), ("masked", m)):
            gold[f"iid/psnr_{target}_{tag}"] = np.array(rmetric.compute_iid_metric(
                p.clone(), g.clone(), target, "psnr", psnr, None if mm is None else mm.clone()))
    names = {"rgb_id": "rgb_0012.png", "i_d_rgb": "3_17_rgb.jpg", "id": "0000000005.png",
             "rgb_i_d": "rgb_cam_00_fr0001.png"}
    gold["pred_names"] = np.array([rdata.get_pred_name(v, getattr(rdata.base_depth_dataset.DepthFileNameMode, k),
                                                       suffix=".npy") for k, v in names.items()])
    # dataset readers, folder and tar
    with tempfile.TemporaryDirectory() as tmp:
        cfgs = write_synthetic_datasets(tmp, as_tar=("nyu", "diode", "nyu_normals"))
        for key, cfg in cfgs.items():
            for mode in (rdata.DatasetMode.RGB_ONLY, rdata.DatasetMode.EVAL):
                ocfg = _AttrDict(cfg)
                ds = rdata.get_dataset(ocfg, base_data_dir=tmp, mode=mode)
                gold[f"ds/{key}/{mode.value}/len"] = np.array(len(ds))
                for i in range(len(ds)):
                    item = ds[i]
                    for k, v in item.items():
                        if isinstance(v, torch.Tensor):
                            v = v.numpy()
                            if v.size > 4096:   # keep the fixture small: shape + moments + a strided probe
                                flat = v.reshape(-1).astype(np.float64)
                                v = np.concatenate([[float(x) for x in v.shape], [np.nansum(flat), np.nansum(flat * flat)],
                                                    flat[::max(1, flat.size // 257)]])
                            gold[f"ds/{key}/{mode.value}/{i}/{k}"] = v
                        elif k == "rgb_relative_path":
                            gold[f"ds/{key}/{mode.value}/{i}/{k}"] = np.array(v)
                if hasattr(ds, "min_depth"):
                    gold[f"ds/{key}/range"] = np.array([ds.min_depth, float(ds.max_depth)])
    np.savez_compressed(OUT, **gold)
    print(f"wrote {OUT}: {len(gold)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


class _AttrDict(dict):
    """The two OmegaConf behaviours get_dataset relies on: attribute access and ** expansion."""
    __getattr__ = dict.__getitem__


if __name__ == "__main__":
    main()
