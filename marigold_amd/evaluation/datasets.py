"""Benchmark dataset readers for inference / evaluation (reference: src/dataset/*.py).

One table (``SPECS``) describes every dataset the reference registers (src/dataset/__init__.py:56-74):
how a ground-truth file decodes, the valid-depth range, the evaluation crop / mask and how prediction
files are named.  ``BenchmarkDataset`` applies a row of that table to a folder or a ``.tar`` archive and
yields per-image dicts with the reference's keys (``rgb_int``, ``rgb_norm``, ``depth_raw_linear``,
``valid_mask_raw``, ``normals``, ``rgb``, ``albedo`` ..., ``index``, ``rgb_relative_path``) as numpy
arrays.  The dataset YAML files of the reference (config/dataset_*/data_*.yaml: ``name``, ``disp_name``,
``dir``, ``filenames`` + per-dataset switches) load unchanged through ``load_dataset_config``.

Only the inference / evaluation modes are provided (the training modes - augmentation, depth
normalisation, mixed sampling - belong to the trainer, which is out of scope).
"""
import io
import os
import tarfile
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Optional

import numpy as np
import yaml
from PIL import Image


class DatasetMode(Enum):
    RGB_ONLY = "rgb_only"
    EVAL = "evaluate"


class PredNameMode(Enum):
    """How a prediction file is named after its RGB file (base_depth_dataset.py:51-57, :263-281)."""
    id = 1        # 0001.png            -> pred_0001.npy
    rgb_id = 2    # rgb_0001.png        -> pred_0001.npy
    i_d_rgb = 3   # 0_1_rgb.png         -> 0_1_pred.npy
    rgb_i_d = 4   # rgb_cam_00_fr01.png -> pred_cam_00_fr01.npy


def get_pred_name(rgb_basename, name_mode, suffix=".png"):
    if name_mode == PredNameMode.rgb_id:
        stem = "pred_" + rgb_basename.split("_")[1]
    elif name_mode == PredNameMode.i_d_rgb:
        stem = rgb_basename.replace("_rgb.", "_pred.")
    elif name_mode == PredNameMode.id:
        stem = "pred_" + rgb_basename
    elif name_mode == PredNameMode.rgb_i_d:
        stem = "pred_" + "_".join(rgb_basename.split("_")[1:])
    else:
        raise NotImplementedError(name_mode)
    return os.path.splitext(stem)[0] + suffix


# ---- storage: a folder or a tar archive ------------------------------------------------------------


class Storage:
    """Relative-path reads from a dataset folder or from a tar archive whose members are ``./<rel>``."""

    def __init__(self, root):
        self._tar = None
        assert os.path.exists(root), f"Dataset does not exist at: {root}"
        self.root = root
        self.is_tar = os.path.isfile(root) and tarfile.is_tarfile(root)

    def read_bytes(self, rel):
        if self.is_tar:
            if self._tar is None:
                self._tar = tarfile.open(self.root)
            return self._tar.extractfile("./" + rel).read()
        with open(os.path.join(self.root, rel), "rb") as f:
            return f.read()

    def read_image(self, rel):
        return np.asarray(Image.open(io.BytesIO(self.read_bytes(rel))))

    def read_npy(self, rel):
        return np.load(io.BytesIO(self.read_bytes(rel)))

    def close(self):
        if self._tar is not None:
            self._tar.close()
            self._tar = None

    def __del__(self):
        self.close()


# ---- the dataset table -----------------------------------------------------------------------------

KITTI_BENCHMARK_HW = (352, 1216)
ETH3D_HW = (4032, 6048)
SINTEL_W, SINTEL_SIDE_CROP = 1024, 221


def kitti_benchmark_crop(a):
    """Bottom-aligned, horizontally centred 352 x 1216 window (kitti_dataset.py:82-112)."""
    h, w = a.shape[-2:]
    top, left = int(h - KITTI_BENCHMARK_HW[0]), int((w - KITTI_BENCHMARK_HW[1]) / 2)
    return a[..., top:top + KITTI_BENCHMARK_HW[0], left:left + KITTI_BENCHMARK_HW[1]]


def _window_mask(shape_hw, rows, cols):
    m = np.zeros(shape_hw, dtype=bool)
    m[rows[0]:rows[1], cols[0]:cols[1]] = True
    return m


def _kitti_eval_window(shape_hw, kind):
    """Garg (ECCV16) / Eigen (NIPS14) evaluation windows as fractions of the image (kitti_dataset.py:114-134)."""
    h, w = shape_hw
    if kind == "garg":
        return _window_mask(shape_hw, (int(0.40810811 * h), int(0.99189189 * h)),
                            (int(0.03594771 * w), int(0.96405229 * w)))
    if kind == "eigen":
        return _window_mask(shape_hw, (int(0.3324324 * h), int(0.91351351 * h)),
                            (int(0.0359477 * w), int(0.96405229 * w)))
    raise AssertionError(f"Unknown crop type: {kind}")


def _scaled_png(divisor):
    return lambda st, rel: st.read_image(rel) / divisor


def _eth3d_raw(st, rel):
    """Headerless little-endian float32 raster, +inf marks missing depth (eth3d_dataset.py:55-73)."""
    d = np.frombuffer(st.read_bytes(rel), dtype=np.float32).copy()
    d[d == np.inf] = 0.0
    return d.reshape(ETH3D_HW)


def _npy_plane(st, rel):
    return st.read_npy(rel).squeeze()


@dataclass(frozen=True)
class Spec:
    kind: str                                   # "depth" | "normals" | "iid"
    min_depth: float = 0.0
    max_depth: float = 0.0
    has_filled_depth: bool = False
    name_mode: Optional[PredNameMode] = None
    decode: Optional[Callable] = None           # (storage, rel_path) -> [H,W] metric depth
    mask_from_file: bool = False                # DIODE: the validity mask is the 3rd column of the split file
    kitti_like: bool = False                    # kitti_bm_crop / valid_mask_crop switches, "None" GT rows dropped
    sintel_crop: bool = False


SPECS = {
    # depth (nyu_dataset.py, kitti_dataset.py, vkitti_dataset.py, eth3d_dataset.py, diode_dataset.py,
    # scannet_dataset.py, hypersim_dataset.py)
    "nyu_depth": Spec("depth", 1e-3, 10.0, True, PredNameMode.rgb_id, _scaled_png(1000.0)),
    "kitti_depth": Spec("depth", 1e-5, 80, False, PredNameMode.id, _scaled_png(256.0), kitti_like=True),
    "vkitti_depth": Spec("depth", 1e-5, 80, False, PredNameMode.id, _scaled_png(100.0), kitti_like=True),
    "eth3d_depth": Spec("depth", 1e-5, float("inf"), False, PredNameMode.id, _eth3d_raw),
    "diode_depth": Spec("depth", 0.6, 350, False, PredNameMode.id, _npy_plane, mask_from_file=True),
    "scannet_depth": Spec("depth", 1e-3, 10, False, PredNameMode.id, _scaled_png(1000.0)),
    "hypersim_depth": Spec("depth", 1e-5, 65.0, False, PredNameMode.rgb_i_d, _scaled_png(1000.0)),
    # normals: [H,W,3] .npy files (base_normals_dataset.py:151-164)
    **{n: Spec("normals") for n in ("hypersim_normals", "interiorverse_normals", "ibims_normals", "nyu_normals",
                                     "scannet_normals", "diode_normals", "oasis_normals")},
    "sintel_normals": Spec("normals", sintel_crop=True),
    # intrinsic decomposition
    "hypersim_iid": Spec("iid"),
    "interiorverse_iid": Spec("iid"),
}


def load_dataset_config(path):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict) or "name" not in cfg:
        raise ValueError(f"{path}: not a dataset config (needs at least name / disp_name / dir / filenames)")
    return cfg


def get_dataset(cfg, base_data_dir, mode):
    """Build the reader a dataset config names (src/dataset/__init__.py:77-107).  ``mixed`` lists exist for
    training only."""
    name = cfg["name"]
    if name == "mixed":
        raise AssertionError("Only training mode supports mixed datasets.")
    if name not in SPECS:
        raise NotImplementedError(name)
    extra = {k: v for k, v in cfg.items() if k not in ("name", "disp_name", "dir", "filenames")}
    return BenchmarkDataset(name, mode, cfg["filenames"], os.path.join(base_data_dir, cfg["dir"]),
                            cfg.get("disp_name", name), **extra)


class BenchmarkDataset:
    def __init__(self, name, mode, filename_ls_path, dataset_dir, disp_name, eigen_valid_mask=False,
                 kitti_bm_crop=False, valid_mask_crop=None, **unused):
        self.spec = SPECS[name]
        self.name, self.mode, self.disp_name = name, DatasetMode(mode), disp_name
        self.filename_ls_path, self.dataset_dir = filename_ls_path, dataset_dir
        self.storage = Storage(dataset_dir)
        self.min_depth, self.max_depth = self.spec.min_depth, self.spec.max_depth
        self.has_filled_depth, self.name_mode = self.spec.has_filled_depth, self.spec.name_mode
        self.eigen_valid_mask = bool(eigen_valid_mask) and name == "nyu_depth"
        self.kitti_bm_crop = bool(kitti_bm_crop) and self.spec.kitti_like
        self.valid_mask_crop = valid_mask_crop if self.spec.kitti_like else None
        assert self.valid_mask_crop in (None, "garg", "eigen"), f"Unknown crop type: {self.valid_mask_crop}"
        with open(filename_ls_path) as f:
            self.filenames = [line.split() for line in f.readlines()]
        if self.spec.kitti_like:
            self.filenames = [f for f in self.filenames if f[1] != "None"]

    def __len__(self):
        return len(self.filenames)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, index):
        line = self.filenames[index]
        sample = {"index": index, "rgb_relative_path": line[0]}
        sample.update(getattr(self, "_load_" + self.spec.kind)(line))
        return sample

    # -- RGB ---------------------------------------------------------------------------------------
    def _rgb_int(self, rel):
        rgb = np.transpose(self.storage.read_image(rel), (2, 0, 1)).astype(np.int64)   # [3,H,W]
        if self.kitti_bm_crop:
            rgb = kitti_benchmark_crop(rgb)
        if self.spec.sintel_crop:
            rgb = rgb[:, :, SINTEL_SIDE_CROP:SINTEL_W - SINTEL_SIDE_CROP]
        return {"rgb_int": rgb.astype(np.int32), "rgb_norm": (rgb / 255.0 * 2.0 - 1.0).astype(np.float32)}

    # -- depth -------------------------------------------------------------------------------------
    def valid_mask(self, depth):
        """(min_depth, max_depth) open interval, intersected with the dataset's evaluation window
        (base_depth_dataset.py:213-217; nyu_dataset.py:54-63; kitti_dataset.py:114-134)."""
        m = (depth > self.min_depth) & (depth < self.max_depth)
        hw = depth.shape[-2:]
        if self.eigen_valid_mask:
            m = m & _window_mask(hw, (45, 471), (41, 601))
        if self.valid_mask_crop is not None:
            m = m & _kitti_eval_window(hw, self.valid_mask_crop)
        return m

    def _depth(self, rel):
        d = np.asarray(self.spec.decode(self.storage, rel)).squeeze().astype(np.float32)[None]   # [1,H,W]
        return kitti_benchmark_crop(d) if self.kitti_bm_crop else d

    def _load_depth(self, line):
        out = self._rgb_int(line[0])
        if self.mode == DatasetMode.RGB_ONLY:
            return out
        raw = self._depth(line[1])
        filled = self._depth(line[2]) if self.has_filled_depth else raw.copy()
        out.update(depth_raw_linear=raw, depth_filled_linear=filled)
        if self.spec.mask_from_file:
            m = self.storage.read_npy(line[2]).squeeze()[None].astype(bool)
            out.update(valid_mask_raw=m, valid_mask_filled=m.copy())
        else:
            out.update(valid_mask_raw=self.valid_mask(raw), valid_mask_filled=self.valid_mask(filled))
        return out

    # -- normals -----------------------------------------------------------------------------------
    def _load_normals(self, line):
        out = self._rgb_int(line[0])
        if self.mode == DatasetMode.RGB_ONLY:
            return out
        n = np.transpose(self.storage.read_npy(line[1]), (2, 0, 1)).astype(np.float32)   # [3,H,W]
        if self.spec.sintel_crop:
            # sky pixels carry no normal: replace by the camera-facing one, then crop (sintel_dataset.py:57-74)
            sky = ~(np.sqrt((n * n).sum(0)) > 0.1)
            n[:, sky] = np.array([0.0, 0.0, 1.0], np.float32)[:, None]
            n = n[:, :, SINTEL_SIDE_CROP:SINTEL_W - SINTEL_SIDE_CROP]
        out["normals"] = n
        return out

    # -- intrinsic image decomposition ---------------------------------------------------------------
    def _iid_image(self, rel):
        """[3,H,W] (or [H,W]) float image in [0,1]; ``.exr`` files are HDR (src/util/image_util.py:99-128)."""
        if rel.endswith(".exr"):
            try:
                import cv2
            except ImportError as e:
                raise RuntimeError(f"'{rel}': OpenEXR files need OpenCV, which this image does not ship") from e
            img = cv2.imdecode(np.frombuffer(self.storage.read_bytes(rel), np.uint8), cv2.IMREAD_UNCHANGED)
            img = np.clip(cv2.cvtColor(img, cv2.COLOR_BGR2RGB), 0, 1)
        else:
            img = self.storage.read_image(rel) / 255.0
        img = np.transpose(img, (2, 0, 1)) if img.ndim == 3 else img
        assert img.min() >= 0 and img.max() <= 1
        return img

    def _load_iid(self, line):
        rgb = self._iid_image(line[0])
        if line[0].endswith(".exr"):
            rgb = rgb ** (1 / 2.2)   # the model works in sRGB
        out = {"rgb": rgb.astype(np.float32)}
        if self.mode == DatasetMode.RGB_ONLY:
            return out
        out.update(self._hypersim_targets(line[1:]) if self.name == "hypersim_iid"
                   else self._interiorverse_targets(line[1:]))
        return out

    def _hypersim_targets(self, rels):
        """albedo / shading / residual ``[H,W,3]`` .npy rasters in linear space; shading and residual are
        clipped at the larger of their two 98th percentiles and divided by it (hypersim_dataset.py:74-143)."""
        albedo, shading_raw, residual_raw = (np.transpose(self.storage.read_npy(r), (2, 0, 1)) for r in rels[:3])
        albedo = albedo.astype(np.float32)
        cut = max(np.quantile(residual_raw, 0.98), np.quantile(shading_raw, 0.98)).astype(residual_raw.dtype)
        shading = (np.clip(shading_raw, 0, cut) / cut).astype(np.float32)
        residual = (np.clip(residual_raw, 0, cut) / cut).astype(np.float32)
        black = np.broadcast_to((albedo == 0).all(axis=0, keepdims=True), albedo.shape)
        return {"albedo": albedo, "shading_raw": shading_raw, "residual_raw": residual_raw,
                "shading": shading, "residual": residual,
                "mask_albedo": ~(~np.isfinite(albedo) | black), "mask_shading": np.isfinite(shading),
                "mask_residual": np.isfinite(residual)}

    def _interiorverse_targets(self, rels):
        """albedo, material (R roughness, G metallicity, B unused -> 0) and the validity mask
        (interiorverse_dataset.py:46-83)."""
        albedo, material = self._iid_image(rels[0]), self._iid_image(rels[1])
        material[2] = 0
        mask = self._iid_image(rels[2]) != 0
        if rels[0].endswith(".exr"):
            albedo = albedo ** (1 / 2.2)
        if rels[1].endswith(".exr"):
            material = material ** (1 / 2.2)
        out = {"albedo": albedo, "material": material, "mask": np.all(mask, axis=0, keepdims=True)}
        if self.mode == DatasetMode.EVAL:
            out.update(mask_albedo=mask.astype(bool), mask_material=mask.astype(bool))
        return out
