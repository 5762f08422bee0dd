"""Image helpers with the reference's semantics (marigold/util/image_util.py), restated on
torch.nn.functional because torchvision is not a dependency here.

``resize`` reproduces torchvision.transforms.functional.resize(img, size, mode, antialias=True)
(SURVEY.md App. C.8): == F.interpolate(..., align_corners=False, antialias=True); uint8 inputs are
computed in float, rounded and cast back; the input is returned unchanged when the size already
matches; NEAREST_EXACT == mode "nearest-exact".  CUDA tensors (uint8 / fp32) go through the HIP
resampling kernels (csrc/resize.hip, MG_OP_RESIZE); host tensors use torch's CPU implementation,
which is also the parity reference of the kernels.
"""
import enum

import numpy as np
import torch


class InterpolationMode(enum.Enum):
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"
    NEAREST_EXACT = "nearest-exact"


_HIP_MODES = {"bilinear": 0, "bicubic": 1, "nearest-exact": 2}


def _resize_hip(img, h, w, mode):
    """Device path (csrc/resize.hip): uint8 or fp32 CUDA tensors [..., H, W]."""
    from .. import ops as O
    src = img.contiguous()
    planes = src.numel() // (src.shape[-2] * src.shape[-1])
    Hin, Win = src.shape[-2:]
    dst = torch.empty(src.shape[:-2] + (h, w), dtype=src.dtype, device=src.device)
    tmp = None
    if mode != "nearest-exact" and Hin != h and Win != w:
        tmp = torch.empty(planes * Hin * w, dtype=torch.float32, device=src.device)
    O.launch(O.resize(src, dst, tmp, planes=planes, Hin=Hin, Win=Win, Hout=h, Wout=w, mode=_HIP_MODES[mode],
                      u8=src.dtype == torch.uint8))
    return dst


def resize(img: torch.Tensor, size, interpolation=InterpolationMode.BILINEAR, antialias=True):
    h, w = int(size[0]), int(size[1])
    if tuple(img.shape[-2:]) == (h, w):
        return img
    mode = interpolation.value
    if img.is_cuda and antialias and img.dtype in (torch.uint8, torch.float32):
        return _resize_hip(img, h, w, mode)
    if mode == "nearest-exact":
        return torch.nn.functional.interpolate(img, size=(h, w), mode=mode)
    x = img
    is_int = not torch.is_floating_point(img)
    if is_int or img.dtype in (torch.bfloat16, torch.float16):
        x = img.to(torch.float32)
    y = torch.nn.functional.interpolate(x, size=(h, w), mode=mode, align_corners=False, antialias=antialias)
    if is_int:
        if mode == "bicubic":
            y = y.clamp(0, 255)
        y = y.round().to(img.dtype)
    elif y.dtype != img.dtype:
        y = y.to(img.dtype)
    return y


def resize_max_res(img: torch.Tensor, max_edge_resolution: int,
                   resample_method: InterpolationMode = InterpolationMode.BILINEAR) -> torch.Tensor:
    """Resize so the longer edge equals ``max_edge_resolution`` (may up-scale; no rounding to a
    multiple of 8) - reference :90-120."""
    assert 4 == img.dim(), f"Invalid input shape {img.shape}"
    original_height, original_width = img.shape[-2:]
    downscale_factor = min(max_edge_resolution / original_width, max_edge_resolution / original_height)
    new_width = int(original_width * downscale_factor)
    new_height = int(original_height * downscale_factor)
    return resize(img, (new_height, new_width), resample_method, antialias=True)


def get_tv_resample_method(method_str: str) -> InterpolationMode:
    table = {"bilinear": InterpolationMode.BILINEAR, "bicubic": InterpolationMode.BICUBIC,
             "nearest": InterpolationMode.NEAREST_EXACT, "nearest-exact": InterpolationMode.NEAREST_EXACT}
    m = table.get(method_str, None)
    if m is None:
        raise ValueError(f"Unknown resampling method: {m}")
    return m


def colorize_depth_maps(depth_map, min_depth, max_depth, cmap="Spectral", valid_mask=None):
    """matplotlib colormap lookup -> float [ (B,) 3, H, W ] in (0, 1) - reference :38-76."""
    import matplotlib

    assert len(depth_map.shape) >= 2, "Invalid dimension"
    if isinstance(depth_map, torch.Tensor):
        depth = depth_map.detach().squeeze().numpy()
    else:
        depth = np.asarray(depth_map).copy().squeeze()
    if depth.ndim < 3:
        depth = depth[np.newaxis, :, :]
    cm = matplotlib.colormaps[cmap]
    depth = ((depth - min_depth) / (max_depth - min_depth)).clip(0, 1)
    img = cm(depth, bytes=False)[:, :, :, 0:3]
    img = np.rollaxis(img, 3, 1)
    if valid_mask is not None:
        if isinstance(valid_mask, torch.Tensor):
            valid_mask = valid_mask.detach().numpy()
        valid_mask = valid_mask.squeeze()
        valid_mask = valid_mask[np.newaxis, np.newaxis] if valid_mask.ndim < 3 else valid_mask[:, np.newaxis]
        img[~np.repeat(valid_mask, 3, axis=1)] = 0
    if isinstance(depth_map, torch.Tensor):
        return torch.from_numpy(img).float()
    return img


_LUT_CACHE = {}


def colormap_lut_u8(cmap):
    """matplotlib's 256-entry table of ``cmap`` as the uint8 RGB values the reference ends up with:
    (cm(k / 256 + eps)[:3] * 255).astype(uint8) for table entry k."""
    import matplotlib
    cm = matplotlib.colormaps[cmap]
    idx = (np.arange(256, dtype=np.float64) + 0.5) / 256.0          # x with int(x * 256) == k
    return (cm(idx, bytes=False)[:, 0:3] * 255).astype(np.uint8)


def colorize_depth_device(depth: torch.Tensor, min_depth=0.0, max_depth=1.0, cmap="Spectral") -> torch.Tensor:
    """Device form of ``(colorize_depth_maps(depth, lo, hi, cmap) * 255).astype(uint8)`` in HWC order: fp32 CUDA map
    [H, W] -> uint8 CUDA image [H, W, 3] (csrc/resize.hip, MG_OP_COLORIZE: one table look-up pass)."""
    from .. import ops as O
    assert depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 2
    key = (cmap, depth.device)
    if key not in _LUT_CACHE:
        _LUT_CACHE[key] = torch.from_numpy(colormap_lut_u8(cmap)).to(depth.device).contiguous()
    d = depth.contiguous()
    out = torch.empty(d.shape + (3,), dtype=torch.uint8, device=d.device)
    O.launch(O.colorize(d, _LUT_CACHE[key], out, n=d.numel(), lo=min_depth, hi=max_depth))
    return out


def chw2hwc(chw):
    assert 3 == len(chw.shape)
    if isinstance(chw, torch.Tensor):
        return torch.permute(chw, (1, 2, 0))
    if isinstance(chw, np.ndarray):
        return np.moveaxis(chw, 0, -1)
    raise TypeError("img should be np.ndarray or torch.Tensor")


def pil_to_tensor(img):
    """PIL RGB -> uint8 [3,H,W] (no scaling), like torchvision's pil_to_tensor."""
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


def float2int(img):
    """[0,1] float -> uint8 by truncation (reference image_util.py:137-141)."""
    if isinstance(img, np.ndarray):
        return (img * 255.0).astype(np.uint8)
    return (img * 255.0).to(torch.uint8)


def srgb2linear(img):
    """gamma-2.2 decode (reference image_util.py:144-145)."""
    return img ** 2.2


def linear2srgb(img):
    return img ** (1.0 / 2.2)
