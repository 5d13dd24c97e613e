"""Image / depth / intrinsics pyramids -- the reference's ``image/gaussian_pyramid.py`` API.

The image pyramid (3x3 binomial blur with reflect padding, then every second row and column,
``image/gaussian_pyramid.py:53-85``) runs as one fused HIP kernel per level (``sp_blur_decimate``) instead of
pad + depthwise conv2d + strided slice.  Depth/mask pyramids are pure strided views or poolings that no hot-path
caller enables (``geo_down`` is never set) and stay thin tensor expressions."""
from __future__ import annotations

import torch
import torch.nn.functional as nnf
from torch import nn

from .. import _lib


def blur_decimate(x):
    """(B,C,H,W) cuda f32 -> (B,C,ceil(H/2),ceil(W/2))."""
    _lib.require_device(x)
    lib = _lib.load()
    B, C, H, W = x.shape
    xin = x.detach().contiguous().float()
    out = torch.empty(B, C, (H + 1) // 2, (W + 1) // 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.sp_blur_decimate(_lib.ptr(xin), B * C, H, W, _lib.ptr(out), _lib.stream_ptr()), "sp_blur_decimate")
    return out


def pyr_depth(depth, mode, kernel_size):
    """One factor-2 depth pyramid step (image/gaussian_pyramid.py:8-29)."""
    k = kernel_size
    if mode == "bilinear":
        return nnf.avg_pool2d(depth, k, k)
    if mode == "nearest_neighbor":
        return depth[:, :, 0::k, 0::k]
    if mode == "max":
        return nnf.max_pool2d(depth, k)
    if mode == "min":
        return -nnf.max_pool2d(-depth, k)
    if mode == "masked_bilinear":
        ok = ~depth.isnan()
        filled = torch.where(ok, depth, torch.zeros_like(depth))
        total = nnf.avg_pool2d(filled, k, k, divisor_override=1)
        count = nnf.avg_pool2d(ok.float(), k, k, divisor_override=1)
        return torch.where(count > 0.0, total / count, torch.zeros((), dtype=depth.dtype, device=depth.device))
    raise ValueError("pyr_depth mode: " + mode + " is not implemented.")


def resize_depth(depth, mode, size):
    """image/gaussian_pyramid.py:31-39 (torchvision resize upstream; interpolate here, no hot-path caller)."""
    if mode == "bilinear":
        return nnf.interpolate(depth, size=size, mode="bilinear", align_corners=False, antialias=True)
    if mode == "nearest_neighbor":
        return nnf.interpolate(depth, size=size, mode="nearest")
    raise ValueError("resize_depth mode: " + mode + " is not implemented.")


def resize_intrinsics(K, image_scale_factors):
    """K_l = [[sx,0,sx],[0,sy,sy],[0,0,1]] @ K -- including the reference's principal-point quirk
    cx_l = s*cx + s (image/gaussian_pyramid.py:42-50)."""
    sy, sx = image_scale_factors[0], image_scale_factors[1]
    S = torch.tensor([[sx, 0, sx], [0, sy, sy], [0, 0, 1]], device=K.device, dtype=K.dtype)
    return S @ K


class GaussianBlurModule(nn.Module):
    """Full-resolution 3x3 binomial blur with reflect padding (image/gaussian_pyramid.py:53-66)."""

    def __init__(self, channels, device, dtype):
        super().__init__()
        k = torch.tensor([1.0, 2.0, 1.0], device=device, dtype=dtype)
        self.gaussian_kernel = (torch.outer(k, k) / 16.0).repeat(channels, 1, 1, 1)

    def forward(self, x):
        return nnf.conv2d(nnf.pad(x, (1, 1, 1, 1), mode="reflect"), self.gaussian_kernel, groups=x.shape[1])


class ImagePyramidModule(nn.Module):
    """Levels start..end-1, returned coarse -> fine (image/gaussian_pyramid.py:69-85)."""

    def __init__(self, channels, start_level, end_level, device, dtype):
        super().__init__()
        self.blur_module = GaussianBlurModule(channels=channels, device=device, dtype=dtype)
        self.start_level, self.end_level = start_level, end_level

    def forward(self, x):
        levels, cur = [], x
        for i in range(self.end_level - 1):
            if i >= self.start_level:
                levels.insert(0, cur)
            cur = blur_decimate(cur)
        levels.insert(0, cur)
        return levels


class DepthPyramidModule(nn.Module):
    def __init__(self, start_level, end_level, mode, device):
        super().__init__()
        self.start_level, self.end_level, self.mode = start_level, end_level, mode

    def forward(self, x):
        levels, cur = [], x
        for i in range(self.end_level - 1):
            if i >= self.start_level:
                levels.insert(0, cur)
            cur = pyr_depth(cur, self.mode, kernel_size=2)
        levels.insert(0, cur)
        return levels


class IntrinsicsPyramidModule(nn.Module):
    def __init__(self, start_level, end_level, device):
        super().__init__()
        self.start_level, self.end_level = start_level, end_level

    def forward(self, K_orig, image_scale_start):
        levels = []
        for i in range(self.start_level, self.end_level):
            s = 2.0 ** (-i)
            levels.insert(0, resize_intrinsics(K_orig, [image_scale_start[0] * s, image_scale_start[1] * s]))
        return levels
