"""Image / depth / intrinsics pyramids with the call surface of the reference's ``image/gaussian_pyramid.py``.

One image level = 3x3 binomial blur with reflect padding followed by dropping every second row and column
(reference ``:53-85``); here that is a single fused HIP kernel per level (``sp_blur_decimate``) instead of
pad + depthwise conv2d + strided slice.  The depth / mask pyramids are only built when ``geo_down`` is set, which no
caller does; they stay thin tensor expressions.  All three pyramid modules share ``_CoarseToFine``:
apply a factor-2 step ``end_level - 1`` times, keep what falls at or above ``start_level``, return coarsest first."""
from __future__ import annotations

import torch
import torch.nn.functional as nnf
from torch import nn

from .. import _lib

_BINOMIAL = (1.0, 2.0, 1.0)


def blur_decimate(x):
    """(B,C,H,W) cuda f32 -> (B,C,ceil(H/2),ceil(W/2)): blur and decimation in one pass over the image."""
    _lib.require_device(x)
    planes, rows, cols = x.shape[0] * x.shape[1], x.shape[2], x.shape[3]
    half = torch.empty(x.shape[0], x.shape[1], (rows + 1) // 2, (cols + 1) // 2, dtype=torch.float32, device=x.device)
    source = x.detach().contiguous().float()
    status = _lib.load().sp_blur_decimate(_lib.ptr(source), planes, rows, cols, _lib.ptr(half), _lib.stream_ptr())
    _lib.check(status, "sp_blur_decimate")
    return half


def _masked_mean_pool(depth, k):
    """Mean over the finite entries of every k x k cell; cells without any become 0."""
    finite = ~torch.isnan(depth)
    sums = nnf.avg_pool2d(depth.masked_fill(~finite, 0.0), k, k, divisor_override=1)
    hits = nnf.avg_pool2d(finite.to(depth.dtype), k, k, divisor_override=1)
    return torch.where(hits > 0, sums / hits, torch.zeros_like(sums))


_DEPTH_STEPS = {
    "bilinear": lambda d, k: nnf.avg_pool2d(d, k, k),
    "nearest_neighbor": lambda d, k: d[..., ::k, ::k],
    "max": lambda d, k: nnf.max_pool2d(d, k),
    "min": lambda d, k: nnf.max_pool2d(d.neg(), k).neg(),
    "masked_bilinear": _masked_mean_pool,
}


def pyr_depth(depth, mode, kernel_size):
    """One factor-``kernel_size`` step of a depth pyramid in the given pooling ``mode`` (reference ``:8-29``)."""
    step = _DEPTH_STEPS.get(mode)
    if step is None:
        raise ValueError(f"pyr_depth mode: {mode} is not implemented.")
    return step(depth, kernel_size)


def resize_depth(depth, mode, size):
    """Resample a depth map to ``size`` (reference ``:31-39`` goes through torchvision; no hot-path caller)."""
    if mode == "nearest_neighbor":
        return nnf.interpolate(depth, size=size, mode="nearest")
    if mode == "bilinear":
        return nnf.interpolate(depth, size=size, mode="bilinear", align_corners=False, antialias=True)
    raise ValueError(f"resize_depth mode: {mode} is not implemented.")


def resize_intrinsics(K, image_scale_factors):
    """Intrinsics of a level scaled by (sy, sx): diag(sx, sy, 1) @ K with the principal point additionally shifted by
    the scale itself, cx_l = sx * cx + sx -- the reference's convention (``:42-50``), reproduced as is."""
    sy, sx = float(image_scale_factors[0]), float(image_scale_factors[1])
    scale = K.new_tensor([[sx, 0.0, sx], [0.0, sy, sy], [0.0, 0.0, 1.0]])
    return scale @ K


class _CoarseToFine(nn.Module):
    def __init__(self, start_level, end_level):
        super().__init__()
        self.start_level, self.end_level = start_level, end_level

    def _step(self, x):
        raise NotImplementedError

    def forward(self, x):
        kept = []
        for level in range(self.end_level):
            if level >= self.start_level:
                kept.append(x)
            if level + 1 < self.end_level:
                x = self._step(x)
        if not kept:                       # start_level >= end_level: the reference still returns its last level
            kept.append(x)
        return kept[::-1]


class GaussianBlurModule(nn.Module):
    """Full-resolution 3x3 binomial blur with reflect padding, as a depthwise convolution (reference ``:53-66``)."""

    def __init__(self, channels, device, dtype):
        super().__init__()
        taps = torch.tensor(_BINOMIAL, device=device, dtype=dtype)
        self.gaussian_kernel = (taps[:, None] * taps[None, :] / taps.sum() ** 2).expand(channels, 1, 3, 3).contiguous()

    def forward(self, x):
        padded = nnf.pad(x, (1, 1, 1, 1), mode="reflect")
        return nnf.conv2d(padded, self.gaussian_kernel, groups=x.shape[1])


class ImagePyramidModule(_CoarseToFine):
    """Image levels ``start_level .. end_level - 1``, coarse -> fine (reference ``:69-85``)."""

    def __init__(self, channels, start_level, end_level, device, dtype):
        super().__init__(start_level, end_level)
        self.blur_module = GaussianBlurModule(channels=channels, device=device, dtype=dtype)   # API parity; unused on the HIP path

    def _step(self, x):
        return blur_decimate(x)


class DepthPyramidModule(_CoarseToFine):
    def __init__(self, start_level, end_level, mode, device):
        super().__init__(start_level, end_level)
        self.mode = mode

    def _step(self, x):
        return pyr_depth(x, self.mode, kernel_size=2)


class IntrinsicsPyramidModule(nn.Module):
    def __init__(self, start_level, end_level, device):
        super().__init__()
        self.start_level, self.end_level = start_level, end_level

    def forward(self, K_orig, image_scale_start):
        y0, x0 = image_scale_start[0], image_scale_start[1]
        fine_to_coarse = [resize_intrinsics(K_orig, [y0 * 0.5 ** level, x0 * 0.5 ** level])
                          for level in range(self.start_level, self.end_level)]
        return fine_to_coarse[::-1]
