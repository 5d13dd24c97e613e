"""Depth render of a keyframe's segments in another pose -- ``core/depth_render.py:7-21`` on the HIP path.

Used for the keyframe decision and to seed a new keyframe's depths (``odometery/odometery.py:294-298``).  The
reference unprojects, transforms and ``scatter_``s z at truncated pixel positions with an undefined winner when
two points land on one pixel; ``sp_depth_splat`` resolves collisions by highest point index (the result of a
sequential scatter), so the output is reproducible."""
from __future__ import annotations

import torch

from .. import _lib
from ..segment_table import table_of


def estimate_depth_kf_native(kf, kf_logdepth, pose=None, mean=False):
    """``mean=True``: the reference's ``scatter_reduce_(..., reduce='mean')`` branch (core/ops.py:84-92) with its default
    ``include_self=True`` -- the zero-initialised image counts as one sample, so a pixel hit by c points holds sum / (c + 1).  No
    reference caller uses it; kept for API completeness (``sp_depth_splat_mean``, golden g6)."""
    _lib.require_device(kf.image, kf_logdepth)
    lib = _lib.load()
    table = table_of(kf)
    dev = table.device
    if pose is None:
        pose = torch.eye(4, device=dev)
    H, W = table.H, table.W
    keys = torch.empty(H * W, dtype=torch.int64, device=dev)
    out = torch.empty(H, W, dtype=torch.float32, device=dev)
    f = lambda t: t.detach().contiguous().float()
    if mean:
        acc = torch.empty(12 * H * W, dtype=torch.uint8, device=dev)
        rc = lib.sp_depth_splat_mean(_lib.ptr(table.pix), _lib.ptr(table.baseL), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L),
                                     _lib.ptr(f(kf_logdepth)), table.N, table.P, H, W, _lib.ptr(f(kf.K)), _lib.ptr(f(pose)),
                                     _lib.ptr(acc), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "sp_depth_splat_mean")
        return out
    rc = lib.sp_depth_splat(_lib.ptr(table.pix), _lib.ptr(table.baseL), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L),
                            _lib.ptr(f(kf_logdepth)), table.N, table.P, H, W, _lib.ptr(f(kf.K)), _lib.ptr(f(pose)),
                            _lib.ptr(keys), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "sp_depth_splat")
    return out
