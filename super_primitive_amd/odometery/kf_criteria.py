"""Keyframe criteria -- the reference's ``odometery/kf_criteria.py:7-34`` on the HIP path.

``translation_difference`` needs the median of the valid rendered depths (``torch.median`` over a boolean-indexed
image: a compaction + a sort + a host sync in the reference) and ``rotation_difference`` round-trips both poses
through numpy/scipy.  Here four launches (``sp_kf_criterion_ws``: count + 4-pass radix select over a grid of workgroups + the two
pose differences) fills a 4-vector that stays on the device; the functions below slice it with the reference's
signatures and return types.  ``keyframe_criterion`` exposes the whole vector, including the depth-validity ratio of
``odometery/odometery.py:1003-1004``, for drivers that want a single read-back per frame."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def keyframe_criterion(pose_src, pose_target, depth, valid_thresh=1e-6):
    """(4,) f32 device tensor: ``[validity_ratio, scale, translation_difference, rotation_difference_deg]``."""
    lib = _lib.load()
    _lib.require_device(depth)
    dev = depth.device
    d = depth.detach().contiguous().float()
    ps = pose_src.detach().to(dev).contiguous().float()
    pt = pose_target.detach().to(dev).contiguous().float()
    assert ps.shape[-2:] == (4, 4) and pt.shape[-2:] == (4, 4) and ps.numel() == 16 and pt.numel() == 16
    out = torch.empty(4, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.sp_kf_criterion_ws_words(), dtype=torch.int32, device=dev)
    _lib.check(lib.sp_kf_criterion_ws(_lib.ptr(d), d.numel(), float(valid_thresh), _lib.ptr(ps), _lib.ptr(pt), _lib.ptr(ws), _lib.ptr(out),
                                      _lib.stream_ptr()), "sp_kf_criterion_ws")
    return out


def translation_difference(pose_src, pose_target, depth):
    """kf_criteria.py:7-21: ``(|t_src - t_target| / (median(depth[depth > 1e-6]) + 1e-6), that median)`` as 0-d tensors."""
    out = keyframe_criterion(pose_src, pose_target, depth)
    return out[2], out[1]


def rotation_difference(pose_src, pose_target):
    """kf_criteria.py:23-34: rotation angle of ``inv(pose_src) @ pose_target`` in degrees, as a numpy float64 (the
    reference goes through numpy and scipy's ``Rotation``; this is the one host read-back of the function)."""
    dev = pose_src.device
    dummy = torch.ones(1, dtype=torch.float32, device=dev)
    out = keyframe_criterion(pose_src, pose_target, dummy)
    return np.float64(out[3].item())
