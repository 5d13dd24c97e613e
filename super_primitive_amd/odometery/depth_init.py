"""Per-segment log-depth (re)initialisation against a rendered / sparse depth map -- the reference's
``odometery/depth_init.py:10-67`` on the HIP path.

For every segment: the mean or (lower) median over its pixels with a valid estimate of
``log(est) - logdepth_perseg``, plus the segment's base log-depth at its keypoint; segments that see no valid
pixel receive the (lower) median of the visible segments' values.  The reference loops over segments in Python
with boolean indexing (N host syncs); here it is one workgroup per segment (compaction + radix select) and a
single tiny launch for the invisible ones (``sp_segment_reinit``)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..segment_table import table_of


def segment_based_depth_reinit(estimated_depth, kf, mode='mean', return_info=False):
    assert mode == 'mean' or mode == 'median'
    lib = _lib.load()
    dev = kf.logdepth_perseg.device
    _lib.require_device(kf.logdepth_perseg)
    if isinstance(estimated_depth, np.ndarray):
        estimated_depth = torch.from_numpy(estimated_depth)
    est = estimated_depth.detach().to(dev).contiguous().float()
    table = table_of(kf)
    scratch = torch.empty(table.P, dtype=torch.float32, device=dev)
    out = torch.empty(table.N, dtype=torch.float32, device=dev)
    visible = torch.empty(table.N, dtype=torch.bool, device=dev)
    rc = lib.sp_segment_reinit(_lib.ptr(table.pix), _lib.ptr(table.baseL), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L),
                               table.N, table.P, table.H, table.W, _lib.ptr(est), 0 if mode == 'mean' else 1,
                               _lib.ptr(scratch), _lib.ptr(out), _lib.ptr(visible), _lib.stream_ptr())
    _lib.check(rc, "sp_segment_reinit")
    # the reference clamps invalid estimates to eps IN PLACE on the tensor it is given (depth_init.py:27-30);
    # callers pass a clone, so that side effect is not reproduced.
    return (out, visible) if return_info else out
