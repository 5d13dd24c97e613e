"""Segment-based depth completion (VOID) -- the reference's ``depth_completion/segment_based_completion.py`` API.

Pipeline per image (``:30-92``): sparse-depth pixels become keypoints -> the frontend turns (image, K, keypoints)
into a KeyFrame (SAM masks + integrated normals; OUT OF SCOPE here, any object with ``process_to_kf`` can be
plugged in) -> per-segment median log-depth shift (``sp_segment_reinit``) -> per-pixel average over the
covering visible segments (``sp_depth_average``, which fuses the dense ``unproject_kf_to_depths`` expansion, the
``depths[mask == 0] = -1`` masking, the visible-segment filter and ``render_depth_avg``) -> rerun with larger
masks when more than 15 % of the pixels stay uncovered."""
from __future__ import annotations

import copy

import torch
import yaml

from .. import _lib
from ..odometery import depth_init
from ..segment_table import table_of
from ..tool import point_utils
from ..tool.etc import to_np


def render_depth_avg(depths):
    """(N,H,W) stack with non-covering entries < 1e-6 -> per-pixel mean of the valid ones and the invalid map
    (segment_based_completion.py:21-27).  Dense-stack form kept for API parity; ``infer_depth`` uses the fused
    table kernel instead and never materialises the stack."""
    invalid = depths.max(dim=0)[0] < 1e-6
    depths[depths < 1e-6] = 0.0
    count = (depths > 1e-6).sum(dim=0) + 1e-6
    return depths.sum(dim=0) / count, invalid


def average_visible_segments(kf, keypoints_logdepth, visible, reduce=None, empty=None):
    """Fused HIP form of: unproject_kf_to_depths -> mask -> keep visible -> render_depth_avg.

    ``reduce``: optional callable(sums int64 (H*W,), counts int32 (H*W,)) run between the accumulation and the final
    division -- the hook of segment-sharded completion (``dist.complete_depth_sharded`` all-reduces the two integer arrays
    there; integer sums are exact, so the sharded result is bitwise the single-GPU one).
    ``kf`` = None with ``empty`` = (H, W, device): a rank that owns NO segment of a sharded image -- zero accumulators, the same
    ``reduce`` call as every other rank, the same division."""
    lib = _lib.load()
    if kf is None:
        H, W, dev = empty
        acc = torch.zeros(3 * H * W, dtype=torch.int32, device=dev)
    else:
        table = table_of(kf)
        dev = table.device
        H, W = table.H, table.W
        acc = torch.empty(3 * H * W, dtype=torch.int32, device=dev)
    depth = torch.empty(H, W, dtype=torch.float32, device=dev)
    invalid = torch.empty(H, W, dtype=torch.bool, device=dev)
    if kf is not None:
        vis = None if visible is None else visible.to(torch.bool).contiguous()
        rc = lib.sp_depth_accumulate(_lib.ptr(table.pix), _lib.ptr(table.baseL), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L),
                                     _lib.ptr(keypoints_logdepth.detach().contiguous().float()), _lib.ptr(vis), table.N, table.P,
                                     H, W, _lib.ptr(acc), _lib.stream_ptr())
        _lib.check(rc, "sp_depth_accumulate")
    if reduce is not None:
        reduce(acc[: 2 * H * W].view(torch.int64), acc[2 * H * W:])
    _lib.check(lib.sp_depth_average_finish(_lib.ptr(acc), H, W, _lib.ptr(depth), _lib.ptr(invalid), _lib.stream_ptr()),
               "sp_depth_average_finish")
    return depth, invalid


def infer_depth(front_processor, image, keypoints, K, partial_depth, rerun=False):
    orig_config = copy.deepcopy(front_processor.config)
    if rerun:   # fall back to segmenting larger regions (segment_based_completion.py:32-35)
        front_processor.config['sam_params']['nms'] = False
        front_processor.config['sam_params']['select_smallest'] = False
    kf = front_processor.process_to_kf(image, K, keypoints=keypoints)
    if rerun:
        front_processor.config = orig_config
    partial_depth = partial_depth.to(kf.image.device)
    kld, visible = depth_init.segment_based_depth_reinit(partial_depth.clone().detach(), kf, mode='median', return_info=True)
    return average_visible_segments(kf, kld, visible)


class DepthCompletion:
    def __init__(self, config_path=None, front_processor=None, config=None):
        """``front_processor``: the (out-of-scope) segmentation frontend; must offer ``.config`` and
        ``.process_to_kf(image, K, keypoints=...) -> KeyFrame``.  With only ``config_path`` given the reference
        would build its SAM/normals frontend here, which this package does not ship."""
        if config is None and config_path is not None:
            config = yaml.load(open(config_path, 'r'), Loader=yaml.FullLoader)
        self.config = config
        if front_processor is None:
            raise RuntimeError("DepthCompletion needs a front_processor: the SAM / normal-integration frontend is out of "
                               "scope of this package (SURVEY.md §2); pass any object with process_to_kf().")
        self.front_processor = front_processor

    def depth_completion(self, image, K, partial_depth, device='cuda:0'):
        rows, cols = torch.where(partial_depth > 1e-6)
        keypoints = torch.stack([rows, cols], dim=1).float()
        H, W = partial_depth.shape
        keypoints = point_utils.normalise_coordinates(keypoints, (H, W)).to(device)
        depths, invalid = infer_depth(self.front_processor, image, keypoints, K, partial_depth)
        ratio = invalid.sum().float() / (invalid.shape[0] * invalid.shape[1])
        if ratio > 0.15:
            print('high invalid depth ration, reruning with larger masks')
            depths_new, invalid_new = infer_depth(self.front_processor, image, keypoints, K, partial_depth, rerun=True)
            depths[invalid] = depths_new[invalid]
            invalid = torch.logical_and(invalid, invalid_new)
        return to_np(depths), to_np(invalid)
