"""One source keyframe against a batch of B target frames -- the reference's ``core/dense_optim_batch.py`` API.

Used by windowed mapping (``odometery/odometery.py:833-839``): ``trg_images (B,3,H,W)``, ``trg_Ks (B,3,3)``,
``poses (B,4,4)``; the source side is evaluated once and shared.  On the HIP path that is a single launch with
grid = (tiles, B): the source table is streamed B times through L2, each target image once.
"""
from __future__ import annotations

import torch

from .. import _lib
from ..segment_table import packed_target, table_of
from . import dense_optim as _do
from .dense_optim import Z_MIN_BATCH, _FusedPhotoCost, _check_mode, _f32c, _run_stats
from .ops import project_points, project_points_batch, transform_points_batch  # noqa: F401  (API parity)
from ..tool import point_utils


def get_pixels_batch(image, points_3d, K, spatial_dim=None):
    """Explicit-point sampling helper (core/dense_optim_batch.py:12-46); validity uses z > 1e-6 here."""
    in_front = points_3d[..., 2].detach() > Z_MIN_BATCH
    uv = project_points(points_3d, K) if points_3d.dim() == 2 else project_points_batch(points_3d, K)
    if spatial_dim is None:
        spatial_dim = image.shape[1:]
    unit = point_utils.normalise_coordinates(uv.flip(-1), spatial_dim).flip(-1)
    if image.dim() == 3:
        image = image[None]
    if unit.dim() == 2:
        unit = unit[None]
    vals, inside = _do.img_interp(image, unit)
    return vals, inside & in_front


def photomeric_cost_batch(src_keyframe, trg_images, trg_Ks, src_keypoint_logdepth, poses, cost_config, affine_comp=None):
    """``residual`` (B,) with autograd to kld, poses (B,4,4) and the affine pairs (src (2,), trg (B,2))
    (core/dense_optim_batch.py:50-147)."""
    _check_mode(cost_config)
    collect_stats = cost_config['collect_stats']
    _lib.require_device(src_keyframe.image, trg_images, trg_Ks, src_keypoint_logdepth, poses)
    _do._debug_finite(src_keypoint_logdepth, "keypoint log-depth")
    B = poses.shape[0]
    assert trg_images.shape[0] == B and trg_Ks.shape[0] == B
    table = table_of(src_keyframe)
    src4 = table.source_level(src_keyframe.image, src_keyframe.K, src_keypoint_logdepth)
    trg4 = packed_target(trg_images)
    K_src, K_trg = _f32c(src_keyframe.K), _f32c(trg_Ks)
    aff_s = aff_t = None
    if affine_comp is not None:
        aff_s, aff_t = affine_comp
    residual = _FusedPhotoCost.apply(src_keypoint_logdepth, poses, aff_s, aff_t, table, src4, trg4, K_src, K_trg, Z_MIN_BATCH)
    if collect_stats <= 0:
        return {'residual': residual}
    kld_s, poses_s = _do._snap(src_keypoint_logdepth), _do._snap(poses)
    aff = None if aff_s is None else (_do._snap(aff_s), _do._snap(aff_t))
    stride = cost_config.get('stats_stride', 1)

    def producer():
        st = _run_stats(table, src4, kld_s, K_src, trg4, K_trg, poses_s, aff, Z_MIN_BATCH, stride=stride)
        full = (st['trg_valid'] & st['src_valid'])[:, None].long()
        out = dict(segm_ids=st['seg_ids'], src_pixels=st['src_rgb'], src_in_trg_pixels=st['trg_rgb'],
                   src_valid_mask=st['src_valid'], trg_valid_mask=st['trg_valid'], full_mask=full,
                   src_pts=st['src_pts'], src_in_trg_pts=st['trg_pts'], residual_raw=st['raw'], median_depth=None)
        if collect_stats > 1:
            with torch.no_grad():
                H, W = src_keyframe.geo_spatial_dim()
                kp_cr = point_utils.denormalise_coordinates(src_keyframe.keypoints, (H, W)).flip(-1)
                kp3 = _do.unproject_points(kp_cr, torch.exp(kld_s), src_keyframe.K)
                kp3 = transform_points_batch(kp3, poses_s)
                _, ok = get_pixels_batch(trg_images, kp3, trg_Ks, spatial_dim=(H, W))
                out.update(src_in_trg_keypoints=project_points_batch(kp3, trg_Ks), src_in_trg_keypoints_z=kp3[..., 2],
                           src_in_trg_keypoints_valid_mask=ok)
        return out

    return _do._with_stats(residual, cost_config, producer)
