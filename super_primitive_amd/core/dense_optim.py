"""Single-target photometric cost -- the reference's ``core/dense_optim.py`` API on the MI355X HIP path.

Same function names, argument meaning, returned-dict keys and error behaviour as the reference module, so that
``odometery/two_frame_sfm.py`` / ``odometery/odometery.py`` / ``depth_completion/segment_based_completion.py``
style drivers call it unchanged.  What differs is *how*: instead of ~150 ATen launches over dense (N,H,W)
tensors plus an autograd backward, one call is

    sp_photo_cost_grad  (fused cost + analytic gradient over the compact segment table)   [+ sp_photo_stats]

wrapped in a ``torch.autograd.Function`` so ``loss.backward()`` still fills ``.grad`` of the log-depths, of the
pose's upstream graph and of the affine pairs (SURVEY.md §8(b) gradient contract).

There is no CPU implementation here: host tensors raise (``_lib.require_device``).
"""
from __future__ import annotations

import os

import torch

from .. import _lib
from .. import segment_table
from ..segment_table import packed_target, table_of
from ..tool import point_utils
from .cost_utils import split_by_mode
from .normal_cost import transform_normals, transform_normals_batch  # noqa: F401  (API parity)
from .ops import project_points  # noqa: F401  (re-exported: tool/viz.py:71,121)

Z_MIN_SINGLE = 1e-7   # core/dense_optim.py:146
Z_MIN_BATCH = 1e-6    # core/dense_optim_batch.py:15
_DEBUG_FINITE = os.environ.get("SP_DEBUG_FINITE", "0") not in ("", "0")


# ---------------------------------------------------------------------------------------------------------
# fused cost + gradient as an autograd node
# ---------------------------------------------------------------------------------------------------------
class _FusedPhotoCost(torch.autograd.Function):
    """residual (B,) = f(kld (N), poses (B,4,4), aff_src (2)|None, aff_trg (B,2)|None); everything else constant."""

    @staticmethod
    def forward(ctx, kld, poses, aff_src, aff_trg, table, src4, trg4, K_src, K_trg, zmin):
        lib = _lib.load()
        B = poses.shape[0]
        dev = poses.device
        Hl, Wl = trg4.shape[1], trg4.shape[2]
        kld_c = kld.detach().contiguous().float()
        poses_c = poses.detach().contiguous().float()
        has_aff = aff_src is not None
        a_s = aff_src.detach().contiguous().float() if has_aff else None
        a_t = aff_trg.detach().reshape(B, 2).contiguous().float() if has_aff else None
        work = torch.empty(B * table.n_tiles * _lib.SP_GRAD_PARTIAL_FLOATS, dtype=torch.float32, device=dev)
        residual = torch.empty(B, dtype=torch.float32, device=dev)
        g_kld = torch.empty(B, table.N, dtype=torch.float32, device=dev)
        g_pose = torch.empty(B, 4, 4, dtype=torch.float32, device=dev)
        g_aff = torch.empty(B, 4, dtype=torch.float32, device=dev)
        rc = lib.sp_photo_cost_grad(
            _lib.ptr(table.pix), _lib.ptr(src4), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L), _lib.ptr(table.tiles),
            _lib.ptr(table.seg_tile_off), table.n_tiles, table.N, table.P, table.H, table.W, _lib.ptr(K_src),
            _lib.ptr(kld_c), _lib.ptr(trg4), Hl, Wl, _lib.ptr(K_trg), _lib.ptr(poses_c), B, _lib.ptr(a_s), _lib.ptr(a_t),
            float(zmin), _lib.ptr(work), _lib.ptr(residual), _lib.ptr(g_kld), _lib.ptr(g_pose), _lib.ptr(g_aff),
            _lib.stream_ptr())
        _lib.check(rc, "sp_photo_cost_grad")
        ctx.save_for_backward(g_kld, g_pose, g_aff)
        ctx.has_aff = has_aff
        ctx.aff_trg_shape = None if not has_aff else tuple(aff_trg.shape)
        return residual

    @staticmethod
    def backward(ctx, grad_out):
        g_kld, g_pose, g_aff = ctx.saved_tensors
        w = grad_out.reshape(-1)
        d_kld = (w[:, None] * g_kld).sum(0)
        d_pose = w[:, None, None] * g_pose
        d_as = d_at = None
        if ctx.has_aff:
            d_as = (w[:, None] * g_aff[:, :2]).sum(0)
            d_at = (w[:, None] * g_aff[:, 2:]).reshape(ctx.aff_trg_shape)
        return d_kld, d_pose, d_as, d_at, None, None, None, None, None, None


def _check_mode(cost_config):
    mode = cost_config['mode']
    if mode != 'colour':
        # The reference never evaluates a normal/kappa residual (residual_cosine stays 0.0,
        # core/dense_optim.py:241-261) and no caller or config selects such a mode.
        raise NotImplementedError(f"residual mode {mode!r}: only 'colour' is on the hot path (see DESIGN.md, out of scope)")
    return mode


def _f32c(t):
    return t.detach().contiguous().float()


def _debug_finite(t, what):
    """The reference asserts finiteness with a host sync on every call (core/dense_optim.py:44,78,311,321,340-343).
    Those syncs are exactly what the fused path removes; the check survives as an opt-in (SP_DEBUG_FINITE=1) and
    raises the same AssertionError."""
    if _DEBUG_FINITE and not bool(torch.isfinite(t).all()):
        raise AssertionError(f"non-finite {what}")


def _run_stats(table, src4, kld, K_src, trg4, K_trg, poses, aff, zmin, want_target=True, stride=1):
    """Per-point diagnostics via sp_photo_stats; returns a dict of raw device tensors (every ``stride``-th table point)."""
    lib = _lib.load()
    dev = table.device
    stride = max(1, int(stride))
    P = (table.P + stride - 1) // stride
    B = 1 if poses is None else poses.shape[0]
    out = dict(src_pts=torch.empty(P, 3, dtype=torch.float32, device=dev),
               src_rgb=torch.empty(1, 3, P, dtype=torch.float32, device=dev),
               src_valid=torch.empty(1, P, dtype=torch.bool, device=dev),
               seg_ids=torch.empty(P, dtype=torch.int64, device=dev))
    trg_pts = trg_rgb = raw = trg_valid = None
    Hl = Wl = 1
    if want_target:
        Hl, Wl = trg4.shape[1], trg4.shape[2]
        trg_pts = torch.empty(B, P, 3, dtype=torch.float32, device=dev)
        trg_rgb = torch.empty(B, 3, P, dtype=torch.float32, device=dev)
        raw = torch.empty(B, 3, P, dtype=torch.float32, device=dev)
        trg_valid = torch.empty(B, P, dtype=torch.bool, device=dev)
        out.update(trg_pts=trg_pts, trg_rgb=trg_rgb, raw=raw, trg_valid=trg_valid)
    a_s = a_t = None
    if aff is not None and want_target:
        a_s, a_t = _f32c(aff[0]), _f32c(aff[1]).reshape(B, 2).contiguous()
    rc = lib.sp_photo_stats(
        _lib.ptr(table.pix), _lib.ptr(src4), _lib.ptr(table.seg_off), _lib.ptr(table.kp_L), table.N, table.P, table.H, table.W,
        _lib.ptr(K_src), _lib.ptr(_f32c(kld)), _lib.ptr(trg4) if want_target else None, Hl, Wl,
        _lib.ptr(K_trg) if want_target else None, _lib.ptr(_f32c(poses)) if want_target else None, B, _lib.ptr(a_s),
        _lib.ptr(a_t), float(zmin), _lib.ptr(out['src_pts']), _lib.ptr(trg_pts), _lib.ptr(out['src_rgb']), _lib.ptr(trg_rgb),
        _lib.ptr(raw), _lib.ptr(out['src_valid']), _lib.ptr(trg_valid), _lib.ptr(out['seg_ids']), stride, _lib.stream_ptr())
    _lib.check(rc, "sp_photo_stats")
    return out


class LazyStats(dict):
    """The dict ``photomeric_cost*`` returns when ``collect_stats > 0``.

    ``'residual'`` is there from the start; the per-point diagnostic tensors (``src_pts``, ``residual_raw``, ... --
    26 MB per call at 640x480x64) are produced by ONE extra launch the first time anything else is looked at
    (``d['src_pts']``, ``d.items()``, ``dict_cpu(d)`` ...).  A driver that configures ``collect_stats`` like the
    reference's (always 2) but whose visualiser is not attached therefore never pays for them.  The inputs that
    define the diagnostics (log-depths, poses, affine pairs) are snapshotted at call time, so looking later -- after
    ``optimizer.step()`` has moved the parameters in place -- still shows the state the residual was computed at.
    ``cost_config['stats_stride'] = s`` reports every s-th table point only (the down-sampled channel for a GUI);
    ``cost_config['stats_lazy'] = False`` restores eager evaluation."""

    def __init__(self, residual, producer):
        super().__init__(residual=residual)
        self._producer = producer

    def materialise(self):
        producer, self._producer = self._producer, None
        if producer is not None:
            super().update(producer())
        return self

    def __getitem__(self, key):
        if key != 'residual':
            self.materialise()
        return super().__getitem__(key)

    def get(self, key, default=None):
        if key != 'residual':
            self.materialise()
        return super().get(key, default)

    def __contains__(self, key):
        if key != 'residual':
            self.materialise()
        return super().__contains__(key)

    def __iter__(self):
        return iter(self.materialise().keys())

    def __len__(self):
        self.materialise()
        return super().__len__()

    def keys(self):
        self.materialise()
        return super().keys()

    def items(self):
        self.materialise()
        return super().items()

    def values(self):
        self.materialise()
        return super().values()

    def copy(self):
        return dict(self.materialise())

    def __repr__(self):
        return dict.__repr__(self.materialise())

    def __reduce__(self):               # pickles (multiprocessing queues) as the plain dict it stands for
        return (dict, (dict(self.materialise()),))


def _snap(t):
    return None if t is None else t.detach().clone()


def _with_stats(residual, cost_config, producer):
    if cost_config.get('stats_lazy', True):
        return LazyStats(residual, producer)
    out = {'residual': residual}
    out.update(producer())
    return out


# ---------------------------------------------------------------------------------------------------------
# reference API
# ---------------------------------------------------------------------------------------------------------
def infer_spatial_size(logdepth_perseg):
    if logdepth_perseg.dim() == 3:
        return logdepth_perseg.shape[1:], True
    assert logdepth_perseg.dim() == 2
    return logdepth_perseg.shape, False


def photomeric_cost(src_keyframe, trg_keyframe, src_keypoint_logdepth, pose, cost_config, affine_comp=None):
    """Photometric cost of one source keyframe against one target frame (core/dense_optim.py:265-363).

    Returns ``{'residual': (1,)}`` connected to autograd for ``src_keypoint_logdepth``, ``pose`` and the affine
    pairs; with ``cost_config['collect_stats'] > 0`` also the reference's per-point diagnostic tensors."""
    _check_mode(cost_config)
    collect_stats = cost_config['collect_stats']
    _lib.require_device(src_keyframe.image, trg_keyframe.image, src_keypoint_logdepth, pose)
    _debug_finite(src_keypoint_logdepth, "keypoint log-depth")
    table = table_of(src_keyframe)
    src4 = table.source_level(src_keyframe.image, src_keyframe.K, src_keypoint_logdepth)
    trg4 = packed_target(trg_keyframe.image)
    K_src, K_trg = _f32c(src_keyframe.K), _f32c(trg_keyframe.K)[None].contiguous()
    aff_s = aff_t = None
    if affine_comp is not None:
        aff_s, aff_t = affine_comp
        if aff_s is None:
            assert aff_t is None
    residual = _FusedPhotoCost.apply(src_keypoint_logdepth, pose[None], aff_s, aff_t, table, src4, trg4, K_src, K_trg,
                                     Z_MIN_SINGLE)
    if collect_stats <= 0:
        return {'residual': residual}
    kld_s, pose_s, aff = _snap(src_keypoint_logdepth), _snap(pose), None if aff_s is None else (_snap(aff_s), _snap(aff_t))
    stride = cost_config.get('stats_stride', 1)

    def producer():
        st = _run_stats(table, src4, kld_s, K_src, trg4, K_trg, pose_s[None], aff, Z_MIN_SINGLE, stride=stride)
        full = (st['trg_valid'] & st['src_valid'])[:, None].long()
        out = dict(segm_ids=st['seg_ids'], src_pixels=st['src_rgb'], src_in_trg_pixels=st['trg_rgb'],
                   src_valid_mask=st['src_valid'], trg_valid_mask=st['trg_valid'], full_mask=full,
                   src_pts=st['src_pts'], src_in_trg_pts=st['trg_pts'][0], residual_raw=st['raw'], median_depth=None)
        if collect_stats > 1:
            out.update(_keypoint_stats(src_keyframe, trg_keyframe, kld_s, pose_s))
        return out

    return _with_stats(residual, cost_config, producer)


def _keypoint_stats(src_kf, trg_kf, kld, pose):
    """The N keypoints pushed through the same warp (dense_optim.py:291-308); N points of diagnostics."""
    with torch.no_grad():
        H, W = src_kf.geo_spatial_dim()
        kp_cr = point_utils.denormalise_coordinates(src_kf.keypoints, (H, W)).flip(-1)
        pts = transform_points(unproject_points(kp_cr, torch.exp(kld.detach()), src_kf.K), pose.detach())
        _, ok = get_pixels(trg_kf.image, pts, trg_kf.K, spatial_dim=(H, W))
        return dict(src_in_trg_keypoints=project_points(pts, trg_kf.K_img), src_in_trg_keypoints_z=pts[:, 2],
                    src_in_trg_keypoints_valid_mask=ok)


def unproject_kf(kf, keypoint_logdepth, jacobian=False):
    """Source points / colours / validity of a keyframe (core/dense_optim.py:176-200); feeds the tracking loop.
    The returned dict is exactly the reference's: plain tensors, nothing hidden -- it may be filtered, sent through
    ``dict_cpu`` / a queue / a checkpoint and handed back to ``photomeric_cost_precomputed``."""
    _lib.require_device(kf.image, keypoint_logdepth)
    table = table_of(kf)
    src4 = table.source_level(kf.image, kf.K, keypoint_logdepth)
    st = _run_stats(table, src4, keypoint_logdepth, _f32c(kf.K), None, None, None, None, Z_MIN_SINGLE, want_target=False)
    return {'src_pixels': st['src_rgb'], 'src_valid_mask': st['src_valid'], 'src_pts': st['src_pts'],
            'segm_ids': st['seg_ids'], 'spatial_size': kf.geo_spatial_dim()}


class _PointList:
    """Compact 24 B/point form of a precomputed dict: xyz (n,3) and rgb (n,3) of the points whose ``src_valid_mask``
    is set (an invalid source point contributes exact zeros in the reference, core/dense_optim.py:389-396), plus the
    original point count -- the residual stays a mean over 3 * P_total values."""

    def __init__(self, pre):
        pts, rgb, ok = pre['src_pts'], pre['src_pixels'], pre['src_valid_mask']
        _lib.require_device(pts, rgb, ok)
        self.P_total = int(pts.shape[0])
        assert rgb.shape[-1] == self.P_total and ok.shape[-1] == self.P_total
        keep = ok.reshape(-1).bool().nonzero().reshape(-1)          # the one host sync of the conversion (sizes the list)
        self.xyz = pts.detach().float()[keep].contiguous()
        self.rgb = rgb.detach().float().reshape(-1, self.P_total)[:3].t()[keep].contiguous()
        self.n = int(keep.numel())
        self.H, self.W = (int(v) for v in pre['spatial_size'])
        self._key = tuple(segment_table._Ident(t) for t in (pts, rgb, ok))

    def matches(self, pre):
        return segment_table._same(self._key, (pre['src_pts'], pre['src_pixels'], pre['src_valid_mask']))


_point_lists = []          # most recent last; entries hold strong references to the dict's tensors (identity keys)


def _point_list_of(pre):
    for pl in reversed(_point_lists):
        if pl.matches(pre):
            return pl
    pl = _PointList(pre)
    if len(_point_lists) >= 8:
        del _point_lists[0]
    _point_lists.append(pl)
    return pl


class _FusedPointCost(torch.autograd.Function):
    """residual (1,) = f(pose (1,4,4), aff_src (2)|None, aff_trg (1,2)|None) over an explicit point list."""

    @staticmethod
    def forward(ctx, poses, aff_src, aff_trg, pl, trg4, K_trg, zmin):
        lib = _lib.load()
        B, dev = poses.shape[0], poses.device
        Hl, Wl = trg4.shape[1], trg4.shape[2]
        has_aff = aff_src is not None
        a_s = aff_src.detach().contiguous().float() if has_aff else None
        a_t = aff_trg.detach().reshape(B, 2).contiguous().float() if has_aff else None
        residual = torch.zeros(B, dtype=torch.float32, device=dev)
        g_pose = torch.zeros(B, 4, 4, dtype=torch.float32, device=dev)
        g_aff = torch.zeros(B, 4, dtype=torch.float32, device=dev)
        if pl.n > 0:                     # (no valid source point at all: residual 0, like the reference's masked mean)
            work = torch.empty(lib.sp_points_workspace_floats(pl.n, B), dtype=torch.float32, device=dev)
            rc = lib.sp_points_cost_grad(_lib.ptr(pl.xyz), _lib.ptr(pl.rgb), pl.n, pl.P_total, pl.H, pl.W, _lib.ptr(trg4), Hl, Wl,
                                         _lib.ptr(K_trg), _lib.ptr(poses.detach().contiguous().float()), B, _lib.ptr(a_s),
                                         _lib.ptr(a_t), float(zmin), _lib.ptr(work), _lib.ptr(residual), _lib.ptr(g_pose),
                                         _lib.ptr(g_aff), _lib.stream_ptr())
            _lib.check(rc, "sp_points_cost_grad")
        ctx.save_for_backward(g_pose, g_aff)
        ctx.has_aff = has_aff
        ctx.aff_trg_shape = None if not has_aff else tuple(aff_trg.shape)
        return residual

    @staticmethod
    def backward(ctx, grad_out):
        g_pose, g_aff = ctx.saved_tensors
        w = grad_out.reshape(-1)
        d_as = d_at = None
        if ctx.has_aff:
            d_as = (w[:, None] * g_aff[:, :2]).sum(0)
            d_at = (w[:, None] * g_aff[:, 2:]).reshape(ctx.aff_trg_shape)
        return w[:, None, None] * g_pose, d_as, d_at, None, None, None, None


def photomeric_cost_precomputed(src_precomputed, trg_keyframe, pose, cost_config, affine_comp=None):
    """Tracking variant: source side fixed, gradient to pose and affine only (core/dense_optim.py:365-403).

    ``src_precomputed`` is any reference-shaped dict ``{src_pts (P,3), src_pixels (1,3,P), src_valid_mask (1,P),
    segm_ids, spatial_size}`` -- the one ``unproject_kf`` returns, a hand-built or filtered one, or one that went
    through ``dict_cpu`` and back to the device.  Its points are compacted ONCE per dict (identity-cached) into a
    24 B/point list; every call is then one fused cost + gradient launch over that list."""
    _check_mode(cost_config)
    pl = _point_list_of(src_precomputed)
    _lib.require_device(trg_keyframe.image, pose, pl.xyz)
    trg4 = packed_target(trg_keyframe.image)
    K_trg = _f32c(trg_keyframe.K)[None].contiguous()
    aff_s = aff_t = None
    if affine_comp is not None:
        aff_s, aff_t = affine_comp
    residual = _FusedPointCost.apply(pose[None], aff_s, aff_t, pl, trg4, K_trg, Z_MIN_SINGLE)
    return {'residual': residual}


def unproject_kf_to_depths(kf, keypoint_logdepth):
    """Dense (N,H,W) per-segment depths exp((L + shift_n) * mask) (core/dense_optim.py:164-174)."""
    return _dense_depths(keypoint_logdepth, kf.keypoints, kf.keypoint_regions, kf.get_logdepth(), log_space=False)


def infer_depth_seeds(keypoint_logdepth, keypoints, keypoint_regions, logdepth_perseg):
    """Dense seeded log-depths (L + (kld_n - L[n,kp])) * mask (core/dense_optim.py:38-80)."""
    return _dense_depths(keypoint_logdepth, keypoints, keypoint_regions, logdepth_perseg, log_space=True)


def expdepth(logdepth):
    return torch.exp(logdepth)


def _dense_depths(kld, keypoints, masks, logdepth, log_space):
    _lib.require_device(kld, keypoints, masks, logdepth)
    _debug_finite(kld, "keypoint log-depth")
    if logdepth.dim() == 2:
        logdepth = logdepth[None].expand(keypoints.shape[0], -1, -1)
    N, H, W = masks.shape
    out = torch.empty(N, H, W, dtype=torch.float32, device=masks.device)
    lib = _lib.load()
    rc = lib.sp_depth_expand(_lib.ptr(masks.contiguous()), _lib.ptr(_f32c(logdepth)), _lib.ptr(_f32c(keypoints)),
                             _lib.ptr(_f32c(kld)), N, H, W, int(log_space), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "sp_depth_expand")
    return out


# ---- explicit-point helpers kept for API parity (visualisers); NOT used by the cost functions above -------
def unproject_points(points_2d, depth_2d, K):
    z = depth_2d.reshape(-1)
    assert points_2d.shape[0] == z.shape[0]
    x = (points_2d[:, 0].reshape(-1).float() - K[0, 2]) * z / K[0, 0]
    y = (points_2d[:, 1].reshape(-1).float() - K[1, 2]) * z / K[1, 1]
    return torch.stack((x, y, z), dim=1)


def unproject_segments(segment_depths, segment_masks, K, include_coords=False):
    seg, row, col = torch.where(segment_masks)
    coords = torch.stack((col, row), dim=1)
    pts = unproject_points(coords, segment_depths[seg, row, col], K)
    return (pts, seg, coords) if include_coords else (pts, seg)


def transform_points(points_3d, pose):
    return points_3d @ pose[:3, :3].T + pose[:3, 3]


def img_interp(img, coords_norm, mode="bilinear"):
    inside = (coords_norm.abs() <= 0.99).all(dim=-1)
    vals = torch.nn.functional.grid_sample(img, coords_norm[:, None], mode=mode, padding_mode='zeros', align_corners=True)
    return vals[:, :, 0], inside


def get_pixels(image, points_3d, K, spatial_dim=None, mode='bilinear'):
    in_front = points_3d[..., 2].detach() > Z_MIN_SINGLE
    if spatial_dim is None:
        spatial_dim = image.shape[1:]
    unit = point_utils.normalise_coordinates(project_points(points_3d, K), (spatial_dim[1], spatial_dim[0]))
    vals, inside = img_interp(image[None], unit[None], mode=mode)
    return vals, inside & in_front


def affine_compensation_batch_v2(trg_pixels, src_affine_comp, trg_affine_comp):
    if src_affine_comp is None:
        assert trg_affine_comp is None
        return trg_pixels
    s = src_affine_comp[None] if src_affine_comp.dim() == 1 else src_affine_comp
    t = trg_affine_comp[None] if trg_affine_comp.dim() == 1 else trg_affine_comp
    gain = torch.exp(-(t[:, 0] - s[:, 0]))[:, None, None]
    bias = (t[:, 1] - s[:, 1])[:, None, None]
    return torch.cat((gain * trg_pixels[:, :3] + bias, trg_pixels[:, 3:]), dim=1)


def calculate_residual(src_pixels, trg_pixels, validity_mask, cost_conifg, return_raw=False, src_depthes=None):
    mode = cost_conifg['mode']
    src_rgb, _, _ = split_by_mode(src_pixels, mode=mode)
    trg_rgb, _, _ = split_by_mode(trg_pixels, mode=mode)
    diff = (src_rgb - trg_rgb) * validity_mask
    raw = diff.detach().clone() if return_raw else None
    return diff.abs().mean(dim=[1, 2]), {'residual_raw': raw, 'median_depth': None}
