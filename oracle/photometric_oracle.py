"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (PyTorch, dense layout, autograd) of the reference's
per-segment photometric cost and the helpers around it.  It follows the
reference's *algorithm* step by step -- dense (N,H,W) depth seeding, mask
multiply, exp, nonzero gather, unproject, rigid transform, guarded projection,
align-corners normalisation, ``grid_sample`` (bilinear / zeros /
align_corners=True), affine brightness, masked L1 mean -- so that

  * its outputs pin the HIP kernels (tests/, ``__graft_entry__.smoke``), and
  * timing it on the GPU box's host cores is the "reference CPU PyTorch path"
    baseline (``bench.py`` ``cpu_baseline``, kind "port"): the reference's own
    Python files cannot travel to the GPU box.

Pinned against the real reference: ``oracle/gen_goldens.py`` imports
``/root/reference`` in the build container, runs it on seeded inputs and stores
inputs+outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
replays this file against those vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.

Reference lines restated (paths relative to /root/reference):
  core/dense_optim.py:19-35 (unproject), :38-80 (depth seeds), :83-86 (exp),
  :89-114 (segment gather), :117-122 (rigid), :128-162 (sampling + validity),
  :164-200 (keyframe unprojection), :202-225 (affine), :228-261 (residual),
  :265-403 (cost, precomputed cost); core/dense_optim_batch.py:12-147;
  core/ops.py:5-40,59-96; core/depth_render.py:7-21;
  image/gaussian_pyramid.py:42-118; image/keyframe.py:77-148;
  lie/lie_algebra.py:11-119,129-137,191-197,223-258;
  tool/point_utils.py:31-40; odometery/depth_init.py:10-67;
  depth_completion/segment_based_completion.py:21-27.
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# data model
# ----------------------------------------------------------------------------
class OracleFrame:
    """Attribute bag with the reference KeyFrame's field names (image/keyframe.py:20-75)."""

    def __init__(self, image, K, logdepth_perseg=None, keypoints=None, keypoint_regions=None, K_img=None, id=None):
        self.image = image
        self.K = K
        self.K_img = K if K_img is None else K_img
        self.id = id
        self.supporting = logdepth_perseg is None or keypoints is None or keypoint_regions is None
        self.logdepth_perseg = None if self.supporting else logdepth_perseg
        self.keypoints = None if self.supporting else keypoints
        self.keypoint_regions = None if self.supporting else keypoint_regions

    def geo_hw(self):
        return tuple(self.logdepth_perseg.shape[-2:])


def frames_from_synth(pair, dtype=torch.float32):
    t = lambda a: torch.from_numpy(a).to(dtype)
    src = OracleFrame(t(pair.src_image), t(pair.K), t(pair.logdepth_perseg), t(pair.keypoints),
                      torch.from_numpy(pair.keypoint_regions))
    trg = OracleFrame(t(pair.trg_image), t(pair.K))
    return src, trg


# ----------------------------------------------------------------------------
# coordinate conventions (tool/point_utils.py:31-40)
# ----------------------------------------------------------------------------
def to_unit_range(px, dims):
    """pixel -> [-1,1] with the align-corners convention 2x/(d-1)-1."""
    scale = 1.0 / (torch.as_tensor(dims, dtype=px.dtype if px.dtype == torch.float64 else torch.float32, device=px.device) - 1)
    return 2 * px * scale - 1


def to_pixel_index(unit, dims):
    """[-1,1] -> integer pixel, torch.round = half-to-even."""
    d = torch.as_tensor(dims, dtype=torch.float32, device=unit.device)
    return (0.5 * (d - 1) * (unit + 1)).round().long()


# ----------------------------------------------------------------------------
# geometry (core/dense_optim.py:19-35,117-122 ; core/ops.py:5-40)
# ----------------------------------------------------------------------------
def backproject(col_row, z, K):
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    z = z.reshape(-1)
    x = (col_row[:, 0].reshape(-1).float() - cx) * z / fx
    y = (col_row[:, 1].reshape(-1).float() - cy) * z / fy
    return torch.stack((x, y, z), dim=1)


def rigid(points, pose):
    return points @ pose[:3, :3].T + pose[:3, 3]


def rigid_many(points, poses):
    R, t = poses[:, :3, :3], poses[:, :3, 3]
    eq = 'bij,nj->bni' if points.dim() == 2 else 'bij,bnj->bni'
    return torch.einsum(eq, R, points) + t[:, None, :]


def pinhole_many(points, Ks, eps=1e-6):
    """(B,P,3),(B,3,3) -> (B,P,2); 1/z replaced by eps where |z| <= eps (core/ops.py:19-40)."""
    fx, fy = Ks[..., 0, 0], Ks[..., 1, 1]
    cx, cy = Ks[..., 0, 2], Ks[..., 1, 2]
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    ok = z.abs() > eps
    zinv = torch.full_like(z, eps)
    zinv[ok] = 1.0 / z[ok]
    u = x * fx[:, None] * zinv + cx[:, None]
    v = y * fy[:, None] * zinv + cy[:, None]
    return torch.stack((u, v), dim=-1)


def pinhole(points, K):
    return pinhole_many(points[None], K[None])[0]


# ----------------------------------------------------------------------------
# depth seeding and segment gather (core/dense_optim.py:38-114)
# ----------------------------------------------------------------------------
def seed_logdepths(kld, frame):
    """Dense (N,H,W): per-segment base log-depth shifted so the keypoint hits kld[n]."""
    L = frame.logdepth_perseg
    N = frame.keypoints.shape[0]
    assert torch.isfinite(kld).all()
    rc = to_pixel_index(frame.keypoints, L.shape[-2:])
    at_kp = L[torch.arange(N, device=L.device), rc[:, 0], rc[:, 1]]
    shifted = L + (kld - at_kp)[:, None, None]
    shifted = shifted * frame.keypoint_regions
    assert torch.isfinite(shifted).all()
    return shifted


def gather_segments(depth, masks, K):
    """Masked pixels in (segment, row, col) order -> 3-D points (P,3), segment id (P), (col,row)."""
    seg, row, col = torch.where(masks)
    z = depth[seg, row, col]
    col_row = torch.stack((col, row), dim=1)
    return backproject(col_row, z, K), seg, col_row


# ----------------------------------------------------------------------------
# sampling (core/dense_optim.py:128-162 ; core/dense_optim_batch.py:12-46)
# ----------------------------------------------------------------------------
def _sample(images, unit_xy):
    """images (B,C,h,w), unit_xy (B,P,2) -> (B,C,P) and the 0.99 band mask (B,P)."""
    inside = (unit_xy.abs() <= 0.99).all(dim=-1)
    vals = F.grid_sample(images, unit_xy[:, None], mode='bilinear', padding_mode='zeros', align_corners=True)
    return vals[:, :, 0, :], inside


def sample_single(image, points, K, geo_hw, zmin=1e-7):
    """One image (C,h,w), points (P,3).  Validity: band & z > 1e-7 (dense_optim.py:146)."""
    front = points[..., 2].detach() > zmin
    uv = pinhole(points, K)
    unit = to_unit_range(uv, (geo_hw[1], geo_hw[0]))
    vals, inside = _sample(image[None], unit[None])
    return vals, inside & front


def sample_many(images, points, Ks, geo_hw, zmin=1e-6):
    """Batch variant: z > 1e-6 (dense_optim_batch.py:15)."""
    front = points[..., 2].detach() > zmin
    uv = pinhole(points, Ks) if points.dim() == 2 else pinhole_many(points, Ks)
    unit = to_unit_range(uv.flip(-1), geo_hw).flip(-1)
    if images.dim() == 3:
        images = images[None]
    if unit.dim() == 2:
        unit = unit[None]
    vals, inside = _sample(images, unit)
    return vals, inside & front


# ----------------------------------------------------------------------------
# brightness model + residual (core/dense_optim.py:202-261)
# ----------------------------------------------------------------------------
def brightness(trg_vals, aff_src, aff_trg):
    if aff_src is None:
        assert aff_trg is None
        return trg_vals
    if aff_src.dim() == 1:
        aff_src = aff_src[None]
    if aff_trg.dim() == 1:
        aff_trg = aff_trg[None]
    da = (aff_trg[:, 0:1] - aff_src[:, 0:1])[:, None].expand(-1, 3, -1)
    db = (aff_trg[:, 1:2] - aff_src[:, 1:2])[:, None].expand(-1, 3, -1)
    rgb = torch.exp(-da) * trg_vals[:, :3] + db
    return torch.cat((rgb, trg_vals[:, 3:]), dim=1)


def masked_l1(src_vals, trg_vals, mask, keep_raw):
    diff = (src_vals[:, :3] - trg_vals[:, :3]) * mask
    raw = diff.detach().clone() if keep_raw else None
    return diff.abs().mean(dim=[1, 2]), raw


# ----------------------------------------------------------------------------
# the cost functions (core/dense_optim.py:265-403 ; core/dense_optim_batch.py:50-147)
# ----------------------------------------------------------------------------
def photometric_cost(src, trg, kld, pose, collect_stats=0, affine=None):
    geo = src.geo_hw()
    depth = torch.exp(seed_logdepths(kld, src))
    pts, seg, _ = gather_segments(depth, src.keypoint_regions, src.K)
    out = {}
    extra = {}
    if collect_stats > 1:
        kp_cr = to_pixel_index(src.keypoints, depth.shape[1:]).flip(-1)
        kp3 = rigid(backproject(kp_cr, torch.exp(kld), src.K), pose)
        _, kp_ok = sample_single(trg.image, kp3, trg.K, geo)
        extra = dict(src_in_trg_keypoints=pinhole(kp3, trg.K_img), src_in_trg_keypoints_z=kp3[:, 2],
                     src_in_trg_keypoints_valid_mask=kp_ok)
    assert torch.isfinite(pts).all()
    moved = rigid(pts, pose)
    src_vals, src_ok = sample_single(src.image, pts, src.K, geo)
    assert torch.isfinite(moved).all()
    trg_vals, trg_ok = sample_single(trg.image, moved, trg.K, geo)
    mask = trg_ok[:, None].long() * src_ok[:, None].long()
    if affine is not None:
        trg_vals = brightness(trg_vals, affine[0], affine[1])
    res, raw = masked_l1(src_vals, trg_vals, mask, collect_stats > 0)
    out['residual'] = res
    if collect_stats > 0:
        out.update(segm_ids=seg, src_pixels=src_vals, src_in_trg_pixels=trg_vals, src_valid_mask=src_ok,
                   trg_valid_mask=trg_ok, full_mask=mask, src_pts=pts, src_in_trg_pts=moved,
                   residual_raw=raw, median_depth=None)
        if collect_stats > 1:
            out.update(extra)
    return out


def unproject_keyframe(frame, kld):
    geo = frame.geo_hw()
    depth = torch.exp(seed_logdepths(kld, frame))
    pts, seg, _ = gather_segments(depth, frame.keypoint_regions, frame.K)
    vals, ok = sample_single(frame.image, pts, frame.K, geo)
    return dict(src_pixels=vals, src_valid_mask=ok, src_pts=pts, segm_ids=seg, spatial_size=geo)


def keyframe_depths(frame, kld):
    return torch.exp(seed_logdepths(kld, frame))


def photometric_cost_precomputed(pre, trg, pose, affine=None):
    moved = rigid(pre['src_pts'], pose)
    trg_vals, trg_ok = sample_single(trg.image, moved, trg.K, pre['spatial_size'])
    mask = trg_ok[:, None].long() * pre['src_valid_mask'][:, None].long()
    if affine is not None:
        trg_vals = brightness(trg_vals, affine[0], affine[1])
    res, _ = masked_l1(pre['src_pixels'], trg_vals, mask, False)
    return {'residual': res}


def photometric_cost_batch(src, trg_images, trg_Ks, kld, poses, collect_stats=0, affine=None):
    geo = src.geo_hw()
    depth = torch.exp(seed_logdepths(kld, src))
    pts, seg, _ = gather_segments(depth, src.keypoint_regions, src.K)
    assert torch.isfinite(pts).all()
    moved = rigid_many(pts, poses)
    src_vals, src_ok = sample_single(src.image, pts, src.K, geo)
    trg_vals, trg_ok = sample_many(trg_images, moved, trg_Ks, geo)
    mask = trg_ok[:, None].long() * src_ok[:, None].long()
    if affine is not None:
        trg_vals = brightness(trg_vals, affine[0], affine[1])
    res, raw = masked_l1(src_vals, trg_vals, mask, collect_stats > 0)
    out = {'residual': res}
    if collect_stats > 0:
        out.update(segm_ids=seg, src_pixels=src_vals, src_in_trg_pixels=trg_vals, src_valid_mask=src_ok,
                   trg_valid_mask=trg_ok, full_mask=mask, src_pts=pts, src_in_trg_pts=moved, residual_raw=raw)
    return out


# ----------------------------------------------------------------------------
# depth render (core/depth_render.py:7-21 ; core/ops.py:59-96)
# ----------------------------------------------------------------------------
def splat_depth(points, K, hw):
    """Last-writer-wins z splat at truncated (v,u).  Order is undefined when two points share a
    pixel; callers of the oracle must use collision-free inputs or compare set-wise."""
    H, W = hw
    ok = points[..., 2] > 1e-6
    vu = pinhole(points, K).flip(-1).long()
    r, c = vu[..., 0], vu[..., 1]
    ok = ok & (r >= 0) & (r < H) & (c >= 0) & (c < W)
    img = torch.zeros(H * W, dtype=torch.float32)
    img.scatter_(0, (r[ok] * W + c[ok]), points[..., 2][ok])
    return img.reshape(H, W), ok


def render_depth(frame, kld, pose=None):
    with torch.no_grad():
        pts = unproject_keyframe(frame, kld)['src_pts']
        if pose is None:
            pose = torch.eye(4)
        img, _ = splat_depth(rigid(pts, pose), frame.K, frame.geo_hw())
    return img


# ----------------------------------------------------------------------------
# pyramids (image/gaussian_pyramid.py:42-118 ; image/keyframe.py:77-148)
# ----------------------------------------------------------------------------
def blur_decimate(img):
    """(1,C,H,W): reflect-pad 1, 3x3 binomial /16, keep even rows/cols."""
    p = F.pad(img, (1, 1, 1, 1), mode='reflect')
    w = (1.0, 2.0, 1.0)
    acc = torch.zeros_like(img)
    H, W = img.shape[-2:]
    for dy in range(3):
        for dx in range(3):
            acc = acc + (w[dy] * w[dx] / 16.0) * p[..., dy:dy + H, dx:dx + W]
    return acc[..., 0::2, 0::2]


def image_levels(img, start, end):
    """List coarse->fine of levels start..end-1 (end exclusive)."""
    levels, cur = [], img
    for i in range(end - 1):
        if i >= start:
            levels.insert(0, cur)
        cur = blur_decimate(cur)
    levels.insert(0, cur)
    return levels


def intrinsics_levels(K, start, end):
    out = []
    for i in range(start, end):
        s = 2.0 ** (-i)
        T = torch.tensor([[s, 0, s], [0, s, s], [0, 0, 1]], dtype=K.dtype)
        out.insert(0, T @ K)
    return out


def frame_pyramid(frame, start, end):
    """geo_down=False flavour (the only one any caller uses): geometry stays full-res."""
    imgs = image_levels(frame.image[:3][None], start, end)
    Ks = intrinsics_levels(frame.K, start, end)
    return [OracleFrame(im[0], frame.K.clone(), frame.logdepth_perseg, frame.keypoints, frame.keypoint_regions,
                        K_img=Kl, id=frame.id) for im, Kl in zip(imgs, Ks)]


# ----------------------------------------------------------------------------
# SE(3) helpers (lie/lie_algebra.py)
# ----------------------------------------------------------------------------
def quat_wxyz_to_R(q):
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    R = torch.stack((1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                     s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                     s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)), -1)
    return R.reshape(q.shape[:-1] + (3, 3))


def R_to_quat_wxyz(R):
    """Best-conditioned of the four candidate quaternions (lie_algebra.py:60-119)."""
    b = R.shape[:-2]
    m = R.reshape(b + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    qa = torch.stack((1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22), -1)
    qa = torch.sqrt(torch.clamp(qa, min=0))
    cand = torch.stack((
        torch.stack((qa[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), -1),
        torch.stack((m21 - m12, qa[..., 1] ** 2, m10 + m01, m02 + m20), -1),
        torch.stack((m02 - m20, m10 + m01, qa[..., 2] ** 2, m12 + m21), -1),
        torch.stack((m10 - m01, m20 + m02, m21 + m12, qa[..., 3] ** 2), -1)), -2)
    cand = cand / (2.0 * qa[..., None].clamp(min=0.1))
    pick = qa.argmax(-1)
    return torch.gather(cand, -2, pick[..., None, None].expand(b + (1, 4)))[..., 0, :]


def renormalise_se3(T):
    T[..., :3, :3] = quat_wxyz_to_R(R_to_quat_wxyz(T[..., :3, :3]))
    return T


def invert_se3(T):
    out = torch.empty_like(T)
    Rt = T[..., :3, :3].transpose(-1, -2)
    out[..., :3, :3] = Rt
    out[..., :3, 3:4] = -(Rt @ T[..., :3, 3:4])
    out[..., 3, :3] = 0
    out[..., 3, 3] = 1
    return out


def pose_to_tq(T):
    """(…,4,4) -> (…,7) [t, q_xyzw]."""
    q = R_to_quat_wxyz(T[..., :3, :3])
    return torch.cat((T[..., :3, 3], q[..., 1:], q[..., :1]), -1)


def so3_log(R, eps=1e-6):
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    d = tr - 3.0
    th = torch.acos(0.5 * (tr - 1))
    mag = torch.where(d < -eps, th / (2.0 * torch.sin(th)), 0.5 - d / 12.0 + d * d / 60.0)
    v = torch.stack((R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]), 1)
    return mag * v


def se3_log_reference_quirk(T, eps=1e-6):
    """lie_algebra.py:247-258 verbatim in meaning, including its elementwise ``(0.5*t)*(w x t)`` term."""
    w = so3_log(T[:, :3, :3])
    th = torch.clamp(torch.linalg.norm(w, dim=1), min=eps)
    wn = w / th
    t = T[:, :3, 3]
    c = torch.linalg.cross(wn, t)
    Vt = t - (0.5 * t) * c + (1.0 - th / (2.0 * torch.tan(0.5 * th))) * torch.linalg.cross(wn, c)
    return torch.cat((w, Vt), -1)


def se3_exp(xi):
    """Twist (…,6) ordered [tau, phi] -> (…,4,4).  Closed form; checked against scipy expm in tests.
    (lietorch itself is not under /root/reference: parity for this function is UNPINNED.)"""
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    small = th2 < 1e-8
    # double-where so autograd never differentiates sqrt at 0 (the Adam loops start from a zero tangent)
    ths = torch.sqrt(torch.where(small, torch.ones_like(th2), th2))
    A = torch.where(small, 1 - th2 / 6, torch.sin(ths) / ths)
    B = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(ths)) / (ths * ths))
    C = torch.where(small, 1.0 / 6 - th2 / 120, (ths - torch.sin(ths)) / (ths ** 3))
    z = torch.zeros_like(phi[..., 0])
    W = torch.stack((z, -phi[..., 2], phi[..., 1], phi[..., 2], z, -phi[..., 0], -phi[..., 1], phi[..., 0], z), -1)
    W = W.reshape(phi.shape[:-1] + (3, 3))
    W2 = W @ W
    I = torch.eye(3, dtype=xi.dtype).expand_as(W)
    R = I + A[..., None] * W + B[..., None] * W2
    V = I + B[..., None] * W + C[..., None] * W2
    T = torch.zeros(xi.shape[:-1] + (4, 4), dtype=xi.dtype)
    T[..., :3, :3] = R
    T[..., :3, 3] = (V @ tau[..., None])[..., 0]
    T[..., 3, 3] = 1
    return T


# ----------------------------------------------------------------------------
# segment statistics (odometery/depth_init.py:10-67 ; segment_based_completion.py:21-27)
# ----------------------------------------------------------------------------
def segment_depth_reinit(est_depth, frame, mode='mean'):
    """Per-segment mean / (lower) median of log(est) - L over valid pixels, plus L at the keypoint."""
    eps = 1e-6
    L, masks = frame.logdepth_perseg, frame.keypoint_regions
    N, H, W = L.shape
    rc = to_pixel_index(frame.keypoints, (H, W))
    est = est_depth.clone()
    good = ~(est < eps)
    est[~good] = eps
    shift = (torch.log(est)[None] - L)
    region = masks & good[None]
    count = region.sum((1, 2))
    seen = count > 0
    out = torch.zeros(N)
    at_kp = L[torch.arange(N), rc[:, 0], rc[:, 1]]
    for n in range(N):
        if not seen[n]:
            continue
        vals = shift[n][region[n]]
        out[n] = (vals.sum() / count[n] if mode == 'mean' else torch.median(vals)) + at_kp[n]
    if (~seen).any():
        out[~seen] = torch.median(out[seen])
    return out, seen


def average_depths(depths):
    """(N,H,W) with -1 outside masks -> per-pixel mean of entries > 1e-6, and the invalid map."""
    invalid = depths.max(dim=0)[0] < 1e-6
    d = depths.clone()
    d[d < 1e-6] = 0.0
    cnt = (d > 1e-6).sum(dim=0) + 1e-6
    return d.sum(dim=0) / cnt, invalid
