"""Seeded synthetic frame pairs for tests, golden vectors and bench.py.

No dataset ships with the build, so every input is rendered
analytically: a slanted textured plane seen from a source camera and from a
target camera displaced by a known SE(3) motion.  Because both images are ray
cast against the same plane, the ground-truth relative pose and the per-segment
keypoint log-depths are known exactly, which is what the convergence tests of
the Adam and Gauss-Newton drivers check against.

Everything is generated with ``numpy.random.default_rng(seed)`` (PCG64 is
stable across numpy versions and machines) in float64 and cast to float32 at
the end, so the same seed gives bit-identical inputs in this container and on
the GPU box.

Conventions follow the reference data model (``image/keyframe.py:20-75``):
``image`` (3,H,W) in [0,1]; ``K`` (3,3); ``logdepth_perseg`` (N,H,W), zero
outside each mask; ``keypoints`` (N,2) normalised (row, col) in [-1,1];
``keypoint_regions`` (N,H,W) bool.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


# ----------------------------------------------------------------------------
# tiny float64 SE(3) helpers (host side only; tangent order [tau, phi] like the
# pose parameter the reference optimises)
# ----------------------------------------------------------------------------
def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp_np(xi):
    """Exp of a twist ``xi = [tau(3), phi(3)]`` -> 4x4 float64."""
    xi = np.asarray(xi, dtype=np.float64)
    tau, phi = xi[:3], xi[3:]
    th = float(np.linalg.norm(phi))
    W = hat(phi)
    if th < 1e-8:
        A, B, C = 1.0 - th * th / 6.0, 0.5 - th * th / 24.0, 1.0 / 6.0 - th * th / 120.0
    else:
        A = math.sin(th) / th
        B = (1.0 - math.cos(th)) / (th * th)
        C = (th - math.sin(th)) / (th ** 3)
    R = np.eye(3) + A * W + B * (W @ W)
    V = np.eye(3) + B * W + C * (W @ W)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ tau
    return T


# ----------------------------------------------------------------------------
@dataclass
class SynthPair:
    """One synthetic source keyframe + target (supporting) frame."""
    H: int
    W: int
    N: int
    K: np.ndarray                 # (3,3) f32
    src_image: np.ndarray         # (3,H,W) f32
    trg_image: np.ndarray         # (3,H,W) f32
    depth: np.ndarray             # (H,W) f32 source depth (ground truth)
    logdepth_perseg: np.ndarray   # (N,H,W) f32
    keypoints: np.ndarray         # (N,2) f32 normalised (row, col)
    keypoint_regions: np.ndarray  # (N,H,W) bool
    kld_gt: np.ndarray            # (N,) f32 log-depth at the keypoints
    kld_init: np.ndarray          # (N,) f32  log(2 + 2*rand)
    pose_gt: np.ndarray           # (4,4) f32  target <- source
    pose_init: np.ndarray         # (4,4) f32  perturbed start
    meta: dict = field(default_factory=dict)


def _texture(X, rng_tex, base_omega):
    """Band-limited RGB texture evaluated at 3-D points X (...,3) -> (3,...)."""
    out = []
    for ch in range(3):
        acc = np.full(X.shape[:-1], 0.5)
        for k in range(rng_tex.shape[1]):
            direction = rng_tex[ch, k, :3]
            omega = base_omega * rng_tex[ch, k, 3]
            phase = rng_tex[ch, k, 4]
            amp = rng_tex[ch, k, 5]
            acc = acc + amp * np.sin(omega * (X @ direction) + phase)
        out.append(acc)
    return np.clip(np.stack(out, 0), 0.0, 1.0)


def _octave_texture_table(rng, n_octaves, per_octave=4, slope=1.0, total_amp=0.45):
    """Component table (3, n_octaves * per_octave, 6) of a multi-octave (~1/f) texture for ``_texture``: octave o has
    ``per_octave`` randomly oriented sinusoids of period ``2^o x (1 ... 1.6) x`` the shortest one and amplitude proportional
    to ``period^slope`` (slope 1 = a 1/f amplitude spectrum, what natural images roughly have), scaled so that the amplitudes of
    a channel sum to ``total_amp``.  Coarse-to-fine alignment has structure to hold on to at every pyramid level -- unlike the
    single-octave band texture, whose basin of attraction is half its shortest period at every level."""
    n = n_octaves * per_octave
    tex = np.empty((3, n, 6))
    d = rng.standard_normal((3, n, 3))
    tex[:, :, :3] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    period = (2.0 ** np.repeat(np.arange(n_octaves), per_octave))[None, :] * rng.uniform(1.0, 1.6, (3, n))
    tex[:, :, 3] = 1.0 / period                    # omega multiplier relative to the shortest period
    tex[:, :, 4] = rng.uniform(0, 2 * math.pi, (3, n))
    amp = period ** slope * rng.uniform(0.7, 1.3, (3, n))
    tex[:, :, 5] = total_amp * amp / amp.sum(axis=1, keepdims=True)
    return tex


def _grid_shape(N):
    gh = int(math.floor(math.sqrt(N)))
    while N % gh:
        gh -= 1
    return gh, N // gh


def make_pair(H=60, W=80, N=6, seed=0, *, overlap=0, shape="grid",
              motion_scale=1.0, init_sigma=0.02, texture_period_px=None,
              drop_border=0, texture="band", init_mode="left", octave_slope=1.0, blob_coverage=None):
    """Render one seeded source/target pair.

    ``shape``: 'grid' = gh x gw rectangular tiling; 'blobs' =
    random overlapping ellipses (emulates SAM masks, rho > 1, ragged sizes).
    ``overlap``: grow each grid tile by this many pixels on every side.
    ``drop_border``: leave an unsegmented band of this many pixels.
    ``texture_period_px``: shortest period of the band-limited texture (the six components per channel span 1x ... 2.9x
    of it).  Default: 14 px up to 320 columns, scaled with the width above (28 px at 640x480) -- the random depth seeds
    ``log(2 + 2 rand)`` (two_frame_sfm.py:103-105) put a segment up to a factor 2 off in depth, i.e. ~8 px of disparity
    at 640x480; with a texture period that does not grow with the resolution such a segment starts outside the basin of
    attraction even at the coarsest of 3 pyramid levels, for the reference's Adam as for Gauss-Newton.
    ``texture``: 'band' = that single-octave texture (six components per channel); 'octaves' = a multi-octave ~1/f texture
    (``_octave_texture_table``; shortest period ``texture_period_px``, default 8 px per 320 columns, octaves up to about the
    image width, amplitude ~ period^``octave_slope``) -- what the reference's own starting distribution needs.
    ``blob_coverage`` (shape='blobs'): scale the ellipses so that their areas sum to about this many image areas (rho; SAM-like
    masks cover the image ~1.2 times) -- the default (None) keeps the unscaled ellipses, whose coverage grows with N (rho ~ 0.11 N).
    ``init_mode``: 'left' = ``pose_init = Exp(sigma xi) T_gt`` (the default of every earlier golden); 'reference' =
    ``T_gt Exp(sigma xi)``, the reference's ``current_T.mul(SE3.Random(sigma=0.05))`` (odometery/two_frame_sfm.py:77-81;
    lietorch's Random is exp(sigma * randn(6)) on the tangent [tau, phi]).
    """
    if texture_period_px is None:
        texture_period_px = (8.0 if texture == "octaves" else 14.0) * max(1.0, W / 320.0)
    rng = np.random.default_rng(seed)
    fx = fy = 0.8 * W
    cx, cy = W / 2.0, H / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    # plane n.X = h in the source camera frame
    n = np.array([0.22, -0.12, 1.0]) + 0.05 * rng.standard_normal(3)
    n /= np.linalg.norm(n)
    h = 3.0

    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rays = np.stack([(cols - cx) / fx, (rows - cy) / fy, np.ones_like(cols)], -1)

    depth = h / (rays @ n)
    Xs = rays * depth[..., None]

    base_omega = 2.0 * math.pi / (texture_period_px * h / fx)
    if texture == "octaves":
        n_oct = max(1, int(math.floor(math.log2(max(W, H) / texture_period_px))) + 1)
        tex = _octave_texture_table(rng, n_oct, slope=octave_slope)
    elif texture == "band":
        tex = np.empty((3, 6, 6))
        d = rng.standard_normal((3, 6, 3))
        tex[:, :, :3] = d / np.linalg.norm(d, axis=-1, keepdims=True)
        tex[:, :, 3] = rng.uniform(0.35, 1.0, (3, 6))
        tex[:, :, 4] = rng.uniform(0, 2 * math.pi, (3, 6))
        tex[:, :, 5] = rng.uniform(0.04, 0.11, (3, 6))
    else:
        raise ValueError(texture)
    src_image = _texture(Xs, tex, base_omega)

    # ground-truth motion target <- source
    xi_gt = motion_scale * np.array([0.06, -0.035, 0.025, 0.012, -0.02, 0.015])
    xi_gt = xi_gt * (1.0 + 0.2 * rng.standard_normal(6))
    T_gt = se3_exp_np(xi_gt)
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    # target rays -> plane (expressed in the source frame)
    num = h + n @ (R.T @ t)
    den = (rays @ R) @ n          # n . R^T ray_t
    d_t = num / den
    Xt = rays * d_t[..., None]
    Xs_from_t = (Xt - t) @ R      # R^T (X_t - t)
    trg_image = _texture(Xs_from_t, tex, base_omega)

    # segments
    masks = np.zeros((N, H, W), dtype=bool)
    kp_rc = np.zeros((N, 2), dtype=np.int64)
    if shape == "grid":
        gh, gw = _grid_shape(N)
        b = drop_border
        r_edges = np.linspace(b, H - b, gh + 1).round().astype(int)
        c_edges = np.linspace(b, W - b, gw + 1).round().astype(int)
        k = 0
        for i in range(gh):
            for j in range(gw):
                r0, r1 = max(r_edges[i] - overlap, 0), min(r_edges[i + 1] + overlap, H)
                c0, c1 = max(c_edges[j] - overlap, 0), min(c_edges[j + 1] + overlap, W)
                masks[k, r0:r1, c0:c1] = True
                kp_rc[k] = ((r_edges[i] + r_edges[i + 1]) // 2, (c_edges[j] + c_edges[j + 1]) // 2)
                k += 1
    elif shape == "blobs":
        bs = 1.0 if blob_coverage is None else math.sqrt(blob_coverage / (N * math.pi * 0.19 * 0.19))
        for k in range(N):
            cr, cc = rng.uniform(0.1 * H, 0.9 * H), rng.uniform(0.1 * W, 0.9 * W)
            ar, ac = bs * rng.uniform(0.08 * H, 0.3 * H), bs * rng.uniform(0.08 * W, 0.3 * W)
            ang = rng.uniform(0, math.pi)
            dr, dc = rows - cr, cols - cc
            a = (dc * math.cos(ang) + dr * math.sin(ang)) / ac
            bb = (-dc * math.sin(ang) + dr * math.cos(ang)) / ar
            masks[k] = (a * a + bb * bb) <= 1.0
            kp_rc[k] = (int(round(cr)), int(round(cc)))
            kp_rc[k, 0] = min(max(kp_rc[k, 0], 0), H - 1)
            kp_rc[k, 1] = min(max(kp_rc[k, 1], 0), W - 1)
            masks[k, kp_rc[k, 0], kp_rc[k, 1]] = True
    elif shape == "sam":
        # SAM-REALISTIC SEGMENT SETS (round 6; what the reference's frontend emits: frontend/segment/mask_generation.py:143-312 -> masks of
        # any size from tens of pixels to a third of the frame, nested, with holes; post_processer.py:160-181 splits what is not connected,
        # so N differs from keyframe to keyframe): areas log-uniform between 30 px and 0.3 HW (a power law: many small masks, a few large
        # ones) scaled towards a total coverage of ``blob_coverage`` image areas; a fifth of the masks have a HOLE, a fifth sit NESTED
        # inside a larger one, a tenth consist of two separate lobes and are SPLIT into their connected components like the reference's
        # post-processing does.  N is what comes out (the nominal N is the number of masks drawn).
        from scipy import ndimage
        rho = 1.2 if blob_coverage is None else float(blob_coverage)
        a_min, a_max = 30.0, 0.3 * H * W
        areas = np.exp(rng.uniform(math.log(a_min), math.log(a_max), N))
        for _ in range(8):
            areas = np.clip(areas * (rho * H * W / areas.sum()), a_min, a_max)

        def ellipse(cr, cc, a, b, ang):
            dr, dc = rows - cr, cols - cc
            u = (dc * math.cos(ang) + dr * math.sin(ang)) / max(a, 0.6)
            v = (-dc * math.sin(ang) + dr * math.cos(ang)) / max(b, 0.6)
            return (u * u + v * v) <= 1.0

        parts = []
        for k in np.argsort(-areas):                        # large first: a nested mask picks its parent among those drawn before
            A, kind = float(areas[k]), rng.uniform()
            aspect, ang = rng.uniform(1.0, 3.0), rng.uniform(0, math.pi)
            b_ax = math.sqrt(A / (math.pi * aspect))
            a_ax = aspect * b_ax
            if kind < 0.2 and parts:
                pr, pc = np.nonzero(parts[int(rng.integers(len(parts)))])
                j = int(rng.integers(len(pr)))
                cr, cc = float(pr[j]), float(pc[j])
            else:
                cr, cc = rng.uniform(0.05 * H, 0.95 * H), rng.uniform(0.05 * W, 0.95 * W)
            if 0.4 <= kind < 0.5 and A > 200:               # two lobes
                d = 1.6 * a_ax / math.sqrt(2.0)
                off_r, off_c = d * math.sin(ang), d * math.cos(ang)
                m = ellipse(cr - off_r, cc - off_c, a_ax / math.sqrt(2.0), b_ax / math.sqrt(2.0), ang) | \
                    ellipse(cr + off_r, cc + off_c, a_ax / math.sqrt(2.0), b_ax / math.sqrt(2.0), ang)
            else:
                m = ellipse(cr, cc, a_ax, b_ax, ang)
                if 0.2 <= kind < 0.4 and A > 400:           # a hole
                    f = rng.uniform(0.3, 0.6)
                    m &= ~ellipse(cr + 0.15 * b_ax * rng.standard_normal(), cc + 0.15 * b_ax * rng.standard_normal(), f * a_ax, f * b_ax, ang)
            lab, n_lab = ndimage.label(m)
            for c in range(1, n_lab + 1):
                part = lab == c
                if part.sum() >= 16:
                    parts.append(part)
        if not parts:
            parts = [ellipse(0.5 * H, 0.5 * W, 0.2 * W, 0.2 * H, 0.0)]
        N = len(parts)
        masks = np.stack(parts)
        kp_rc = np.zeros((N, 2), dtype=np.int64)
        for k in range(N):                                  # keypoint: the mask pixel nearest the centroid (a ring's centroid is in its hole)
            pr, pc = np.nonzero(masks[k])
            j = int(np.argmin((pr - pr.mean()) ** 2 + (pc - pc.mean()) ** 2))
            kp_rc[k] = (pr[j], pc[j])
    else:
        raise ValueError(shape)

    logD = np.log(depth)
    offs = rng.uniform(-0.7, 0.7, N)     # per-segment unknown scale of the integrated depth
    L = (logD[None] - offs[:, None, None]) * masks
    keypoints = np.stack([2.0 * kp_rc[:, 0] / (H - 1) - 1.0, 2.0 * kp_rc[:, 1] / (W - 1) - 1.0], 1)
    kld_gt = logD[kp_rc[:, 0], kp_rc[:, 1]]
    kld_init = np.log(2.0 + 2.0 * rng.uniform(size=N))      # odometery/two_frame_sfm.py:103-105
    noise = se3_exp_np(init_sigma * rng.standard_normal(6))
    if init_mode == "left":
        T_init = noise @ T_gt
    elif init_mode == "reference":
        T_init = T_gt @ noise
    else:
        raise ValueError(init_mode)

    f32 = np.float32
    return SynthPair(
        H=H, W=W, N=N, K=K.astype(f32),
        src_image=src_image.astype(f32), trg_image=trg_image.astype(f32),
        depth=depth.astype(f32), logdepth_perseg=L.astype(f32),
        keypoints=keypoints.astype(f32), keypoint_regions=masks,
        kld_gt=kld_gt.astype(f32), kld_init=kld_init.astype(f32),
        pose_gt=T_gt.astype(f32), pose_init=T_init.astype(f32),
        meta=dict(seed=seed, shape=shape, overlap=overlap, xi_gt=xi_gt, kp_rc=kp_rc, texture=texture,
                  texture_period_px=float(texture_period_px), init_sigma=float(init_sigma), init_mode=init_mode),
    )


def observable_segments(p, min_points=16):
    """(N,) bool: segments with at least ``min_points`` pixels whose ground-truth re-projection lands inside the target's validity band (0.99 of
    the normalised frame, core/dense_optim.py:128-162).  A segment that the target frame does not see has NO depth to converge to -- the
    reference's Adam leaves it at its seed (zero gradient), Gauss-Newton likewise -- so it must not count as a depth error (SAM-realistic
    scenes have 30-pixel masks at the image border)."""
    H, W = p.depth.shape
    fx, fy, cx, cy = float(p.K[0, 0]), float(p.K[1, 1]), float(p.K[0, 2]), float(p.K[1, 2])
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = p.depth.astype(np.float64)
    X = np.stack(((cols - cx) / fx * d, (rows - cy) / fy * d, d), -1) @ p.pose_gt[:3, :3].astype(np.float64).T + p.pose_gt[:3, 3].astype(np.float64)
    u, v = fx * X[..., 0] / X[..., 2] + cx, fy * X[..., 1] / X[..., 2] + cy
    ok = (np.abs(2 * u / (W - 1) - 1) <= 0.99) & (np.abs(2 * v / (H - 1) - 1) <= 0.99) & (X[..., 2] > 1e-6)
    return (p.keypoint_regions & ok[None]).reshape(p.N, -1).sum(1) >= min_points


def stepped_logdepth(pair, seed=0, n_boxes=3, factor=(0.55, 0.75)):
    """Per-segment log-depths of ``pair`` after pulling a few axis-aligned boxes towards the camera: depth steps
    inside segments, the situation ``frontend/segment/post_processer.py`` exists for.  Returns (N,H,W) f32."""
    rng = np.random.default_rng(seed)
    d = pair.depth.astype(np.float64).copy()
    H, W = d.shape
    for _ in range(n_boxes):
        r0, c0 = int(rng.uniform(0.05, 0.6) * H), int(rng.uniform(0.05, 0.6) * W)
        r1, c1 = r0 + int(rng.uniform(0.15, 0.35) * H), c0 + int(rng.uniform(0.15, 0.35) * W)
        d[r0:r1, c0:c1] *= rng.uniform(*factor)
    return (np.log(d)[None] * pair.keypoint_regions).astype(np.float32)


# ----------------------------------------------------------------------------
# several frames of one scene (windowed mapping / MonoVO-shaped inputs)
# ----------------------------------------------------------------------------
@dataclass
class SynthFrame:
    """One frame of a synthetic sequence.  ``T_wc`` is camera-to-world (the reference's ``kf_poses`` /
    supporting-frame poses, ``odometery/odometery.py:793``: relative pose = inv(T_trg) @ T_src)."""
    image: np.ndarray             # (3,H,W) f32
    K: np.ndarray                 # (3,3) f32
    T_wc: np.ndarray              # (4,4) f32 ground truth
    depth: np.ndarray             # (H,W) f32
    logdepth_perseg: np.ndarray | None = None   # keyframes only
    keypoints: np.ndarray | None = None
    keypoint_regions: np.ndarray | None = None
    kld_gt: np.ndarray | None = None


def make_sequence(H, W, N, twists, keyframe_ids, seed=0, overlap=0, texture_period_px=None):
    """Frames of ONE textured plane seen from cameras ``T_wc[k] = Exp(twists[k])`` (world = camera of the zero twist).
    Frames listed in ``keyframe_ids`` also get an N-segment grid tiling with per-segment log-depths (the same
    construction as ``make_pair``); the others are supporting frames (image + K only).  Returns [SynthFrame]."""
    if texture_period_px is None:
        texture_period_px = 14.0 * max(1.0, W / 320.0)
    rng = np.random.default_rng(seed)
    fx = fy = 0.8 * W
    cx, cy = W / 2.0, H / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    n = np.array([0.22, -0.12, 1.0]) + 0.05 * rng.standard_normal(3)
    n /= np.linalg.norm(n)
    h = 3.0
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rays = np.stack([(cols - cx) / fx, (rows - cy) / fy, np.ones_like(cols)], -1)
    base_omega = 2.0 * math.pi / (texture_period_px * h / fx)
    tex = np.empty((3, 6, 6))
    d = rng.standard_normal((3, 6, 3))
    tex[:, :, :3] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    tex[:, :, 3] = rng.uniform(0.35, 1.0, (3, 6))
    tex[:, :, 4] = rng.uniform(0, 2 * math.pi, (3, 6))
    tex[:, :, 5] = rng.uniform(0.04, 0.11, (3, 6))
    gh, gw = _grid_shape(N)
    r_edges = np.linspace(0, H, gh + 1).round().astype(int)
    c_edges = np.linspace(0, W, gw + 1).round().astype(int)
    f32 = np.float32
    frames = []
    for k, xi in enumerate(twists):
        T = se3_exp_np(np.asarray(xi, dtype=np.float64))
        R, t = T[:3, :3], T[:3, 3]
        depth = (h - n @ t) / ((rays @ R.T) @ n)           # world point = R ray d + t on the plane n.X = h
        Xw = (rays * depth[..., None]) @ R.T + t
        fr = SynthFrame(image=_texture(Xw, tex, base_omega).astype(f32), K=K.astype(f32), T_wc=T.astype(f32),
                        depth=depth.astype(f32))
        if k in keyframe_ids:
            masks = np.zeros((N, H, W), dtype=bool)
            kp_rc = np.zeros((N, 2), dtype=np.int64)
            q = 0
            for i in range(gh):
                for j in range(gw):
                    r0, r1 = max(r_edges[i] - overlap, 0), min(r_edges[i + 1] + overlap, H)
                    c0, c1 = max(c_edges[j] - overlap, 0), min(c_edges[j + 1] + overlap, W)
                    masks[q, r0:r1, c0:c1] = True
                    kp_rc[q] = ((r_edges[i] + r_edges[i + 1]) // 2, (c_edges[j] + c_edges[j + 1]) // 2)
                    q += 1
            logD = np.log(depth)
            offs = rng.uniform(-0.7, 0.7, N)
            fr.logdepth_perseg = ((logD[None] - offs[:, None, None]) * masks).astype(f32)
            fr.keypoints = np.stack([2.0 * kp_rc[:, 0] / (H - 1) - 1.0, 2.0 * kp_rc[:, 1] / (W - 1) - 1.0], 1).astype(f32)
            fr.keypoint_regions = masks
            fr.kld_gt = logD[kp_rc[:, 0], kp_rc[:, 1]].astype(f32)
        frames.append(fr)
    return frames


def window_inputs(seed, n_kf, H=48, W=64, N=6):
    """A synthetic MonoVO window: keyframes at even frames, one supporting frame after each.  Returns the frames and
    perturbed initial estimates (poses, keypoint log-depths, affines)."""
    rng = np.random.default_rng(seed)
    base = np.array([0.05, -0.02, 0.015, 0.01, -0.015, 0.008])
    twists = [k * base * (1.0 + 0.15 * rng.standard_normal(6)) for k in range(2 * n_kf)]
    frames = make_sequence(H, W, N, twists, keyframe_ids=list(range(0, 2 * n_kf, 2)), seed=seed, overlap=1)
    est = []
    for k, f in enumerate(frames):
        T0 = f.T_wc.astype(np.float64) if k == 0 else f.T_wc.astype(np.float64) @ se3_exp_np(0.004 * rng.standard_normal(6))
        est.append(T0.astype(np.float32))
    klds = [(frames[k].kld_gt + 0.03 * rng.standard_normal(N)).astype(np.float32) for k in range(0, 2 * n_kf, 2)]
    affs = (0.01 * rng.standard_normal((2 * n_kf, 2))).astype(np.float32)
    affs[0] = 0
    return frames, est, klds, affs


def reference_window_inputs(seed, n_kf=5, n_supp=2, n_running=2, H=224, W=288, N=40, pose_sigma=0.004, kld_sigma=0.03, step_scale=0.6):
    """A mapping window at the reference's extent (config/tum/odom_desk.yaml: ``window_size: 5``, ``supp_every_n: 3`` -> two
    supporting frames per keyframe, plus the two running ones of the latest keyframe, odometery.py:1327-1360): keyframe k is
    followed by its ``n_supp`` supporting frames (the last keyframe by ``n_running``), all on one smooth trajectory.
    Returns (frames, kf_index [n_kf], supp_index [n_kf][...], est poses per frame, klds per keyframe, affs per frame)."""
    rng = np.random.default_rng(seed)
    base = step_scale * np.array([0.05, -0.02, 0.015, 0.01, -0.015, 0.008])
    kf_index, supp_index, n = [], [], 0
    for k in range(n_kf):
        kf_index.append(n)
        m = n_running if k == n_kf - 1 else n_supp
        supp_index.append(list(range(n + 1, n + 1 + m)))
        n += 1 + m
    twists = [k * base * (1.0 + 0.05 * rng.standard_normal(6)) for k in range(n)]
    frames = make_sequence(H, W, N, twists, keyframe_ids=kf_index, seed=seed, overlap=1)
    est = []
    for k, f in enumerate(frames):
        T0 = f.T_wc.astype(np.float64) if k == 0 else f.T_wc.astype(np.float64) @ se3_exp_np(pose_sigma * rng.standard_normal(6))
        est.append(T0.astype(np.float32))
    klds = [(frames[k].kld_gt + kld_sigma * rng.standard_normal(N)).astype(np.float32) for k in kf_index]
    affs = (0.01 * rng.standard_normal((n, 2))).astype(np.float32)
    affs[0] = 0
    return frames, kf_index, supp_index, est, klds, affs
