"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Gauss-Newton normal equations of the photometric cost, built the slow, independent way: the oracle's own
per-point residual vector (float64, the reference's formulation via ``photometric_oracle``) differentiated by
float64 central differences (h = 1e-7; the residual is piecewise bilinear in the sampling position, so the
truncation error is ~1e-9 relative) with respect to a left SE(3) perturbation xi = [tau, phi] of the pose and
the per-segment keypoint log-depths, then  H = J^T W J,  b = J^T W r  with the IRLS weights of the L1 cost
W = diag(1 / max(|r|, eps)).

The reference itself contains no Gauss-Newton solver: this file pins the *derivative* the HIP GN
kernel accumulates to the reference's own residual definition (core/dense_optim.py:228-261,265-363); the solver
on top (Schur complement + LM) is checked through cost decrease and convergence to the synthetic ground truth.
"""
from __future__ import annotations

import torch

from . import photometric_oracle as orc


def residual_vector(src, trg, kld, pose, affine=None):
    """(3P,) = ((I_src - I_trg') * mask) flattened channel-major, plus the mask (P,)."""
    out = orc.photometric_cost(src, trg, kld, pose, collect_stats=0, affine=affine)
    # rebuild the raw residual with autograd attached (photometric_cost detaches its stats copy)
    geo = src.geo_hw()
    depth = torch.exp(orc.seed_logdepths(kld, src))
    pts, seg, _ = orc.gather_segments(depth, src.keypoint_regions, src.K)
    moved = orc.rigid(pts, pose)
    src_vals, src_ok = orc.sample_single(src.image, pts, src.K, geo)
    trg_vals, trg_ok = orc.sample_single(trg.image, moved, trg.K, geo)
    mask = (trg_ok & src_ok)[:, None].to(src_vals.dtype)
    if affine is not None:
        trg_vals = orc.brightness(trg_vals, affine[0], affine[1])
    raw = (src_vals[:, :3].detach() - trg_vals[:, :3]) * mask
    return raw.reshape(-1), mask.reshape(-1), seg, out["residual"]


def normal_equations(src, trg, kld, pose, eps=1e-3, affine=None, columns=None, with_affine=False):
    """Returns dict(H (6+N,6+N), b (6+N), cost (mean |r| like the reference), n_valid) in float64.  ``columns``: only
    these unknowns (indices into [xi(6), kld(N)]) are differentiated -- H and b are then the corresponding sub-block
    (full-size keyframes: 6 pose columns + a few segments instead of 2 (6+N) dense evaluations).
    ``with_affine``: the TARGET frame's brightness pair (a_t, b_t) of ``affine = ((a_s, b_s), (a_t, b_t))`` joins the unknowns as
    the last two columns: x = [xi(6), kld(N), a_t, b_t] (the window optimiser's Gauss-Newton flavour, sp_pairs_cost mode 2)."""
    src64 = orc.OracleFrame(src.image.double(), src.K.double(), src.logdepth_perseg.double(), src.keypoints.double(),
                            src.keypoint_regions)
    trg64 = orc.OracleFrame(trg.image.double(), trg.K.double())
    kld0, pose0 = kld.double(), pose.double()
    N = kld0.numel()
    assert not with_affine or affine is not None

    aff64 = None if affine is None else (affine[0].double(), affine[1].double())

    def f(x):
        T = orc.se3_exp(x[:6][None])[0] @ pose0
        aff = aff64 if not with_affine else (aff64[0], x[6 + N: 8 + N])
        r, _, _, _ = residual_vector(src64, trg64, x[6: 6 + N], T, aff)
        return r

    x0 = torch.cat((torch.zeros(6, dtype=torch.float64), kld0) + ((aff64[1],) if with_affine else ()))
    h = 1e-7
    cols = []
    with torch.no_grad():
        for i in (range(x0.numel()) if columns is None else columns):
            e = torch.zeros_like(x0)
            e[i] = h
            cols.append((f(x0 + e) - f(x0 - e)) / (2 * h))
    J = torch.stack(cols, dim=1)
    r, mask, seg, res = residual_vector(src64, trg64, kld0, pose0, aff64)
    m3 = mask.repeat(3)
    w = m3 / torch.clamp(r.abs(), min=eps)
    H = J.T @ (w[:, None] * J)
    b = J.T @ (w * r)
    return dict(H=H, b=b, cost=float(res), n_valid=float(mask.sum()), N=N)
