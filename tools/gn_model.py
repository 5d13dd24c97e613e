#!/usr/bin/env python
"""Developer tool (CPU, numpy fp64): a line-by-line MODEL of the many-pairs Gauss-Newton / LM path -- the sums k_cost_pairs<1>
accumulates (sp_cost.hip fold_gn2 / finish_gn2), the per-pair solver (sp_solve_device.h solve_gn: Schur complement, LM accept /
undo, per-pair phase advance) and the coarse-to-fine schedule of PairBatch.run_scheduled -- for ONE frame pair, so that schedule
and damping policies can be explored on the CPU build container before they are put on the device.  Not imported by the product,
the tests or bench.py; results quoted in DESIGN.md come from the GPU (tools/sigma05_sweep.py).

    python tools/gn_model.py --seed 501 [--size 240x320x8] [--policy ...]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth  # noqa: E402
from parity_util import pose_depth_errors  # noqa: E402


def blur_decimate(img):
    """image/gaussian_pyramid.py:53-85: reflect pad 1, 3x3 binomial / 16, keep [::2, ::2].  img (3,H,W)."""
    p = np.pad(img, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    k = np.array([1.0, 2.0, 1.0])
    out = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            out += k[dy] * k[dx] / 16.0 * p[:, dy:dy + img.shape[1], dx:dx + img.shape[2]]
    return out[:, ::2, ::2]


def bilinear(img, ix, iy):
    """value and both slopes of a (3,H,W) image at float positions (taps assumed inside)."""
    H, W = img.shape[1:]
    x0 = np.clip(np.floor(ix).astype(int), 0, W - 2)
    y0 = np.clip(np.floor(iy).astype(int), 0, H - 2)
    wx, wy = ix - x0, iy - y0
    a, b, c, d = img[:, y0, x0], img[:, y0, x0 + 1], img[:, y0 + 1, x0], img[:, y0 + 1, x0 + 1]
    e1, e2, e3 = b - a, c - a, (d - c) - (b - a)
    Iy = e2 + wx * e3
    Ix = e1 + wy * e3
    val = a + wx * e1 + wy * Iy
    return val, Ix, Iy


class PairModel:
    def __init__(self, pair, n_levels=3, strides=None):
        self.pair = pair
        H, W = pair.H, pair.W
        self.H, self.W = H, W
        K = pair.K.astype(np.float64)
        self.fx, self.fy, self.cx, self.cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        src, trg = [pair.src_image.astype(np.float64)], [pair.trg_image.astype(np.float64)]
        for _ in range(n_levels - 1):
            src.append(blur_decimate(src[-1]))
            trg.append(blur_decimate(trg[-1]))
        self.src, self.trg = src, trg
        seg, row, col = np.nonzero(pair.keypoint_regions)
        self.seg, self.row, self.col = seg, row.astype(np.float64), col.astype(np.float64)
        L = pair.logdepth_perseg.astype(np.float64)
        self.baseL = L[seg, row, col]
        kp = pair.keypoints.astype(np.float64)
        kr = np.rint(0.5 * (H - 1) * (kp[:, 0] + 1)).astype(int)
        kc = np.rint(0.5 * (W - 1) * (kp[:, 1] + 1)).astype(int)
        self.kp_L = L[np.arange(pair.N), kr, kc]
        self.N = pair.N
        self.P = len(seg)
        self.strides = strides or [1] * n_levels
        self.sel = {s: np.nonzero((row % s == 0) & (col % s == 0))[0] for s in set(self.strides)}
        # source colours per level at the point's own pixel (align-corners mapping of the geometry grid onto the level)
        self.src_rgb = {}
        for l in range(n_levels):
            Hl, Wl = src[l].shape[1:]
            self.src_rgb[l] = bilinear(src[l], self.col * (Wl - 1) / (W - 1), self.row * (Hl - 1) / (H - 1))[0]

    def system(self, level, stride, pose, kld, eps):
        """Normal equations over x = [tau, phi, kld_0..N-1] (left perturbation) on the stride lattice of `level`.  Returns cost (the
        reference's mean over 3 P_lattice values), Hpp (6,6), bp (6), hpd (N,6), D (N), bd (N), n_valid."""
        idx = self.sel[stride]
        seg = self.seg[idx]
        d = np.exp(self.baseL[idx] + (kld - self.kp_L)[seg])
        x = (self.col[idx] - self.cx) * d / self.fx
        y = (self.row[idx] - self.cy) * d / self.fy
        p = np.stack((x, y, d), 0)
        R, t = pose[:3, :3], pose[:3, 3]
        q = R @ p + t[:, None]
        zg = np.abs(q[2]) > 1e-6
        zinv = np.where(zg, 1.0 / np.where(zg, q[2], 1.0), 1e-6)
        u = q[0] * self.fx * zinv + self.cx
        v = q[1] * self.fy * zinv + self.cy
        xn = 2 * u / (self.W - 1) - 1
        yn = 2 * v / (self.H - 1) - 1
        ok = (np.abs(xn) <= 0.99) & (np.abs(yn) <= 0.99) & (q[2] > 1e-7) & (d > 1e-7)
        img = self.trg[level]
        Hl, Wl = img.shape[1:]
        ax, ay = (Wl - 1) / (self.W - 1), (Hl - 1) / (self.H - 1)
        ix = np.where(ok, (xn + 1) * 0.5 * (Wl - 1), 0.0)
        iy = np.where(ok, (yn + 1) * 0.5 * (Hl - 1), 0.0)
        val, Ix, Iy = bilinear(img, ix, iy)
        r = (self.src_rgb[level][:, idx] - val) * ok
        cost = np.abs(r).sum() / (3.0 * max(len(idx), 1))
        w = ok / np.maximum(np.abs(r), eps)                     # (3,n)
        # A (2x7): d(ix,iy)/d[tau, phi, kld_seg]
        ga, gb = ax * self.fx * zinv * ok, ay * self.fy * zinv * ok
        zi = np.where(zg, zinv, 0.0) * ok
        ux, vy = q[0] * zi, q[1] * zi
        e = q - t[:, None]
        A0 = np.stack((np.ones_like(ux), np.zeros_like(ux), -ux, -ux * q[1], q[2] + ux * q[0], -q[1], e[0] - ux * e[2]), 0) * ga
        A1 = np.stack((np.zeros_like(ux), np.ones_like(ux), -vy, -(q[2] + vy * q[1]), vy * q[0], q[0], e[1] - vy * e[2]), 0) * gb
        # J_ch = -(Ix_ch A0 + Iy_ch A1)
        W00 = (w * Ix * Ix).sum(0); W01 = (w * Ix * Iy).sum(0); W11 = (w * Iy * Iy).sum(0)
        V0 = -(w * r * Ix).sum(0); V1 = -(w * r * Iy).sum(0)
        B0 = W00 * A0 + W01 * A1
        B1 = W01 * A0 + W11 * A1
        Hfull = A0 @ B0.T + A1 @ B1.T                              # 7x7 with the depth column shared by all segments
        b = A0 @ V0 + A1 @ V1
        Hpp, bp = Hfull[:6, :6], b[:6]
        N = self.N
        hpd = np.zeros((N, 6)); D = np.zeros(N); bd = np.zeros(N)
        c6 = A0[:6] * B0[6] + A1[:6] * B1[6]
        dd = A0[6] * B0[6] + A1[6] * B1[6]
        bb = A0[6] * V0 + A1[6] * V1
        for k in range(6):
            hpd[:, k] = np.bincount(seg, weights=c6[k], minlength=N)
        D = np.bincount(seg, weights=dd, minlength=N)
        bd = np.bincount(seg, weights=bb, minlength=N)
        nv = np.bincount(seg, weights=ok.astype(np.float64), minlength=N)
        return dict(cost=cost, Hpp=Hpp, bp=bp, hpd=hpd, D=D, bd=bd, n_valid=float(ok.sum()), seg_valid=nv, n=len(idx))


def se3_exp(xi):
    return synth.se3_exp_np(xi)


def lm_step(sys_, lam, depth_clamp=0.5, depth_prior=0.0, pose_only=False):
    """solve_gn: Schur complement onto the pose block with D(1+lam) [+ depth_prior * n_valid_seg], clamp on the depth step."""
    Hpp, bp, hpd, D, bd = sys_["Hpp"], sys_["bp"], sys_["hpd"], sys_["D"], sys_["bd"]
    Dd = D * (1 + lam) + depth_prior * sys_["seg_valid"]
    inv = np.where(Dd > 1e-12, 1.0 / np.where(Dd > 1e-12, Dd, 1.0), 0.0)
    if pose_only:
        inv = np.zeros_like(inv)
    S = Hpp - (hpd * inv[:, None]).T @ hpd
    S = S + np.diag(lam * np.diag(Hpp) + 1e-12)
    rhs = -(bp - (hpd * inv[:, None]).T @ bd)
    try:
        dxi = np.linalg.solve(S, rhs)
    except np.linalg.LinAlgError:
        dxi = np.zeros(6)
    dd = (-bd - hpd @ dxi) * inv
    return dxi, np.clip(dd, -depth_clamp, depth_clamp)


def run_schedule(model, phases, lam0=1e-4, lm_up=8.0, lm_down=0.5, lm_min=1e-7, verbose=False, valid_guard=0.0):
    """phases: list of dict(level, stride, max_iters, irls_eps, conv_tol, [depth_clamp, depth_prior, pose_only]).  Mirrors solve_gn's
    accept / undo / phase-advance logic (one cost evaluation per iteration, the step's effect is judged by the NEXT evaluation)."""
    pair = model.pair
    pose = pair.pose_init.astype(np.float64).copy()
    kld = pair.kld_init.astype(np.float64).copy()
    lam = lam0
    total = 0
    for ph in phases:
        last, rejected_prev, it = -1.0, False, 0
        last_valid = None
        backup = (pose.copy(), kld.copy())
        while it < ph["max_iters"]:
            s = model.system(ph["level"], ph["stride"], pose, kld, ph["irls_eps"])
            cost = s["cost"]
            total += 1
            worse = last >= 0 and cost > last * (1 + 1e-6) and not rejected_prev
            if valid_guard > 0 and last_valid is not None and not rejected_prev and s["n_valid"] < (1 - valid_guard) * last_valid:
                worse = True
            if worse:
                pose, kld = backup[0].copy(), backup[1].copy()
                lam *= lm_up
                rejected_prev = True
                it += 1
                if verbose:
                    print(f"   L{ph['level']}/s{ph['stride']} it {it:2d} cost {cost:.6f} REJECT lam -> {lam:.1e}")
                continue
            if last >= 0 and not rejected_prev and ph["conv_tol"] > 0 and (last - cost) <= ph["conv_tol"] * last:
                if verbose:
                    print(f"   L{ph['level']}/s{ph['stride']} converged after {it} iterations, cost {cost:.6f}")
                break
            if not rejected_prev:
                lam = max(lam * lm_down, lm_min)
            backup = (pose.copy(), kld.copy())
            dxi, dd = lm_step(s, lam, ph.get("depth_clamp", 0.5), ph.get("depth_prior", 0.0), ph.get("pose_only", False))
            if "pose_clamp" in ph:
                nrm = np.linalg.norm(dxi)
                if nrm > ph["pose_clamp"]:
                    dxi *= ph["pose_clamp"] / nrm
            pose = se3_exp(dxi) @ pose
            kld = kld + dd
            last, rejected_prev, last_valid = cost, False, s["n_valid"]
            it += 1
            if verbose:
                e = pose_depth_errors(pose, kld, pair.pose_gt, pair.kld_gt)
                print(f"   L{ph['level']}/s{ph['stride']} it {it:2d} cost {cost:.6f} lam {lam:.1e} valid {s['n_valid'] / s['n']:.3f} |dxi| {np.linalg.norm(dxi):.4f} "
                      f"max|dd| {np.abs(dd).max():.3f} -> err {e[0]:.4f} {e[1]:.4f} {e[2]:.3f}")
    return pose, kld, total


def default_phases(n_levels=3, strides=(1, 2, 4), max_iters=25, conv_tol=2e-3, polish_max=15, polish_eps=1e-5, polish_tol=1e-4, irls_eps=1e-3, **extra):
    ph = [dict(level=l, stride=strides[l], max_iters=max_iters, irls_eps=irls_eps, conv_tol=conv_tol, **extra) for l in reversed(range(n_levels))]
    ph.append(dict(level=0, stride=1, max_iters=polish_max, irls_eps=polish_eps, conv_tol=polish_tol))
    return ph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=501)
    ap.add_argument("--size", default="240x320x8")
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    H, W, N = (int(v) for v in a.size.split("x"))
    pair = synth.make_pair(H, W, N, seed=a.seed, overlap=3 if W <= 320 else 4, init_sigma=0.05, texture="octaves", init_mode="reference")
    strides = tuple(2 ** l for l in range(a.levels))
    model = PairModel(pair, a.levels, strides)
    pose, kld, n = run_schedule(model, default_phases(a.levels, strides), verbose=a.verbose)
    print("iterations", n, "errors vs gt", pose_depth_errors(pose, kld, pair.pose_gt, pair.kld_gt))


if __name__ == "__main__":
    main()
