#!/usr/bin/env python
"""Developer tool: does a level-0 phase on the stride-2 lattice in front of the all-points level-0 phase pay?  (The frame-pair rate is
bound by the cost kernel's throughput once the resident set is full -- slot-level batching -- so what is left is the WORK per pair.)
    python tools/phase_sweep.py [--pairs 384] [--reference-start]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import FRAME_PAIR_SCHEDULE, REFERENCE_START_SCHEDULE, PairBatch

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=384); ap.add_argument("--scenes", type=int, default=8)
ap.add_argument("--reference-start", action="store_true"); a = ap.parse_args()
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
G, R = a.scenes, a.pairs // a.scenes
if a.reference_start:
    scenes = [synth.make_pair(480, 640, 64, seed=5000 + s, overlap=4, init_sigma=0.05, texture="octaves", init_mode="reference") for s in range(G)]
    rng = np.random.default_rng(77)
    poses, klds = [], []
    for r in range(R):
        for p in scenes:
            if r == 0: poses.append(p.pose_init); klds.append(p.kld_init)
            else:
                poses.append((p.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * rng.standard_normal(6))).astype(np.float32))
                klds.append(np.log(2.0 + 2.0 * rng.uniform(size=p.N)).astype(np.float32))
    base = REFERENCE_START_SCHEDULE
else:
    scenes = [synth.make_pair(480, 640, 64, seed=1000 + s, overlap=4, init_sigma=0.004) for s in range(G)]
    rng = np.random.default_rng(0)
    poses = [(synth.se3_exp_np(0.002 * rng.standard_normal(6)) @ p.pose_init.astype(np.float64)).astype(np.float32) for r in range(R) for p in scenes]
    klds = [p.kld_init for r in range(R) for p in scenes]
    base = FRAME_PAIR_SCHEDULE
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in scenes]
batch = PairBatch(src, [t(p.trg_image) for p in scenes], [t(p.K) for p in scenes], torch.from_numpy(np.stack(poses)), [t(k) for k in klds], levels=(0, 3),
                  replicate=R, point_stride=(1, 2, 4), extra_tables=[(0, 2), (0, 4)], granule=64)
kw = {k: v for k, v in base.items() if k not in ("check_every", "pose_first_iters")}
ct, mi, ie = kw["conv_tol"], kw["max_iters_per_level"], 1e-3
pf = base.get("pose_first_iters", 0)
def phases(l0):
    ph = []
    if pf: ph.append(dict(level=2, stride=4, max_iters=pf, irls_eps=ie, conv_tol=ct, pose_only=True))
    ph += [dict(level=2, stride=4, max_iters=mi, irls_eps=ie, conv_tol=ct), dict(level=1, stride=2, max_iters=mi, irls_eps=ie, conv_tol=ct)]
    ph += l0
    return ph
full = dict(level=0, stride=1, max_iters=mi, irls_eps=ie, conv_tol=ct)
pol = dict(level=0, stride=1, max_iters=kw["polish_max"], irls_eps=kw["polish_eps"], conv_tol=kw["polish_tol"])
variants = {
    "shipped: L0 all points + polish all points": phases([full, pol]),
    "L0 stride 2, then L0 all points, polish all points": phases([dict(full, stride=2), full, pol]),
    "L0 stride 2, polish all points": phases([dict(full, stride=2), pol]),
    "L0 stride 2, polish stride 2, polish all points": phases([dict(full, stride=2), dict(pol, stride=2), pol]),
    "L0 stride 4, L0 stride 2, polish stride 2, polish all points": phases([dict(full, stride=4), dict(full, stride=2), dict(pol, stride=2), pol]),
}
def errors():
    P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
    e = np.zeros((batch.M, 3))
    for m in range(batch.M):
        gt = scenes[m % G]; ls = float(np.mean(gt.kld_gt - K[m])); Rm = P[m][:3, :3].T @ gt.pose_gt[:3, :3].astype(np.float64)
        e[m] = (np.arctan2(0.5 * np.linalg.norm([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]), 0.5 * (np.trace(Rm) - 1)),
                np.abs(P[m][:3, 3] * np.exp(ls) - gt.pose_gt[:3, 3]).max(), np.abs(np.expm1(K[m] + ls - gt.kld_gt)).max())
    return e
for rep in range(2):
  for name, ph in variants.items():
    for slots in (None, batch.M // 4):
        batch.restore_initial(); batch.run_scheduled(phases=ph, slots=slots); batch.restore_initial()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = batch.run_scheduled(phases=ph, slots=slots)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e = errors(); conv = (e[:, 0] <= 2e-3) & (e[:, 1] <= 2e-3) & (e[:, 2] <= 2e-2)
        its = (batch.lm_state[:, 2] + batch.lm_state[:, 3]).mean().item()
        if rep == 1:
            print(f"{name} [{'all resident' if slots is None else f'{slots} slots'}]: {batch.M / dt:.0f} pairs/s, {n} rounds, {its:.1f} iterations per pair, converged {conv.mean():.4f}, "
                  f"worst converged error {e[conv].max(axis=0)}", flush=True)
