#!/usr/bin/env python
"""Developer tool: the reference-start pairs of bench.py's leg that the shipped Gauss-Newton schedule loses (named in the bench line),
phase by phase: errors against the ground truth after every phase of REFERENCE_START_SCHEDULE, and under a few schedule variants.
    python tools/lost_starts.py 1380 1482"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth
from super_primitive_amd.optim.pair_batch import REFERENCE_START_SCHEDULE, PairBatch
from parity_util import pose_depth_errors

ids = [int(a) for a in sys.argv[1:]] or [1380, 1482]
G, N = 8, 64
scenes = [synth.make_pair(480, 640, N, seed=5000 + s, overlap=4, init_sigma=0.05, texture="octaves", init_mode="reference") for s in range(G)]
rng = np.random.default_rng(77)
starts = {}
for r in range(0, max(ids) // G + 1):
    for s in range(G):
        if r == 0:
            starts[s] = (scenes[s].pose_init, scenes[s].kld_init)
        else:
            xi, u = rng.standard_normal(6), rng.uniform(size=N)
            starts[r * G + s] = ((scenes[s].pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * xi)).astype(np.float32), np.log(2.0 + 2.0 * u).astype(np.float32))
pairs = []
for m in ids:
    p = copy.copy(scenes[m % G]); p.pose_init, p.kld_init = starts[m]; pairs.append(p)
base = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
ct, mi, ie = base["conv_tol"], base["max_iters_per_level"], 1e-3
stride = {3: 8, 2: 4, 1: 2, 0: 2}
pol = dict(level=0, stride=1, max_iters=base["polish_max"], irls_eps=base["polish_eps"], conv_tol=base["polish_tol"])
joint = lambda l, eps=ie, n=mi: dict(level=l, stride=stride[l], max_iters=n, irls_eps=eps, conv_tol=ct)
po = lambda l, n, eps=ie: dict(level=l, stride=stride[l], max_iters=n, irls_eps=eps, conv_tol=ct, pose_only=True)


def errors(batch):
    P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
    out = []
    for i, p in enumerate(pairs):
        e = pose_depth_errors(P[i], K[i], p.pose_gt, p.kld_gt)
        worst = int(np.argmax(np.abs(np.expm1(K[i] - p.kld_gt + np.mean(p.kld_gt - K[i])))))
        out.append(f"{ids[i]}: {e[0]:.1e} rad {e[1]:.1e} t {e[2]:.1e} d (segment {worst})")
    return " | ".join(out)


shipped = [po(2, base["pose_first_iters"]), joint(2), joint(1), joint(0), pol]
batch = PairBatch.from_synth(pairs, levels=(0, 3), point_stride=(2, 2, 4), granule=64)
print("start:", errors(batch))
for ph in shipped:
    batch.run_scheduled(phases=[ph])
    torch.cuda.synchronize()
    its = (batch.lm_state[:, 2] + batch.lm_state[:, 3]).cpu().numpy()
    print(f"after level {ph['level']} stride {ph['stride']}{' pose only' if ph.get('pose_only') else ''} (eps {ph['irls_eps']:g}): {errors(batch)}; iterations so far {its}")
VARIANTS = {
    "shipped in one run": shipped,
    "joint L2 with eps 1e-2, then shipped tail": [po(2, 30), joint(2, 1e-2), joint(2), joint(1), joint(0), pol],
    "pose-only 30 @L2 eps 1e-2": [po(2, 30, 1e-2), joint(2), joint(1), joint(0), pol],
    "pose-only again at L1 and L0 before each joint phase": [po(2, 30), joint(2), po(1, 10), joint(1), po(0, 10), joint(0), pol],
    "joint phases capped at 40": [po(2, 30), joint(2, n=40), joint(1, n=40), joint(0, n=40), pol],
}
for name, phases in VARIANTS.items():
    b = PairBatch.from_synth(pairs, levels=(0, 3), point_stride=(2, 2, 4), granule=64)
    b.run_scheduled(phases=phases)
    torch.cuda.synchronize()
    print(f"{name}: {errors(b)}")
# a fourth pyramid level in front (80x60, stride-8 lattice)
for name, phases in {"pose-only 15 @L3 + 15 @L2, then shipped tail": [po(3, 15), po(2, 15), joint(2), joint(1), joint(0), pol],
                     "pose-only 15 @L3 + 30 @L2, then shipped tail": [po(3, 15), po(2, 30), joint(2), joint(1), joint(0), pol],
                     "pose-only 15 @L3, joint L3, pose-only 15 @L2, then shipped tail": [po(3, 15), joint(3), po(2, 15), joint(2), joint(1), joint(0), pol]}.items():
    b = PairBatch.from_synth(pairs, levels=(0, 4), point_stride=(2, 2, 4, 8), granule=64)
    b.run_scheduled(phases=phases)
    torch.cuda.synchronize()
    its = (b.lm_state[:, 2] + b.lm_state[:, 3]).cpu().numpy()
    print(f"{name}: {errors(b)}; iterations {its}")
