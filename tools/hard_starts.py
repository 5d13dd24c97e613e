#!/usr/bin/env python
"""Developer tool: the reference-start pairs of bench.py that the Gauss-Newton schedule loses (1 of 384, 9 of 1536: the 2-4 sigma tail of
the start distribution, rotation errors of 0.09-0.22 rad) under schedule variants.  Pairs are named like bench.py names them: pair m = scene
5000 + m % 8, replica m // 8 of default_rng(77)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth
from super_primitive_amd.optim.pair_batch import REFERENCE_START_SCHEDULE, PairBatch
from parity_util import pose_depth_errors

HARD = [105, 984, 1005, 1006, 1159, 1161, 1272, 1417]
EASY = [0, 1, 2, 3, 100, 500, 900, 1300]
G, N = 8, 64
scenes = [synth.make_pair(480, 640, N, seed=5000 + s, overlap=4, init_sigma=0.05, texture="octaves", init_mode="reference") for s in range(G)]
rng = np.random.default_rng(77)
starts = {}
for r in range(0, max(HARD + EASY) // G + 1):
    for s in range(G):
        if r == 0:
            starts[s] = (scenes[s].pose_init, scenes[s].kld_init)
        else:
            xi, u = rng.standard_normal(6), rng.uniform(size=N)
            starts[r * G + s] = ((scenes[s].pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * xi)).astype(np.float32), np.log(2.0 + 2.0 * u).astype(np.float32))
ids = HARD + EASY
import copy
pairs = []
for m in ids:
    p = copy.copy(scenes[m % G]); p.pose_init, p.kld_init = starts[m]; pairs.append(p)
base = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
ct, mi, ie = base["conv_tol"], base["max_iters_per_level"], 1e-3
pol = dict(level=0, stride=1, max_iters=base["polish_max"], irls_eps=base["polish_eps"], conv_tol=base["polish_tol"])
def tail(levels):
    stride = {3: 8, 2: 4, 1: 2, 0: 2}
    return [dict(level=l, stride=stride[l], max_iters=mi, irls_eps=ie, conv_tol=ct) for l in levels] + [pol]
def po(l, n, eps=ie, tol=ct):
    return dict(level=l, stride={3: 8, 2: 4, 1: 2, 0: 2}[l], max_iters=n, irls_eps=eps, conv_tol=tol, pose_only=True)
VARIANTS = {
    "shipped (pose-only 15 @L2)": ((0, 3), (2, 2, 4), [po(2, 15)] + tail([2, 1, 0])),
    "pose-only 30 @L2": ((0, 3), (2, 2, 4), [po(2, 30)] + tail([2, 1, 0])),
    "pose-only 15 @L2, tol 2e-4": ((0, 3), (2, 2, 4), [po(2, 15, tol=2e-4)] + tail([2, 1, 0])),
    "pose-only 15 @L3 + 15 @L2": ((0, 4), (2, 2, 4, 8), [po(3, 15), po(2, 15)] + tail([2, 1, 0])),
    "pose-only 15 @L3 + 15 @L2, joint from L3": ((0, 4), (2, 2, 4, 8), [po(3, 15), po(2, 15)] + tail([3, 2, 1, 0])),
    "pose-only 15 @L2 eps 1e-2": ((0, 3), (2, 2, 4), [po(2, 15, eps=1e-2)] + tail([2, 1, 0])),
    "pose-only 15 @L2, joint L2 eps 1e-2 first": ((0, 3), (2, 2, 4), [po(2, 15), dict(level=2, stride=4, max_iters=mi, irls_eps=1e-2, conv_tol=ct)] + tail([2, 1, 0])),
}
for name, (levels, stride, phases) in VARIANTS.items():
    batch = PairBatch.from_synth(pairs, levels=levels, point_stride=stride, granule=64)
    n = batch.run_scheduled(phases=phases)
    torch.cuda.synchronize()
    P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
    e = np.array([pose_depth_errors(P[i], K[i], pairs[i].pose_gt, pairs[i].kld_gt) for i in range(len(ids))])
    ok = (e[:, 0] <= 2e-3) & (e[:, 1] <= 2e-3) & (e[:, 2] <= 2e-2)
    its = (batch.lm_state[:, 2] + batch.lm_state[:, 3]).cpu().numpy()
    print(f"{name}: hard {int(ok[:len(HARD)].sum())}/{len(HARD)} {['ok' if o else 'FAIL' for o in ok[:len(HARD)]]}, easy {int(ok[len(HARD):].sum())}/{len(EASY)}; iterations hard {its[:len(HARD)].mean():.0f} easy {its[len(HARD):].mean():.0f}; rounds {n}", flush=True)
