#!/usr/bin/env python
"""Developer tool: the g20y pairs (ragged masks, reference converges) ALONE under schedule variants -- outcomes near a basin boundary flip with the
summation order, so a pair is probed alone, as a batch of two and of three copies (other span partitions)."""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth
from super_primitive_amd.optim.pair_batch import REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, REFERENCE_START_SCHEDULE, PairBatch
from parity_util import pose_depth_errors
BASE = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
ct, ie = BASE["conv_tol"], 1e-3
po = lambda cap, eps=ie: dict(level=2, stride=4, max_iters=cap, irls_eps=eps, conv_tol=ct, pose_only=True)
jt = lambda level, stride, damp=0.0, cap=25: dict(level=level, stride=stride, max_iters=cap, irls_eps=ie, conv_tol=ct, depth_damp=damp)
pol = dict(level=0, stride=1, max_iters=15, irls_eps=1e-5, conv_tol=1e-4)
def sched(damps, first=po(15, 1e-2)):
    return [first] + [jt(2, 4, d, c) for d, c in damps] + [jt(2, 4), jt(1, 2), jt(0, 2), pol]
VARIANTS = {
    "shipped": dict(BASE),
    "d16c16": dict(BASE, coarse_damped=(16.0, 16)),
    "d31c12": dict(BASE, coarse_damped=(31.0, 12)),
    "d8c12": dict(BASE, coarse_damped=(8.0, 12)),
    "d16c12+d4c8": dict(BASE, phases=sched([(16.0, 12), (4.0, 8)])),
    "d31c8+d4c8": dict(BASE, phases=sched([(31.0, 8), (4.0, 8)])),
    "d16c12,L1 d1": dict(BASE, phases=[po(15, 1e-2), jt(2, 4, 16.0, 12), jt(2, 4), jt(1, 2, 1.0, 8), jt(1, 2), jt(0, 2), pol], retry_phases=None),
    "pose cap 30": dict(BASE, pose_first_iters=30),
    "pose eps 1e-3 cap 30": dict(BASE, pose_first_iters=30, pose_first_eps=None),
    "retry damped too": dict(BASE, retry_phases=[po(30), jt(2, 4, 31.0, 12)], retry_join=1),
}
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g20y_sigma05_blobs_pair*.npz"))):
    gx = np.load(path)
    pair = synth.make_pair(480, 640, 64, seed=int(gx["scene_seed"]), init_sigma=0.05, texture="octaves", init_mode="reference", shape="blobs", blob_coverage=1.2)
    pair.pose_init, pair.kld_init = gx["pose_init"].copy(), gx["kld_init"].copy()
    if "pair2219" in path and len(sys.argv) < 2:
        continue
    for name, kw in VARIANTS.items():
        out = []
        for copies in (1, 2, 3, 5):
            b = PairBatch.from_synth([pair] * copies, levels=REFERENCE_START_LEVELS, point_stride=REFERENCE_START_POINT_STRIDE, granule=64)
            b.run_scheduled(**kw)
            e = pose_depth_errors(b.poses()[0].double().cpu().numpy(), b.klds()[0].double().cpu().numpy(), gx["final_pose"], gx["final_kld"])
            out.append(f"x{copies}: {'ok ' if e[0] < 1e-4 else 'BAD'} {int(b.status[0]):#x} a{int(b.attempts[0])}")
        print(f"pair {int(gx['pair_index'])} {name:22s} " + " | ".join(out), flush=True)
