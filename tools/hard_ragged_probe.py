#!/usr/bin/env python
"""Developer tool: the ragged-mask reference starts the shipped schedule flags after both attempts (tools/verdict_sweep.py --shape blobs: pairs 2437,
9847, 8479 of 12288), under schedule variants, as a batch of one (64-point spans) and replicated 96 times (4096-point spans, a large batch's partition).
    python tools/hard_ragged_probe.py 2437 9847 8479"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth
from super_primitive_amd.optim.pair_batch import REFERENCE_START_SCHEDULE, PairBatch
from parity_util import pose_depth_errors

ids = [int(a) for a in sys.argv[1:]] or [2437, 9847, 8479]
G, N = 8, 64
need = sorted({m % G for m in ids})
scenes = {s: synth.make_pair(480, 640, N, seed=5000 + s, init_sigma=0.05, texture="octaves", init_mode="reference", shape="blobs", blob_coverage=1.2) for s in need}
rng = np.random.default_rng(77)
starts = {}
for r in range(1, max(ids) // G + 1):
    for s in range(G):
        xi, u = rng.standard_normal(6), rng.uniform(size=N)
        if r * G + s in ids:
            sc = scenes[s]
            starts[r * G + s] = ((sc.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * xi)).astype(np.float32), np.log(2.0 + 2.0 * u).astype(np.float32))
BASE = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
ct, ie = BASE["conv_tol"], 1e-3
po = lambda cap, eps=ie, level=2, stride=4: dict(level=level, stride=stride, max_iters=cap, irls_eps=eps, conv_tol=ct, pose_only=True)
jt = lambda level, stride, damp=0.0, cap=25, eps=ie: dict(level=level, stride=stride, max_iters=cap, irls_eps=eps, conv_tol=ct, depth_damp=damp)
pol = dict(level=0, stride=1, max_iters=15, irls_eps=1e-5, conv_tol=1e-4)
L4 = (2, 2, 4, 8)
VARIANTS = {
    "shipped": ((2, 2, 4), dict(BASE)),
    "L3 pose-only 15 + L2 15 first": (L4, dict(BASE, retry_phases=None, phases=[po(15, 1e-2, 3, 8), po(15, 1e-2), jt(2, 4, 31.0, 12), jt(2, 4), jt(1, 2), jt(0, 2), pol])),
    "L3 pose-only eps 1e-3": (L4, dict(BASE, retry_phases=None, phases=[po(15, ie, 3, 8), po(15, ie), jt(2, 4, 31.0, 12), jt(2, 4), jt(1, 2), jt(0, 2), pol])),
    "L3 pose-only + L3 damped": (L4, dict(BASE, retry_phases=None, phases=[po(15, 1e-2, 3, 8), jt(3, 8, 31.0, 12), po(15, 1e-2), jt(2, 4, 31.0, 12), jt(2, 4), jt(1, 2), jt(0, 2), pol])),
    "pose-only L2 eps 1e-1": ((2, 2, 4), dict(BASE, retry_phases=None, pose_first_eps=1e-1)),
    "pose-only L2 eps 3e-2 cap 25": ((2, 2, 4), dict(BASE, retry_phases=None, pose_first_eps=3e-2, pose_first_iters=25)),
    "joint eps 1e-2 at L2": ((2, 2, 4), dict(BASE, retry_phases=None, phases=[po(15, 1e-2), jt(2, 4, 31.0, 12, 1e-2), jt(2, 4, 0.0, 25, 1e-2), jt(2, 4), jt(1, 2), jt(0, 2), pol])),
    "damped 31 at L2 and L1 (12 each)": ((2, 2, 4), dict(BASE, retry_phases=None, phases=[po(15, 1e-2), jt(2, 4, 31.0, 12), jt(1, 2, 31.0, 12), jt(2, 4), jt(1, 2), jt(0, 2), pol])),
}
for m in ids:
    p = copy.copy(scenes[m % G]); p.pose_init, p.kld_init = starts[m]
    print(f"pair {m}: start error {pose_depth_errors(p.pose_init, p.kld_init, p.pose_gt, p.kld_gt)}", flush=True)
    for name, spec in VARIANTS.items():
        stride, kw = spec[0], spec[1]
        out = []
        for copies in (1,):
            b = PairBatch.from_synth([p] * copies, levels=(0, len(stride)), point_stride=stride, granule=64)
            if len(spec) > 2:
                b.reset_lm(spec[2])
            b.run_scheduled(**kw)
            e = pose_depth_errors(b.poses()[0].double().cpu().numpy(), b.klds()[0].double().cpu().numpy(), p.pose_gt, p.kld_gt)
            out.append(f"x{copies}: {'ok ' if e[0] < 2e-3 else 'BAD'} {int(b.status[0]):#x} a{int(b.attempts[0])} ({e[0]:.1e} rad, {int(b.lm_state[0, 2] + b.lm_state[0, 3])} it)")
            del b
        print(f"   {name:30s} " + " | ".join(out), flush=True)
