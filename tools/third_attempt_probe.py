#!/usr/bin/env python
"""Developer tool (round 6): the ragged-mask reference starts the Gauss-Newton schedule loses at both attempts
(tools/verdict_sweep.py --shape blobs --starts 49152: 13 pairs), put through THE REFERENCE'S OWN OPTIMISER on the device
(PairBatch.run(mode="adam"): Adam at lr 1e-3 / 1e-2, odometery/two_frame_sfm.py:116-123) with several budgets, followed by the
Gauss-Newton tail -- which budget brings them home -- and, for the verdict of ONE pair, the per-segment mean |r| of wrong-basin end
states next to those of converged pairs.
    python tools/third_attempt_probe.py [ids...]"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_primitive_amd import synth
from super_primitive_amd.core import dense_optim
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import REFERENCE_START_SCHEDULE, PairBatch
from parity_util import pose_depth_errors

LOST = [2437, 8479, 9847, 12847, 18230, 18932, 21595, 23155, 24324, 32875, 35432, 41662, 44492]
ids = [int(a) for a in sys.argv[1:]] or LOST
G, N = 8, 64
extra = [m for m in range(8, 8 + 48) if m not in ids]           # converged company: the first replicas of every scene
want = set(ids) | set(extra)
scenes = {s: synth.make_pair(480, 640, N, seed=5000 + s, init_sigma=0.05, texture="octaves", init_mode="reference", shape="blobs", blob_coverage=1.2) for s in range(G)}
rng = np.random.default_rng(77)
starts = {}
for r in range(1, max(want) // G + 1):
    for s in range(G):
        xi, u = rng.standard_normal(6), rng.uniform(size=N)
        if r * G + s in want:
            sc = scenes[s]
            starts[r * G + s] = ((sc.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * xi)).astype(np.float32), np.log(2.0 + 2.0 * u).astype(np.float32))


def pairs_of(ms):
    out = []
    for m in ms:
        p = copy.copy(scenes[m % G]); p.pose_init, p.kld_init = starts[m]; out.append(p)
    return out


def errs(b, ps):
    P, K = b.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in b.klds()]
    return np.array([pose_depth_errors(P[i], K[i], p.pose_gt, p.kld_gt) for i, p in enumerate(ps)])


BASE = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
ct, ie = BASE["conv_tol"], 1e-3
jt = lambda level, stride, damp=0.0, cap=25, eps=ie: dict(level=level, stride=stride, max_iters=cap, irls_eps=eps, conv_tol=ct, depth_damp=damp)
pol = dict(level=0, stride=1, max_iters=15, irls_eps=1e-5, conv_tol=1e-4)
TAILS = {"L2,L1,L0+polish": [jt(2, 4), jt(1, 2), jt(0, 2), pol], "L1,L0+polish": [jt(1, 2), jt(0, 2), pol], "L0+polish": [jt(0, 2), pol]}
ADAM = {
    "reference budget 500/500/500": (500, 500, 500),
    "500 @L2 only": (500, 0, 0),
    "300 @L2 only": (300, 0, 0),
    "150 @L2 only": (150, 0, 0),
    "500 @L2 + 300 @L1": (500, 300, 0),
    "300 @L2 + 200 @L1": (300, 200, 0),
}
lost = pairs_of(ids)
print(f"{len(ids)} lost starts: {ids}", flush=True)
for name, budget in ADAM.items():
    for tail_name, tail in TAILS.items():
        if sum(1 for x in budget if x) == 3 and tail_name != "L0+polish":
            continue
        b = PairBatch.from_synth(lost, levels=(0, 3), point_stride=(2, 2, 4), granule=64)
        b.run(list(budget), mode="adam")
        e_adam = errs(b, lost)
        b.reset_lm()
        b.run_scheduled(phases=tail, verdict=dict(cost_outlier=0.0))
        torch.cuda.synchronize()
        e = errs(b, lost)
        ok = (e[:, 0] < 2e-3) & (e[:, 1] < 2e-3) & (e[:, 2] < 2e-2)
        st = b.status.cpu().numpy()
        print(f"Adam {name:30s} + GN {tail_name:16s}: {int(ok.sum())} of {len(ids)} home; after Adam worst rot {e_adam[:, 0].max():.1e} median {np.median(e_adam[:, 0]):.1e}; "
              f"not home: {[(ids[i], hex(int(st[i]))) for i in np.nonzero(~ok)[0]]}; worst of the home {e[ok].max(axis=0) if ok.any() else None}", flush=True)
        del b

# ---- per-segment costs: wrong-basin end states (the shipped schedule on the lost starts) next to converged pairs ----
def segment_costs(p, pose, kld):
    dev = pose.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    kf = KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions))
    trg = KeyFrame(t(p.trg_image), t(p.K))
    st = dense_optim.photomeric_cost(kf, trg, kld, pose, dict(mode="colour", collect_stats=1))
    seg = st["segm_ids"]
    mask = st["full_mask"].reshape(-1).float()
    r = st["residual_raw"].abs().sum(dim=1).reshape(-1) * mask
    n = int(seg.max()) + 1
    s = torch.zeros(n, device=dev).index_add_(0, seg, r)
    c = torch.zeros(n, device=dev).index_add_(0, seg, mask)
    return (s / (3 * c.clamp(min=1))).cpu().numpy(), c.cpu().numpy()


for label, ms in (("LOST (shipped schedule, both attempts)", ids), ("converged company", extra)):
    ps = pairs_of(ms)
    b = PairBatch.from_synth(ps, levels=(0, 3), point_stride=(2, 2, 4), granule=64)
    b.run_scheduled(**BASE, verdict=dict(cost_outlier=0.0))
    torch.cuda.synchronize()
    e = errs(b, ps)
    st = b.status.cpu().numpy()
    print(f"--- {label}")
    for i, m in enumerate(ms):
        c, cnt = segment_costs(ps[i], b.poses()[i].clone(), b.klds()[i].clone())
        big = cnt >= 64
        cs = np.sort(c[big])
        med = np.median(cs)
        print(f"pair {m}: status {int(st[i]):#x} err {e[i, 0]:.1e} rad | cost {float(b.diag[i, 0]):.3e} | per-segment mean|r|: min {cs[0]:.2e} median {med:.2e} p90 {cs[int(0.9 * len(cs))]:.2e} "
              f"max {cs[-1]:.2e} | max/median {cs[-1] / med:.1f} p90/median {cs[int(0.9 * len(cs))] / med:.1f} mean/median {c[big].mean() / med:.2f} "
              f"segments > 4 median: {int((cs > 4 * med).sum())} of {len(cs)}", flush=True)
    del b
