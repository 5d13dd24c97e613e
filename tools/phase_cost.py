#!/usr/bin/env python
"""Developer tool (round 6): what ONE iteration of every phase of REFERENCE_START_SCHEDULE costs per pair --
1536 resident reference-start pairs, all in the same phase (a scheduled run whose only phase has a fixed budget and no
convergence test), all resident and through the slot queue -- next to that phase's algorithmic bytes: where the schedule's
0.46-of-the-level-0-pass comes from, lattice by lattice.  Then the whole schedule on 1 / 2 / 3 streams sharing the queue.
    python tools/phase_cost.py [--shape blobs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, REFERENCE_START_SCHEDULE, PairBatch

args = bench.parse(["--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
G, M = 8, 4 * args.pairs
scenes = [bench._render_sigma05((args.segments, 5000 + s, args.shape, args.coverage)) for s in range(G)]
rng = np.random.default_rng(77)
poses, klds = [], []
for r in range(M // G):
    for p in scenes:
        if r == 0:
            poses.append(p.pose_init); klds.append(p.kld_init)
        else:
            poses.append((p.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * rng.standard_normal(6))).astype(np.float32))
            klds.append(np.log(2.0 + 2.0 * rng.uniform(size=p.N)).astype(np.float32))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in scenes]
batch = PairBatch(src, [t(p.trg_image) for p in scenes], [t(p.K) for p in scenes], torch.from_numpy(np.stack(poses)), [t(k) for k in klds],
                  levels=REFERENCE_START_LEVELS, replicate=M // G, point_stride=REFERENCE_START_POINT_STRIDE, granule=args.granule)
sync = torch.cuda.synchronize
IT = 16
print(f"{M} resident pairs ({args.shape}); per phase: {IT} iterations of every pair, no convergence test")
ONLY = os.environ.get("SP_PHASES", "")
for name, spec in (("pose-only L2 stride 4", dict(level=2, stride=4, pose_only=True)), ("joint L2 stride 4", dict(level=2, stride=4)), ("joint L1 stride 2", dict(level=1, stride=2)),
                   ("joint L0 stride 2", dict(level=0, stride=2)), ("polish L0 all points", dict(level=0, stride=1)), ("Adam L2 stride 4", dict(level=2, stride=4, adam=True))):
    if ONLY and not any(k in name for k in ONLY.split(",")):
        continue
    ph = dict(spec, max_iters=IT, irls_eps=1e-3, conv_tol=0.0)
    lay = batch.coarse[(ph["level"], ph["stride"])] if ph["stride"] > 1 else None
    pts = np.asarray(lay.points if lay is not None else batch.Ps, dtype=np.float64)
    hw = np.asarray(batch.level_hw[ph["level"]], dtype=np.float64)
    nbytes = float((20.0 * pts + 12.0 * hw[:, 0] * hw[:, 1]).sum())
    for how, kw in (("all resident", {}), ("768 slots", dict(slots=768)), ("768 slots, 2 streams", dict(slots=768, streams=2))):
        for rep in range(2):
            batch.restore_initial()
            sync(); t0 = time.perf_counter()
            rounds = batch.run_scheduled(phases=[ph], verdict=False, check_every=8, **kw)
            sync(); dt = time.perf_counter() - t0
        print(f"  {name:24s} {how:22s} [{(lay.n_spans if lay is not None else batch.n_spans)} spans]: {1e6 * dt / (M * IT):7.3f} us per pair-iteration, {rounds} rounds; {pts.mean():9.0f} points, {nbytes / M / 1e6:6.2f} MB per pair-iteration "
              f"-> {nbytes * IT / dt / 1e12:5.2f} TB/s = {nbytes * IT / dt / 8e12:5.3f} of HBM; {1e12 * dt / (IT * pts.sum()):6.2f} ps per point", flush=True)
kw = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
for slots in (768, 384):
    for streams in (1, 2, 3):
        for rep in range(2):
            batch.restore_initial()
            sync(); t0 = time.perf_counter()
            rounds = batch.run_scheduled(slots=slots, streams=streams, **kw)
            sync(); dt = time.perf_counter() - t0
        st = batch.status.cpu().numpy()
        print(f"whole schedule, {slots} slots, {streams} stream(s): {M / dt:8.0f} pairs/s, {rounds} rounds (by stream {batch._queue_stats['rounds_by_stream']}), flagged {int((st & 0x23f) != 0).sum() if False else int(((st & 0x23f) != 0).sum())}", flush=True)
