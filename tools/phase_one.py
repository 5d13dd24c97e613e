#!/usr/bin/env python
"""Developer tool: ONE phase of the reference-start schedule on 1536 resident pairs, all in that phase (for counter passes: tools/phase_traffic.sh):
    python tools/phase_one.py <level> <stride> [pose_only]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, PairBatch

level, stride = int(sys.argv[1]), int(sys.argv[2])
args = bench.parse(["--no-cpu-baseline", "--no-pmc"])
dev = torch.device("cuda", 0)
G, M = 8, 4 * args.pairs
scenes = [bench._render_sigma05((args.segments, 5000 + s, args.shape, args.coverage)) for s in range(G)]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in scenes]
batch = PairBatch(src, [t(p.trg_image) for p in scenes], [t(p.K) for p in scenes], torch.from_numpy(np.stack([p.pose_init for p in scenes] * (M // G))),
                  [t(p.kld_init) for p in scenes] * (M // G), levels=REFERENCE_START_LEVELS, replicate=M // G, point_stride=REFERENCE_START_POINT_STRIDE, granule=args.granule)
IT = 16
ph = dict(level=level, stride=stride, max_iters=IT, irls_eps=1e-3, conv_tol=0.0, pose_only=len(sys.argv) > 3)
lay = batch.coarse[(level, stride)] if stride > 1 else None
pts = np.asarray(lay.points if lay is not None else batch.Ps, dtype=np.float64)
hw = np.asarray(batch.level_hw[level], dtype=np.float64)
nbytes = float((20.0 * pts + 12.0 * hw[:, 0] * hw[:, 1]).sum())
for rep in range(2):
    batch.restore_initial()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds = batch.run_scheduled(phases=[ph], verdict=False, check_every=8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"level {level} stride {stride}: {rounds} rounds, {1e3 * dt / rounds:.3f} ms per round, algorithmic bytes per cost launch {nbytes / 1e6:.1f} MB ({nbytes / M / 1e6:.2f} MB per pair)")
