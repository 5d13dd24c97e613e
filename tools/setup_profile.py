#!/usr/bin/env python
"""Developer tool: where the time goes when a PairBatch of G DISTINCT 640x480x64 pairs is built from device-resident frames
(masks, images, intrinsics, seeds) -- the set-up a pipeline pays for every new frame pair, outside the iteration loop."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import FRAME_PAIR_POINT_STRIDE, FRAME_PAIR_SCHEDULE, PairBatch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base = [synth.make_pair(480, 640, 64, seed=7000 + s, overlap=4, init_sigma=0.004) for s in range(min(G, 8))]
pairs = [base[i % len(base)] for i in range(G)]
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in pairs]   # distinct device copies
trg = [t(p.trg_image) for p in pairs]
Ks = [t(p.K) for p in pairs]
klds = [t(p.kld_init) for p in pairs]
poses = torch.stack([t(p.pose_init) for p in pairs])


def build(**kw):
    return PairBatch(src, trg, Ks, poses, klds, levels=(0, 3), **kw)


for kw in ({}, {"point_stride": FRAME_PAIR_POINT_STRIDE}):
    build(**kw)
    build(**kw)                    # (twice: the allocator's pools settle on the second pass)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b = build(**kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"PairBatch of {G} distinct pairs {kw}: {dt * 1e3:.1f} ms = {dt / G * 1e6:.0f} us per pair")
sk = {k: v for k, v in FRAME_PAIR_SCHEDULE.items() if k != "check_every"}
b.run_scheduled(**sk)
b.restore_initial()
torch.cuda.synchronize()
t0 = time.perf_counter()
b.run_scheduled(**sk)
torch.cuda.synchronize()
print(f"run_scheduled on them: {(time.perf_counter() - t0) / G * 1e6:.0f} us per pair")
pr = cProfile.Profile()
pr.enable()
build(point_stride=FRAME_PAIR_POINT_STRIDE)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
# GPU-side time of one build (events on the current stream; includes the host-side gaps while the GPU waits)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
t0 = time.perf_counter()
b = build(point_stride=FRAME_PAIR_POINT_STRIDE)
t1 = time.perf_counter()
e1.record()
torch.cuda.synchronize()
print(f"build: host returns after {(t1 - t0) * 1e3:.2f} ms, GPU done after {e0.elapsed_time(e1):.2f} ms")
