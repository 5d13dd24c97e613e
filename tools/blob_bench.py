#!/usr/bin/env python
"""Developer tool: GN step throughput on ragged, overlapping segments (synth shape='blobs': random ellipses like SAM
masks) next to the bench's rectangular tiling, same image size and segment count."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import PairBatch

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for shape, N in (("grid", 64), ("blobs", 64), ("blobs", 128)):
    pairs = [synth.make_pair(480, 640, N, seed=10 + k, shape=shape, overlap=4 if shape == "grid" else 0, init_sigma=0.004) for k in range(4)]
    src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in pairs]
    R = 48
    b = PairBatch(src, [t(p.trg_image) for p in pairs], [t(p.K) for p in pairs], torch.stack([t(p.pose_init) for p in pairs]),
                  [t(p.kld_init) for p in pairs], levels=(0, 1), replicate=R)
    for _ in range(300):
        b.gn_step(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        b.gn_step(0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    by = b.algorithmic_bytes(0)
    sizes = np.concatenate([p.keypoint_regions.reshape(N, -1).sum(1) for p in pairs])
    print(f"{shape:5s} N={N}: {b.M} pairs, P/pair {np.mean(b.Ps):.0f} (padded {np.mean(b.Ppads):.0f}), segment sizes min/median/max "
          f"{sizes.min()}/{int(np.median(sizes))}/{sizes.max()}, spans {b.n_spans}: {dt*1e6:.0f} us/step, {b.M/dt:.0f} GN it/s, "
          f"{by/dt/1e12:.2f} TB/s algorithmic = {by/dt/8e12*100:.1f} % of 8 TB/s")
