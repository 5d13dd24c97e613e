#!/usr/bin/env python
"""Developer tool: speed of the drop-in Python API loop (SfM.run: photomeric_cost + autograd + torch Adam) at
640x480x64, i.e. what an unmodified reference driver gets, next to the fused on-device loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.odometery.two_frame_sfm import SfM

dev = torch.device("cuda:0")
p = synth.make_pair(480, 640, 64, seed=1, overlap=4, init_sigma=0.004)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions))
trg = KeyFrame(t(p.trg_image), t(p.K))
cfg = {"aligment": {"pyramid_min": 0, "pyramid_max": 3, "cost_params": {}}}
for stats, lazy in ((0, True), (2, True), (2, False)):
    cfg["aligment"]["cost_params"] = {"stats_lazy": lazy}
    sfm = SfM(cfg, src, [trg], [t(p.pose_init)], num_iters=5, collect_stats=stats)
    sfm.init_optimisation(kld_init=t(p.kld_init)); sfm.run(); torch.cuda.synchronize()
    n = 100
    sfm = SfM(cfg, src, [trg], [t(p.pose_init)], num_iters=n, collect_stats=stats)
    sfm.init_optimisation(kld_init=t(p.kld_init))
    t0 = time.perf_counter(); sfm.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"collect_stats={stats} ({'lazy, nobody looks' if lazy else 'eager'}): {3*n/dt:.0f} Adam it/s through the reference API ({dt/(3*n)*1e6:.0f} us/iter), final loss {float(sfm.losses[-1]):.5f}")
