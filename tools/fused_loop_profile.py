#!/usr/bin/env python
"""Developer tool: run the fused driver loops long enough for `rocprofv3 --kernel-trace --stats` to attribute their time.
    rocprofv3 --kernel-trace --stats -d out -o fused -- python tools/fused_loop_profile.py [sfm|map|track]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.odometery.loops import map_window, track_frame_fused
from super_primitive_amd.odometery.two_frame_sfm import SfM

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "sfm"
if which == "sfm":
    for (H, W, N) in ((240, 320, 8), (480, 640, 64)):
        p = synth.make_pair(H, W, N, seed=1, init_sigma=0.01, overlap=3)
        src = KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions))
        trg = KeyFrame(t(p.trg_image), t(p.K))
        sfm = SfM({"aligment": {"pyramid_min": 0, "pyramid_max": 3, "cost_params": {}}}, src, [trg], [t(p.pose_init)], num_iters=1000)
        sfm.init_optimisation(kld_init=t(p.kld_init))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sfm.run(fused=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"sfm {W}x{H}x{N}: {3000 / dt:.0f} it/s ({1e6 * dt / 3000:.1f} us/iteration)")
elif which == "map":
    frames, est, klds, affs = synth.window_inputs(300, 3, H=224, W=288, N=40)
    kfs = [KeyFrame(t(f.image), t(f.K), t(f.logdepth_perseg), t(f.keypoints), t(f.keypoint_regions)) for f in frames[0::2]]
    supp = [[(KeyFrame(t(frames[2 * k + 1].image), t(frames[2 * k + 1].K)), t(est[2 * k + 1]), t(affs[2 * k + 1]))] for k in range(3)]
    args = (kfs, [t(est[2 * k]) for k in range(3)], [t(k) for k in klds], [t(affs[2 * k]) for k in range(3)], supp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = map_window(*args, 3000, window_size=3, initialised=False, fused=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"map 3 KF + 3 supp 224x288x40: {3000 / dt:.0f} it/s ({1e6 * dt / 3000:.1f} us/iteration)")
