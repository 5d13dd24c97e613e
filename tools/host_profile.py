#!/usr/bin/env python
"""Developer tool: interpreter time of a PairBatch build from device-resident frames (384 distinct 640x480x64 pairs), by function."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.batch_prepare import _Timer
from super_primitive_amd.optim.pair_batch import FRAME_PAIR_POINT_STRIDE, PairBatch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 384
REPS = 5
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base = [synth.make_pair(480, 640, 64, seed=7000 + s, overlap=4, init_sigma=0.004) for s in range(4)]
pairs = [base[i % len(base)] for i in range(G)]
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in pairs]
trg, Ks, klds = [t(p.trg_image) for p in pairs], [t(p.K) for p in pairs], [t(p.kld_init) for p in pairs]
poses = torch.stack([t(p.pose_init) for p in pairs])
build = lambda tm=None: PairBatch(src, trg, Ks, poses, klds, levels=(0, 3), point_stride=FRAME_PAIR_POINT_STRIDE, timer=tm, granule=64)
build(); build()
torch.cuda.synchronize()
host, busy = [], []
for _ in range(REPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = build()
    host.append(time.perf_counter() - t0)
    busy.append(host[-1] - b.setup_host_wait_s)
    torch.cuda.synchronize()
print(f"{G} pairs: the constructor returns after {1e3 * np.median(host):.2f} ms (median of {REPS}; min {1e3 * min(host):.2f}), of which the interpreter is busy "
      f"{1e3 * np.median(busy):.2f} ms (min {1e3 * min(busy):.2f}) and waits for the counts the rest")
tm = _Timer()
b = build(tm)
torch.cuda.synchronize()
ev, marks = tm.timeline()
print("host marks (ms): " + "  ".join(f"{n} {v:.2f}" for n, v in marks))
pr = cProfile.Profile()
pr.enable()
for _ in range(REPS):
    b = build()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
