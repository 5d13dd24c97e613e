#!/usr/bin/env python
"""Developer tool: per-pass HIP-event times of the batched set-up (optim/batch_prepare.py) for 384 distinct 640x480x64 pairs, e.g.
under different SP_FILL_VARIANT settings:  SP_FILL_VARIANT=64 python tools/setup_bench.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.batch_prepare import _Timer
from super_primitive_amd.optim.pair_batch import FRAME_PAIR_POINT_STRIDE, PairBatch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base = [synth.make_pair(480, 640, 64, seed=7000 + s, overlap=4, init_sigma=0.004) for s in range(4)]
pairs = [base[i % len(base)] for i in range(G)]
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in pairs]
if os.environ.get("SP_BOXES"):                 # the segment-box hint on every keyframe (KeyFrame.segment_boxes)
    from super_primitive_amd.optim.batch_prepare import segment_boxes_of
    for kf in src:
        kf.segment_boxes = segment_boxes_of(kf.keypoint_regions)
trg, Ks, klds = [t(p.trg_image) for p in pairs], [t(p.K) for p in pairs], [t(p.kld_init) for p in pairs]
poses = torch.stack([t(p.pose_init) for p in pairs])
GRAN = int(os.environ.get('SP_GRANULE', '256'))
build = lambda tm=None: PairBatch(src, trg, Ks, poses, klds, levels=(0, 3), point_stride=FRAME_PAIR_POINT_STRIDE, timer=tm, granule=GRAN)
build(); build()
acc = {}
wall = []
for _ in range(4):
    tm = _Timer()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = build(tm)
    torch.cuda.synchronize(); wall.append(time.perf_counter() - t0)
    for k, v in tm.milliseconds().items():
        acc.setdefault(k, []).append(v)
nb = b.setup_bytes
print(f"granule {GRAN}: {G} pairs, set-up {1e3 * np.median(wall):.2f} ms = {1e6 * np.median(wall) / G:.1f} us/pair; "
      + "  ".join(f"{k} {np.median(v):.2f} ms ({nb[k] / np.median(v) / 1e6 / 8000:.3f})" for k, v in acc.items()))
ev, marks = tm.timeline()
print("timeline of the last build (ms): GPU " + "  ".join(f"{n} {a:.2f}-{b:.2f}" for n, a, b in ev) + " | host " + "  ".join(f"{n} {t:.2f}" for n, t in marks)
      + f" | wall {1e3 * wall[-1]:.2f}")
