#!/usr/bin/env python
"""Developer tool: what it costs to make one keyframe / frame pair ready for the optimiser at 640x480x64 --
segment table, pyramid, per-level source sampling and target packing -- i.e. everything outside the iteration loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame, keyframe_pyramid
from super_primitive_amd.segment_table import SegmentTable, packed_target, table_of
from super_primitive_amd.optim.pair_batch import PairBatch

dev = torch.device("cuda:0")
p = synth.make_pair(480, 640, 64, seed=1, overlap=4)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
img_s, img_t, K, L, kp, M = t(p.src_image), t(p.trg_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"segment table (masks {tuple(M.shape)} -> P points, tiles): {timed(lambda: SegmentTable(M, L, kp)):.3f} ms")
print(f"3-level pyramids of both frames (keyframe_pyramid x2): {timed(lambda: (keyframe_pyramid(KeyFrame(img_s, K, L, kp, M), 0, 3), keyframe_pyramid(KeyFrame(img_t, K), 0, 3))):.3f} ms")
tab = SegmentTable(M, L, kp)
kld = t(p.kld_init)
print(f"source sampling of one level (src4): {timed(lambda: tab.sample_source(img_s, K) if hasattr(tab, 'sample_source') else tab.source_level(img_s.clone(), K, kld)):.3f} ms")
print(f"target packing of one level (HWC3): {timed(lambda: packed_target(img_t.clone())):.3f} ms")
def whole():
    src = KeyFrame(img_s, K, L, kp, M)
    b = PairBatch([src], [img_t], [K], t(p.pose_init)[None], [kld], levels=(0, 3), tile_points=2048)
    return b
print(f"PairBatch of one pair from raw tensors (table + pyramids + 3 levels of src4/trg3 + descriptors): {timed(whole, 10):.3f} ms")
b = whole()
print(f"30 GN iterations (3 levels x 10) as hipGraphs: {timed(lambda: b.run(10, mode='gn', use_graph=True), 10):.3f} ms")
