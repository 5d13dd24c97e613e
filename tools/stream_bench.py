#!/usr/bin/env python
"""Developer tool: sustained frame pairs per second from raw device-resident frames over many batches back to back: one batch at a
time, against PairStream (set-up of the next batch overlapped with the optimisation of the current one), against PairStream with
several optimiser streams (continuous batching: the bulk of the next batch fills the tail of the current one)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import FRAME_PAIR_POINT_STRIDE, FRAME_PAIR_SCHEDULE, PairBatch
from super_primitive_amd.optim.pair_stream import PairStream

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base = [synth.make_pair(480, 640, 64, seed=1000 + s, overlap=4, init_sigma=0.004) for s in range(4)]
sk = {k: v for k, v in FRAME_PAIR_SCHEDULE.items() if k != "check_every"}


def make_items(per_batch, n_batches, distinct_batches):
    items = []
    for b in range(distinct_batches):
        prs = [base[(b * per_batch + i) % len(base)] for i in range(per_batch)]
        items.append(dict(src_frames=[KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in prs],
                          trg_images=[t(p.trg_image) for p in prs], trg_Ks=[t(p.K) for p in prs],
                          poses=torch.stack([t(p.pose_init) for p in prs]), klds=[t(p.kld_init) for p in prs]))
    return [items[i % distinct_batches] for i in range(n_batches)]


for per_batch, n_batches, distinct in ((384, 8, 2), (128, 24, 3), (64, 24, 3)):
    items = make_items(per_batch, n_batches, distinct)
    pipes = {f"pipelined, {k} optimiser stream{'s' if k > 1 else ''}": PairStream(levels=(0, 3), schedule=FRAME_PAIR_SCHEDULE, optimisers=k, depth=max(1, k - 1))
             for k in (1, 2, 3)}                                            # long-lived: their streams' allocator pools are reused
    for label in ("sequential",) + tuple(pipes):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if label == "sequential":
                for it in items:
                    b = PairBatch(it["src_frames"], it["trg_images"], it["trg_Ks"], it["poses"], it["klds"], levels=(0, 3),
                                  point_stride=FRAME_PAIR_POINT_STRIDE)
                    b.run_scheduled(**sk)
                    res = (b.poses().clone(), [k.clone() for k in b.klds()])
            else:
                for res in pipes[label].run(iter(items)):
                    pass
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{n_batches} batches x {per_batch} pairs, {label}: {dt * 1e3:.1f} ms = {n_batches * per_batch / dt:.0f} pairs/s", flush=True)
    del items
