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
scenes = [synth.make_pair(480, 640, 64, seed=1000 + s, overlap=4, init_sigma=0.004) for s in range(4)]
sk = {k: v for k, v in FRAME_PAIR_SCHEDULE.items() if k != "check_every"}


def make_items(per_batch, n_batches, distinct_batches):
    items = []
    for b in range(distinct_batches):
        prs = [scenes[(b * per_batch + i) % len(scenes)] for i in range(per_batch)]
        items.append(dict(src_frames=[KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in prs],
                          trg_images=[t(p.trg_image) for p in prs], trg_Ks=[t(p.K) for p in prs],
                          poses=torch.stack([t(p.pose_init) for p in prs]), klds=[t(p.kld_init) for p in prs]))
    return [items[i % distinct_batches] for i in range(n_batches)]


import gc

CONFIGS = ((384, 8, 2), (128, 16, 3))
if len(sys.argv) > 1:                  # e.g.  stream_bench.py 384 [distinct input sets]   (only the batch size given)
    CONFIGS = tuple(c for c in CONFIGS if c[0] == int(sys.argv[1]))
if len(sys.argv) > 2:
    CONFIGS = tuple((c[0], c[1], int(sys.argv[2])) for c in CONFIGS)
GRAN = int(os.environ.get('SP_GRANULE', '64'))
for per_batch, n_batches, distinct in CONFIGS:
    items = make_items(per_batch, n_batches, distinct)
    for k in (0, 1, 2, 3):
        label = "sequential" if k == 0 else f"pipelined, {k} optimiser stream{'s' if k > 1 else ''}"
        # one long-lived PairStream at a time (its streams' allocator pools are reused from batch to batch; the pools of a
        # discarded one are returned to the device before the next configuration runs)
        pipe = None if k == 0 else PairStream(levels=(0, 3), schedule=FRAME_PAIR_SCHEDULE, optimisers=k, depth=max(1, k - 1), granule=GRAN)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if pipe is None:
                for it in items:
                    b = PairBatch(it["src_frames"], it["trg_images"], it["trg_Ks"], it["poses"], it["klds"], levels=(0, 3),
                                  point_stride=FRAME_PAIR_POINT_STRIDE, granule=GRAN)
                    b.run_scheduled(**sk)
                    res = (b.poses().clone(), [k_.clone() for k_ in b.klds()])
                del b
            else:
                for res in pipe.run(iter(items)):
                    pass
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{n_batches} batches x {per_batch} pairs, {label}: {dt * 1e3:.1f} ms = {n_batches * per_batch / dt:.0f} pairs/s "
              f"(device memory reserved {torch.cuda.memory_reserved() / 2**30:.0f} GiB)", flush=True)
        if pipe is not None and pipe.trace:
            tr = pipe.trace[-2 * n_batches:]
            base = min(t[2] for t in tr)
            for what, idx, a, b_ in sorted(tr, key=lambda t: t[2]):
                print(f"      {what:9s} batch {idx}: {1e3 * (a - base):8.1f} -> {1e3 * (b_ - base):8.1f} ms ({1e3 * (b_ - a):.1f})")
        del pipe, res
        gc.collect()
        torch.cuda.empty_cache()
    del items
    gc.collect()
    torch.cuda.empty_cache()
