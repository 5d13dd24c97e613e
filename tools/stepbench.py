#!/usr/bin/env python
"""Developer tool: per-kernel timing of one optimiser step (cost + solver) with HIP events."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from super_primitive_amd import _lib

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=96); ap.add_argument("--distinct", type=int, default=4)
ap.add_argument("--tile-points", type=int, default=8192); ap.add_argument("--segments", type=int, default=64); a = ap.parse_args()
dev = torch.device("cuda:0")
a.no_depth_table = True          # (the single-launch forms timed below read log-depth tables)
batch, _ = bench.build_batch(a, 0, dev)
for _ in range(5): batch.gn_step(0)
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for _ in range(20):
    e = [E() for _ in range(3)]
    e[0].record(); batch.cost_pass(0, 1); e[1].record()
    _lib.check(batch.lib.sp_pairs_gn_step(_lib.ptr(batch.desc[0]), batch.M, batch.max_N, _lib.ptr(batch.partials), _lib.ptr(batch.seg_partials), 8.0, 0.5, 1e-7,
               _lib.ptr(batch.lm_state), _lib.ptr(batch.backup), _lib.ptr(batch._costs), _lib.stream_ptr()), "gn"); e[2].record()
    rows.append(e)
torch.cuda.synchronize()
c = np.median([r[0].elapsed_time(r[1]) for r in rows]) * 1e3; s = np.median([r[1].elapsed_time(r[2]) for r in rows]) * 1e3
print(f"GN: cost kernel {c:.1f} us, solver {s:.1f} us")
rows = []
for _ in range(20):
    e = [E() for _ in range(3)]
    e[0].record(); batch.cost_pass(0, 0); e[1].record()
    _lib.check(batch.lib.sp_pairs_adam_step(_lib.ptr(batch.desc[0]), batch.M, batch.max_N, _lib.ptr(batch.partials), _lib.ptr(batch.seg_partials), 1e-3, 1e-2, 5e-3,
               _lib.ptr(batch.adam_state), _lib.ptr(batch._costs), _lib.stream_ptr()), "adam"); e[2].record()
    rows.append(e)
torch.cuda.synchronize()
c = np.median([r[0].elapsed_time(r[1]) for r in rows]) * 1e3; s = np.median([r[1].elapsed_time(r[2]) for r in rows]) * 1e3
print(f"Adam: cost kernel {c:.1f} us, solver {s:.1f} us")

import itertools
variants = (("GN fused", lambda: batch.gn_step(0, fused=True)), ("GN 2-launch", lambda: batch.gn_step(0, fused=False)),
            ("Adam fused", lambda: batch.adam_step(0, fused=True)), ("Adam 2-launch", lambda: batch.adam_step(0, fused=False)))
res = {n: [] for n, _ in variants}
for rnd in range(4):
    for name, fn in variants:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 40 * 1e6)
for n, v in res.items():
    print(f"{n}: " + " ".join(f"{x:.1f}" for x in v) + f"  us/iter (median {np.median(v):.1f})")
