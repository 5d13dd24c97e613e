#!/usr/bin/env python
"""Developer tool (round 6): what one ROUND costs when a single pair is left in a scheduled run -- the tail of a batch whose last pair is in its
third attempt (1500 Adam iterations, one after the other).  A batch of ONE pair has 64-point spans (every wave a trip or two), a batch of
1536 has 3840-point spans on the coarse lattice (60 trips one after the other in ONE wave): the difference is what finer spans for the tail
would buy; the poll interval is the other knob.   python tools/lone_pair_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from super_primitive_amd.optim.pair_batch import REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, PairBatch

args = bench.parse(["--no-cpu-baseline", "--no-pmc", "--shape", "blobs"])
p = bench._render_sigma05((64, 5000, "blobs", 1.2))
sync = torch.cuda.synchronize
IT = 400
for copies in ((1,) if os.environ.get("SP_LONE_ONLY") else (1, 8, 64, 512)):
    b = PairBatch.from_synth([p] * copies, levels=REFERENCE_START_LEVELS, point_stride=REFERENCE_START_POINT_STRIDE, granule=64)
    for name, spec in (("Adam L2 stride 4", dict(level=2, stride=4, adam=True)), ("Adam L0 stride 2", dict(level=0, stride=2, adam=True)), ("GN joint L2 stride 4", dict(level=2, stride=4))):
        ph = dict(spec, max_iters=IT, irls_eps=1e-5, conv_tol=0.0)
        lay = b.coarse[(ph["level"], ph["stride"])]
        for ce in (4, 32):
            for rep in range(2):
                b.restore_initial()
                # every pair but the first is finished from the start: phase = n_phases (a scheduled run of ONE active pair among `copies` slots)
                sync(); t0 = time.perf_counter()
                if copies == 1:
                    rounds = b.run_scheduled(phases=[ph], verdict=False, check_every=ce)
                else:
                    # all resident, but only pair 0 gets the budget: the others converge at once (conv_tol large) -- emulate by a queue of one slot
                    rounds = b.run_scheduled(phases=[ph], verdict=False, check_every=ce, slots=copies // 2 if copies > 1 else None) if False else b.run_scheduled(phases=[ph], verdict=False, check_every=ce)
                sync(); dt = time.perf_counter() - t0
            print(f"{copies:4d} identical pairs, {name:22s} spans of {int(np.diff(lay.s_off).max())} x <= {int(lay.spans[:, 2].max())} points, poll every {ce:2d}: {1e6 * dt / rounds:7.1f} us per round ({rounds} rounds)", flush=True)
    del b
