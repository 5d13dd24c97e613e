#!/usr/bin/env python
"""Developer tool: the KERNELS of one round of a schedule phase timed apart with HIP events -- the cost pass of every resident pair in that phase
(sp_pairs_schedule_cost) and the solver (sp_pairs_schedule_gn_step) -- per point, next to the vector-issue floor (roofline.valu_ceiling of the
bench line: ~150 wave instructions per 64 points x 4 cycles / 1024 SIMDs).  Splits what tools/phase_cost.py reports per round into the cost
kernel's own efficiency on the lattice and the fixed cost of a round (solver latency, launch gaps, polls).   python tools/phase_kernel_time.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from super_primitive_amd import _lib, synth
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.optim.pair_batch import REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, PairBatch

args = bench.parse(["--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
G = 8
scenes = [bench._render_sigma05((args.segments, 5000 + s, args.shape, args.coverage)) for s in range(G)]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in scenes]
for M in (768, 1536):
    rng = np.random.default_rng(77)
    poses, klds = [], []
    for r in range(M // G):
        for p in scenes:
            poses.append((p.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.004 * rng.standard_normal(6))).astype(np.float32))
            klds.append((p.kld_gt + 0.01 * rng.standard_normal(p.N)).astype(np.float32))
    batch = PairBatch(src, [t(p.trg_image) for p in scenes], [t(p.K) for p in scenes], torch.from_numpy(np.stack(poses)), [t(k) for k in klds],
                      levels=REFERENCE_START_LEVELS, replicate=M // G, point_stride=REFERENCE_START_POINT_STRIDE, granule=args.granule)
    lib = batch.lib
    print(f"{M} resident pairs ({args.shape})")
    for name, spec in (("pose-only L2 stride 4", dict(level=2, stride=4, pose_only=True)), ("joint L2 stride 4", dict(level=2, stride=4)), ("joint L1 stride 2", dict(level=1, stride=2)),
                       ("joint L0 stride 2", dict(level=0, stride=2)), ("polish L0 all points", dict(level=0, stride=1))):
        ph = dict(spec, max_iters=1000000, irls_eps=1e-3, conv_tol=0.0)
        sched = batch.schedule(phases=[ph])
        lay = batch.coarse[(ph["level"], ph["stride"])] if ph["stride"] > 1 else None
        pts = float(np.asarray(lay.points if lay is not None else batch.Ps, dtype=np.float64).sum())
        batch.restore_initial()
        batch.phase.zero_(); batch.phase_iters.zero_()
        s = _lib.stream_ptr()
        cost = lambda: _lib.check(lib.sp_pairs_schedule_cost(ctypes.addressof(sched), _lib.ptr(batch.phase), s), "cost")
        step = lambda: _lib.check(lib.sp_pairs_schedule_gn_step(ctypes.addressof(sched), batch.M, batch.max_N, 8.0, 0.5, 1e-7, _lib.ptr(batch.lm_state), _lib.ptr(batch.backup),
                                                                 _lib.ptr(batch._costs), _lib.ptr(batch.phase), _lib.ptr(batch.phase_iters), None, s), "step")
        for _ in range(3):
            cost(); step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        R = 40
        ev[0].record()
        for _ in range(R): cost()
        ev[1].record()
        for _ in range(R): step()
        ev[2].record()
        for _ in range(R): cost(); step()
        ev[3].record()
        torch.cuda.synchronize()
        tc, ts, tb = (ev[i].elapsed_time(ev[i + 1]) * 1e3 / R for i in range(3))
        print(f"  {name:22s}: cost pass {tc:7.1f} us = {1e6 * tc / pts:5.2f} ps per point; solver {ts:6.1f} us (back to back); cost + solver in turn {tb:7.1f} us = {1e6 * tb / pts:5.2f} ps per point "
              f"({pts / M:8.0f} points per pair)", flush=True)
    del batch
