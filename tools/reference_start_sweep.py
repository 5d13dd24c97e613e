#!/usr/bin/env python
"""Developer tool: the pose-only first phase of REFERENCE_START_SCHEDULE (iteration cap, IRLS epsilon) on bench.py's own reference-start
leg -- 384 pairs resident and 1536 pairs on 384 slots: frame pairs per second, converged fraction, iterations per pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import super_primitive_amd.optim.pair_batch as pb

args = bench.parse(["--no-cpu-baseline"])
dev = torch.device("cuda:0")
for name, extra in (("cap 15, eps 1e-3 (round 3)", dict(pose_first_iters=15)), ("cap 30, eps 1e-3 (shipped)", dict(pose_first_iters=30)),
                    ("cap 15, eps 1e-2", dict(pose_first_iters=15, pose_first_eps=1e-2)), ("cap 25, eps 1e-2", dict(pose_first_iters=25, pose_first_eps=1e-2)),
                    ("cap 25, eps 3e-3", dict(pose_first_iters=25, pose_first_eps=3e-3))):
    pb.REFERENCE_START_SCHEDULE = dict(pb.FRAME_PAIR_SCHEDULE, **extra)
    r = bench.reference_start_leg(args, 0, dev, 384)
    q = r["slot_level_continuous_batching"]
    print(f"pose-only {name}: 384 resident {r['frame_pairs_per_sec']:.0f} pairs/s, converged {r['converged_fraction']:.4f}, {r['iterations_per_pair']['mean']:.1f} iterations per pair | "
          f"1536 on 384 slots {q['frame_pairs_per_sec']:.0f} pairs/s, converged {q['converged_fraction']:.4f} ({round((1 - q['converged_fraction']) * q['pairs'])} lost), "
          f"{q['iterations_per_pair']['mean']:.1f} iterations per pair, worst converged error {q['worst_error_of_converged_vs_ground_truth']}", flush=True)
