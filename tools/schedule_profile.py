#!/usr/bin/env python
"""The reference-start leg of bench.py alone (1536 resident pairs, 768 / 384 slots, REFERENCE_START_SCHEDULE), for
`rocprofv3 --kernel-trace --stats`: the per-kernel split of a SCHEDULED run at bench scale.
    python tools/schedule_profile.py [--shape blobs] [--pairs 384]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

argv = sys.argv[1:]
args = bench.parse(["--no-cpu-baseline", "--no-pmc"] + argv)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rec = bench.reference_start_leg(args, 0, dev, args.pairs, slot_only=True)
rec.pop("unconverged", None)
print(json.dumps(rec))
