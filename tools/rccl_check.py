"""RCCL on hardware with whatever ranks are there: run under ``python -m torch.distributed.run --nproc-per-node N``.
Initialises the ``nccl`` (= RCCL) process group on cuda:LOCAL_RANK and pushes the package's two collectives through it --
``dist.gather_results`` (the final all_gather of poses and log-depths) and ``dist.complete_depth_sharded`` (the integer all_reduce of the
segment-sharded depth completion) -- comparing the latter with the single-process result (bitwise: integer accumulators).  Rank 0 prints
one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    from super_primitive_amd import dist as spd, synth
    from super_primitive_amd.depth_completion.segment_based_completion import average_visible_segments
    from super_primitive_amd.image.keyframe import KeyFrame
    from super_primitive_amd.odometery import depth_init
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # (1) final gather of ragged shards
    M, N = 5 * world + 1, 7
    lo, hi = spd.shard_range(M, rank, world)
    poses = torch.arange(M * 16, dtype=torch.float32, device=dev).reshape(M, 4, 4)
    klds = torch.arange(M * N, dtype=torch.float32, device=dev).reshape(M, N)
    P, K = spd.gather_results(poses[lo:hi].contiguous(), klds[lo:hi].contiguous())
    gather_ok = bool(torch.equal(P, poses) and torch.equal(K, klds))
    # (2) segment-sharded depth completion: one integer all_reduce
    pair = synth.make_pair(120, 160, 24, seed=7, shape="blobs", blob_coverage=1.3)
    kf = KeyFrame(t(pair.src_image), t(pair.K), t(pair.logdepth_perseg), t(pair.keypoints), t(pair.keypoint_regions))
    rng = np.random.default_rng(3)
    sparse = np.where(rng.uniform(size=(120, 160)) < 0.02, np.exp(pair.logdepth_perseg.max(0)) * 1.7, 0.0).astype(np.float32)
    depth, invalid = spd.complete_depth_sharded(kf, t(sparse))
    kld, vis = depth_init.segment_based_depth_reinit(t(sparse).clone(), kf, mode="median", return_info=True)
    d1, i1 = average_visible_segments(kf, kld, vis)
    void_ok = bool(torch.equal(depth, d1) and torch.equal(invalid, i1))
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    pr = torch.cuda.get_device_properties(dev)
    if rank == 0:
        print(json.dumps({"backend": dist.get_backend(), "world": world, "all_reduce_of_ones": float(ones.item()), "gather_results_ok": gather_ok,
                          "complete_depth_sharded_equals_single_process": void_ok, "device": pr.name,
                          "pci_bus_id": f"{int(getattr(pr, 'pci_domain_id', 0)):04x}:{int(getattr(pr, 'pci_bus_id', -1)):02x}:{int(getattr(pr, 'pci_device_id', 0)):02x}",
                          "gathered_shapes": [list(P.shape), list(K.shape)], "nccl_version": list(torch.cuda.nccl.version())}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
