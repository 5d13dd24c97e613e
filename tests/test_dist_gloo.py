"""CPU, world_size = 2, gloo: the sharding + final-gather layer used by bench.py and the batched drivers.
(The data path has no collective: pairs are independent; the N > 1 GPU run differs only by backend 'nccl' = RCCL.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, N, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from super_primitive_amd.dist import gather_results, shard_list, shard_range
    lo, hi = shard_range(n_pairs, rank, world)
    ids = shard_list(list(range(n_pairs)))
    assert ids == list(range(lo, hi))
    # stand-in for "optimise my shard": results are a deterministic function of the global pair id
    poses = torch.stack([torch.eye(4) * (i + 1) for i in ids]) if ids else torch.zeros(0, 4, 4)
    klds = torch.stack([torch.full((N,), float(i)) for i in ids]) if ids else torch.zeros(0, N)
    P, Kd = gather_results(poses, klds)
    q.put((rank, P.numpy(), Kd.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 4, 1])
def test_shard_and_gather_world2(n_pairs):
    world, N = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, P, Kd in got:
        assert P.shape == (n_pairs, 4, 4) and Kd.shape == (n_pairs, N)
        for i in range(n_pairs):
            np.testing.assert_array_equal(P[i], np.eye(4) * (i + 1))
            np.testing.assert_array_equal(Kd[i], np.full(N, float(i)))


def test_shard_range_is_a_balanced_partition():
    from super_primitive_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
