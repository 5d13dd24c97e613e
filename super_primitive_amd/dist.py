"""Multi-GPU layer: frame pairs (or, for depth completion, images) are independent problems, so they are sharded
across ranks with NO data-path collective; the only exchange is the final gather of the optimised poses and
keypoint log-depths (a few KB per rank) -- RCCL over xGMI when the process group backend is "nccl", gloo on CPU
for the tests.  One process per GPU, launched by torch.distributed.run."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of the items owned by ``rank`` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_list(items, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(len(items), rank, world)
    return items[lo:hi]


def gather_results(poses, klds, n_total=None):
    """All ranks receive every rank's results in global pair order.

    poses: (m_local,4,4) ; klds: (m_local, N) padded to a common N (rows may differ per rank by at most one, as
    produced by shard_range).  Returns (poses (M,4,4), klds (M,N)).  Ragged shard sizes are handled by padding to
    the largest shard and trimming -- a single all_gather per tensor, no point-to-point traffic."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return poses, klds
    world = dist.get_world_size()
    m_local = torch.tensor([poses.shape[0]], device=poses.device, dtype=torch.int64)
    sizes = [torch.zeros_like(m_local) for _ in range(world)]
    dist.all_gather(sizes, m_local)
    sizes = [int(s.item()) for s in sizes]
    m_max = max(sizes)

    def padded(t):
        if t.shape[0] == m_max:
            return t.contiguous()
        pad = torch.zeros((m_max - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat((t, pad)).contiguous()

    out = []
    for t in (poses, klds):
        bufs = [torch.empty((m_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for _ in range(world)]
        dist.all_gather(bufs, padded(t))
        out.append(torch.cat([b[:n] for b, n in zip(bufs, sizes)]))
    return out[0], out[1]
