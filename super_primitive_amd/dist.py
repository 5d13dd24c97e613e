"""Multi-GPU layer: frame pairs (or, for depth completion, images) are independent problems, so they are sharded
across ranks with NO data-path collective; the only exchange is the final gather of the optimised poses and
keypoint log-depths (a few KB per rank) -- RCCL over xGMI when the process group backend is "nccl", gloo on CPU
for the tests.  One process per GPU, launched by torch.distributed.run.

Depth completion (BASELINE.json configs[3]) offers both layouts of SURVEY.md section 8(e): images sharded across ranks
(``shard_list``: replicas, no collective -- what a throughput number should use) and the SEGMENTS of one image sharded
across ranks (``complete_depth_sharded``: per-rank median shifts and per-pixel {sum, count} accumulators, one
``all_reduce(SUM)`` of two integer maps -- 12 bytes per pixel -- then a local division)."""
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
    if not dist.is_initialized():
        return poses, klds                  # (a one-rank group still goes through the backend: tests/test_gpu_rccl.py)
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


# ---------------------------------------------------------------------------------------------------------------------
# VOID depth completion, segments of ONE image sharded across ranks
# ---------------------------------------------------------------------------------------------------------------------
def reduce_depth_accumulators(sums, counts, group=None):
    """all_reduce(SUM) of the per-pixel fixed-point depth sums (int64) and visible-segment counts (int32) of
    ``sp_depth_accumulate``, in place.  Integer addition is exact and commutative: every rank ends up with bitwise the
    accumulators a single GPU would have built over all segments; ``count == 0`` (invalid pixel) is the AND over ranks of
    the local invalidity, i.e. validity is OR-ed."""
    if dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return sums, counts


def shard_keyframe_segments(kf, rank=None, world=None):
    """The keyframe restricted to this rank's contiguous share of the segments (same image and intrinsics)."""
    from .image.keyframe import KeyFrame
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(kf.keypoint_regions.shape[0], rank, world)
    if hi == lo:
        return None, (lo, hi)
    return KeyFrame(kf.image, kf.K, kf.get_logdepth()[lo:hi].contiguous(), kf.keypoints[lo:hi].contiguous(),
                    kf.keypoint_regions[lo:hi].contiguous(), K_img=kf.K_img), (lo, hi)


def complete_depth_sharded(kf, sparse_depth, rank=None, world=None, group=None):
    """Segment-sharded form of ``depth_completion.segment_based_completion.infer_depth`` after the frontend
    (segment_based_completion.py:45-55): every rank re-initialises the log-depths of ITS segments from the sparse depth
    (per-segment medians are independent; invisible segments are dropped before the average, so their fill-in value --
    the one cross-segment statistic of ``segment_based_depth_reinit`` -- never reaches the output), accumulates them, and
    the accumulators are summed across ranks.  Returns (depth (H,W), invalid (H,W)) on every rank."""
    from .depth_completion import segment_based_completion as sbc
    from .odometery import depth_init
    average_visible_segments = sbc.average_visible_segments          # (looked up at call time: the world-8 gloo rehearsal swaps it)
    sub, (lo, hi) = shard_keyframe_segments(kf, rank, world)
    H, W = kf.geo_spatial_dim()
    red = lambda s, c: reduce_depth_accumulators(s, c, group)
    if sub is None:          # more ranks than segments: this rank contributes zero accumulators (and takes part in the same all_reduce)
        return average_visible_segments(None, None, None, reduce=red, empty=(H, W, kf.image.device))
    kld, visible = depth_init.segment_based_depth_reinit(sparse_depth.to(kf.image.device).clone().detach(), sub, mode='median',
                                                         return_info=True)
    return average_visible_segments(sub, kld, visible, reduce=red)
