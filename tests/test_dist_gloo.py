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


@pytest.mark.parametrize("n_pairs,world", [(5, 2), (4, 2), (1, 2), (1030, 8), (5, 8)])
def test_shard_and_gather(n_pairs, world):
    """(1030 pairs on 8 ranks: config 5's 1024 + a ragged remainder; 5 pairs on 8 ranks: ranks without any pair take part in the same
    all_gather with empty shards.)"""
    N = 3
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


def _void_worker(rank, world, port, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from super_primitive_amd import dist as spd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        sums = torch.randint(0, 2 ** 40, (48 * 64,), generator=g, dtype=torch.int64)
        counts = torch.randint(0, 3, (48 * 64,), generator=g, dtype=torch.int32)
        mine = (sums.clone(), counts.clone())
        spd.reduce_depth_accumulators(sums, counts)
        out.put((rank, mine[0].numpy(), mine[1].numpy(), sums.numpy(), counts.numpy()))       # by value: the worker exits
    finally:
        dist.destroy_process_group()


def test_void_segment_sharding_collective_world2():
    """The one collective of segment-sharded depth completion: integer SUM of the per-pixel depth
    sums and counts.  Both ranks end up with bitwise the single-rank accumulators; validity is OR-ed (count > 0)."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_void_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    S = got[0][1] + got[1][1]
    C = got[0][2] + got[1][2]
    for _, _, _, s, c in got:
        assert np.array_equal(s, S) and np.array_equal(c, C)
    assert np.array_equal(C > 0, (got[0][2] > 0) | (got[1][2] > 0))


def test_segment_shards_partition_the_keyframe():
    from super_primitive_amd import dist as spd
    for n, world in ((1200, 8), (5, 8), (64, 3)):
        ranges = [spd.shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1


class _FakeBatch:
    """What bench.main() touches of a PairBatch, without a GPU: the N > 1 control flow is what is rehearsed."""

    def __init__(self, M):
        self.M, self.Ps, self.max_N, self.span_points = M, [1000] * M, 4, 256
        self.pose = torch.eye(4).reshape(1, 16).repeat(M, 1).contiguous()
        self.kld = torch.zeros(M * 4)
        self.calls = 0

    def cost_pass(self, level, mode, irls_eps=1e-3):
        self.calls += 1

    def solve_gn(self, level=0):
        self.pose += 1.0

    solve_adam = solve_gn

    def gn_step(self, level=0):
        self.cost_pass(level, 1)
        self.solve_gn(level)

    adam_step = gn_step

    def algorithmic_bytes(self, level):
        return 20 * sum(self.Ps)

    def restore_initial(self):
        self.restored = getattr(self, "restored", 0) + 1

    def run_scheduled(self, **kw):
        self.scheduled = getattr(self, "scheduled", 0) + 1
        return 1


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import contextlib
    import io
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    fake = _FakeBatch(6)
    bench.build_batch = lambda args, rank, dev: (fake, [])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "2", "--settle-ms", "0", "--dry-run"])
    out = buf.getvalue().strip()
    q.put((rank, json.loads(out) if out else None, fake.calls, fake.scheduled, fake.restored))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_control_flow_dry_run(world):
    """bench.py's N > 1 path (process group, barriers, MAX all_reduce of the timers, final all_gather of poses and log-depths,
    rank-0-only JSON line) executed once under gloo with the batch mocked -- so that path has run before a real 8-GPU node
    sees it; at world 8 exactly as the driver launches it.  No measurement is made."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    line0 = got[0][1]
    assert all(g[1] is None for g in got[1:]), "only rank 0 prints"
    # the driver's contract: every key of the bench line, plus the roofline object (cpu_baseline is an N = 1 leg)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line0, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line0["roofline"]) and line0["roofline"]["bound"] == "hbm"
    assert "workload" in line0["config"] and line0["unit"] == "iters/s" and line0["higher_is_better"] is True and line0["vs_baseline"] is None
    assert line0["n_gpus"] == world and line0["steps"] == 3 and line0["warmup"] == 2 and line0["scaling"] == "weak"
    assert line0["config"]["pairs_per_gpu"] == 6 and line0["data"].startswith("DRY RUN")
    np.testing.assert_allclose(line0["value"], world * 6 * 3 / (line0["ms_per_step"] * 3e-3), rtol=1e-6)     # whole-job aggregate
    assert all(g[2] == 2 + 3 for g in got)                                                            # warm-up + timed cost passes
    # the whole-job frame-pair leg: every rank ran the schedule twice (one untimed pass) from restored initial values
    assert all(g[3] == 2 and g[4] == 3 for g in got) and line0["frame_pairs_per_sec"] > 0
    # the self-proving part of an N > 1 line: the collective saw `world` ranks, and every rank reported its own record
    assert line0["rccl_world"] == world
    assert [r["rank"] for r in line0["ranks"]] == list(range(world))
    for r in line0["ranks"]:
        assert set(r) == {"rank", "device_index", "pci_bus_id", "kernel_ms", "elapsed_ms", "pairs"}
        assert r["pairs"] == 6 and r["kernel_ms"] >= 0 and r["elapsed_ms"] > 0
    assert max(r["elapsed_ms"] for r in line0["ranks"]) == pytest.approx(line0["ms_per_step"] * 3, rel=1e-6)


def _sharded_completion_worker(rank, world, port, n_segments, q):
    """complete_depth_sharded under gloo with the two device steps swapped for integer stand-ins (the HIP kernels need a GPU): what is
    rehearsed is the CONTROL FLOW -- every rank, also one that owns no segment, makes the same all_reduce calls in the same order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from super_primitive_amd import dist as spd
        from super_primitive_amd.depth_completion import segment_based_completion as sbc
        from super_primitive_amd.image.keyframe import KeyFrame
        from super_primitive_amd.odometery import depth_init
        H, W = 12, 16
        g = torch.Generator().manual_seed(7)
        masks = torch.rand(n_segments, H, W, generator=g) < 0.3
        kf = KeyFrame(torch.zeros(3, H, W), torch.eye(3), torch.zeros(n_segments, H, W), torch.zeros(n_segments, 2), masks)
        calls = []

        def reinit(sparse, sub, mode="median", return_info=False):
            n = sub.keypoint_regions.shape[0]
            return torch.zeros(n), torch.ones(n, dtype=torch.bool)

        def average(sub, kld, visible, reduce=None, empty=None):
            if sub is None:
                Hh, Ww, _ = empty
                sums, counts = torch.zeros(Hh * Ww, dtype=torch.int64), torch.zeros(Hh * Ww, dtype=torch.int32)
            else:
                m = sub.keypoint_regions
                lo = spd.shard_range(n_segments, rank, world)[0]
                ids = torch.arange(lo + 1, lo + 1 + m.shape[0], dtype=torch.int64)[:, None, None]
                sums, counts = (m.long() * ids).sum(0).reshape(-1), m.sum(0).reshape(-1).to(torch.int32)
            calls.append("reduce")
            reduce(sums, counts)
            return sums, counts

        depth_init.segment_based_depth_reinit = reinit
        sbc.average_visible_segments = average
        sums, counts = spd.complete_depth_sharded(kf, torch.zeros(H, W))
        ids = torch.arange(1, n_segments + 1, dtype=torch.int64)[:, None, None]
        want = ((masks.long() * ids).sum(0).reshape(-1), masks.sum(0).reshape(-1).to(torch.int32))
        q.put((rank, bool(torch.equal(sums, want[0]) and torch.equal(counts, want[1])), len(calls)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_segments", [1200, 5])
def test_segment_sharded_completion_control_flow_world8(n_segments):
    """BASELINE configs[3] (VOID, segments of one image sharded over 8 ranks) as a gloo rehearsal: 1200 segments (150 per rank) and 5
    segments (three ranks own none and still take part in the one all_reduce pair); every rank ends with the single-rank accumulators."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_completion_worker, args=(r, world, port, n_segments, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == list(range(world)) and all(g[1] for g in got) and all(g[2] == 1 for g in got)
