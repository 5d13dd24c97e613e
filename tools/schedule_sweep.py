"""Developer sweep: frame pairs per second and worst end error of candidate coarse-to-fine schedules on bench.py's workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench


def worst_error(batch, pairs):
    P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
    worst = [0.0, 0.0, 0.0]
    for m in range(batch.M):
        gt = pairs[m % len(pairs)]
        ls = float(np.mean(gt.kld_gt - K[m]))
        R = P[m][:3, :3].T @ gt.pose_gt[:3, :3].astype(np.float64)
        rot = float(np.arctan2(0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]), 0.5 * (np.trace(R) - 1)))
        tt = float(np.abs(P[m][:3, 3] * np.exp(ls) - gt.pose_gt[:3, 3]).max())
        dd = float(np.abs(np.expm1(K[m] + ls - gt.kld_gt)).max())
        worst = [max(a, b) for a, b in zip(worst, (rot, tt, dd))]
    return worst


def main():
    args = bench.parse(["--no-cpu-baseline"] + sys.argv[1:])
    import super_primitive_amd.optim.pair_batch as pb
    pb.FRAME_PAIR_POINT_STRIDE = (1, 2, 4)
    orig = pb.PairBatch.__init__

    def init(self, *a, **kw):
        kw["extra_tables"] = [(0, 2), (1, 4), (2, 8), (0, 4), (2, 2)]
        orig(self, *a, **kw)
    pb.PairBatch.__init__ = init
    batch, pairs = bench.build_batch(args, 0, torch.device("cuda:0"))
    M = batch.M
    ph = lambda level, stride, n, tol, eps=1e-3: dict(level=level, stride=stride, max_iters=n, conv_tol=tol, irls_eps=eps)
    pol = ph(0, 1, 15, 1e-4, 1e-5)
    cands = {
        "base (1,2,4) tol 2e-3": [ph(2, 4, 25, 2e-3), ph(1, 2, 25, 2e-3), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 15": [ph(2, 4, 15, 2e-3), ph(1, 2, 25, 2e-3), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 10": [ph(2, 4, 10, 2e-3), ph(1, 2, 25, 2e-3), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 10, L1 cap 10": [ph(2, 4, 10, 2e-3), ph(1, 2, 10, 2e-3), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 12 tol 5e-3, L1 cap 12 tol 5e-3": [ph(2, 4, 12, 5e-3), ph(1, 2, 12, 5e-3), ph(0, 1, 25, 2e-3), pol],
        "coarse tol 5e-3": [ph(2, 4, 25, 5e-3), ph(1, 2, 25, 5e-3), ph(0, 1, 25, 2e-3), pol],
        "coarse tol 1e-2": [ph(2, 4, 25, 1e-2), ph(1, 2, 25, 1e-2), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 10 + L0/s2": [ph(2, 4, 10, 2e-3), ph(1, 2, 10, 2e-3), ph(0, 2, 10, 2e-3), ph(0, 1, 25, 2e-3), pol],
        "L2 cap 8, L1 cap 8, L0 tol 5e-3": [ph(2, 4, 8, 2e-3), ph(1, 2, 8, 2e-3), ph(0, 1, 25, 5e-3), pol],
        "caps 10, L0/s4 + L0/s2 + polish": [ph(2, 4, 10, 2e-3), ph(1, 4, 10, 2e-3), ph(0, 4, 10, 1e-3), ph(0, 2, 10, 5e-4), pol],
    }
    for name, phases in cands.items():
        for ce in (4,):
            best = 0.0
            for rep in range(3):
                batch.restore_initial()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = batch.run_scheduled(check_every=ce, phases=phases)
                torch.cuda.synchronize()
                best = max(best, M / (time.perf_counter() - t0))
            n_it = (batch.lm_state[:, 2] + batch.lm_state[:, 3])
            w = worst_error(batch, pairs)
            print(f"{name:32s} check {ce}: {best:8.0f} pairs/s  launched {n:3d}  iters/pair mean {float(n_it.mean()):5.1f} max {int(n_it.max()):3d}  "
                  f"worst rot {w[0]:.1e} t {w[1]:.1e} depth {w[2]:.1e}", flush=True)


def trace():
    """Per-round phase occupancy of the quoted schedule: how many pairs sit in each phase after every iteration."""
    import ctypes
    from super_primitive_amd import _lib
    args = bench.parse(["--no-cpu-baseline"])
    import super_primitive_amd.optim.pair_batch as pb
    batch, pairs = bench.build_batch(args, 0, torch.device("cuda:0"))
    kw = {k: v for k, v in pb.FRAME_PAIR_SCHEDULE.items() if k != "check_every"}
    sched = batch.schedule(**kw)
    batch.restore_initial()
    batch.phase.zero_(); batch.phase_iters.zero_()
    rounds = []
    for it in range(120):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(batch.lib.sp_pairs_schedule_cost(ctypes.addressof(sched), _lib.ptr(batch.phase), _lib.stream_ptr()), "c")
        _lib.check(batch.lib.sp_pairs_schedule_gn_step(ctypes.addressof(sched), batch.M, batch.max_N, 8.0, 0.5, 1e-7, _lib.ptr(batch.lm_state),
                                                       _lib.ptr(batch.backup), _lib.ptr(batch._costs), _lib.ptr(batch.phase),
                                                       _lib.ptr(batch.phase_iters), _lib.stream_ptr()), "s")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        occ = torch.bincount(batch.phase, minlength=sched.n_phases + 1).cpu().numpy()
        rounds.append((dt, occ))
        print(f"round {it:3d}: {dt * 1e6:7.1f} us  pairs per phase after it {occ.tolist()}  rejected so far {int(batch.lm_state[:, 3].sum())}", flush=True)
        if occ[-1] == batch.M:
            break
    print(f"total {sum(r[0] for r in rounds) * 1e3:.2f} ms")




def scenes(n=48):
    """The quoted schedule on n DISTINCT synthetic scenes (bench.py renders 4 and copies them): worst error of each against its
    ground truth, with the decimated coarse levels and on all points."""
    import super_primitive_amd.optim.pair_batch as pb
    from super_primitive_amd import synth
    prs = [synth.make_pair(480, 640, 64, seed=2000 + s, overlap=4, init_sigma=0.004) for s in range(n)]
    batch = pb.PairBatch.from_synth(prs, levels=(0, 3), device="cuda:0", point_stride=pb.FRAME_PAIR_POINT_STRIDE)
    kw = {k: v for k, v in pb.FRAME_PAIR_SCHEDULE.items() if k != "check_every"}
    for label, use_coarse in (("decimated coarse levels", True), ("all points at every level", False)):
        batch.restore_initial()
        n_it = batch.run_scheduled(use_coarse=use_coarse, **kw)
        torch.cuda.synchronize()
        P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
        errs = []
        for m, gt in enumerate(prs):
            ls = float(np.mean(gt.kld_gt - K[m]))
            R = P[m][:3, :3].T @ gt.pose_gt[:3, :3].astype(np.float64)
            rot = float(np.arctan2(0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]), 0.5 * (np.trace(R) - 1)))
            errs.append((rot, float(np.abs(P[m][:3, 3] * np.exp(ls) - gt.pose_gt[:3, 3]).max()), float(np.abs(np.expm1(K[m] + ls - gt.kld_gt)).max())))
        e = np.array(errs)
        bad = [m for m in range(n) if e[m, 0] > 1e-4 or e[m, 1] > 1.5e-4 or e[m, 2] > 1.2e-3]
        its = (batch.lm_state[:, 2] + batch.lm_state[:, 3]).cpu().numpy()
        print(f"{n} distinct scenes, {label}: launched {n_it}, iterations per pair {its.mean():.1f} (max {int(its.max())}); worst rot {e[:, 0].max():.1e} "
              f"t {e[:, 1].max():.1e} depth {e[:, 2].max():.1e}; median depth {np.median(e[:, 2]):.1e}; outside the bar: {bad}", flush=True)
        for m in bad:
            print(f"   scene seed {2000 + m}: rot {e[m, 0]:.1e} t {e[m, 1]:.1e} depth {e[m, 2]:.1e}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        sys.argv.pop(1)
        trace()
    elif len(sys.argv) > 1 and sys.argv[1] == "scenes":
        scenes(int(sys.argv[2]) if len(sys.argv) > 2 else 48)
    else:
        main()
