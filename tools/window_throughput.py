"""Config 3's THROUGHPUT form: S independent mapping windows at the reference's extent (5 keyframes x 2 supporting
frames + 2 running ones: 14 free nodes = 112 camera unknowns, 200 log-depths, 28 photometric terms) optimised SIDE BY SIDE -- every
window on its own HIP stream, driven by its own host loop (``sp_window_gn_run`` releases the interpreter lock), the windows built
beforehand -- as PairBatch / PairStream is config 2's.  One window's Gauss-Newton iteration is four small dependent launches (cost pass
over 28 edges, per-edge reduce, per-keyframe Schur terms, one update workgroup): alone it leaves the chip empty; S of them fill it.
    python tools/window_throughput.py [S ...]      -> windows/s for every S, and the worst end state against the ground truth"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from window_bench import T, build, dev, rot_angle                      # noqa: E402
from super_primitive_amd.odometery.loops import MAP_GN_SCHEDULE, _build_map_window   # noqa: E402
from super_primitive_amd.optim.window import PoseWindowBatch                        # noqa: E402

S_list = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]
frames, kfi, si, kfs, poses, klds, affs, supp = build(5, 2, 2, 40)
gn = dict(MAP_GN_SCHEDULE)
REPS = 6


def make(span_points=None):
    win, supp_node, src_ids = _build_map_window(kfs, poses, klds, affs, supp, 25, 1e-4, True, True, True, 1e-8, 'map', gn, span_points=span_points)
    return win, win.nodes.clone(), win.kld.clone()


def optimise(win, nodes0, kld0):
    win.nodes.copy_(nodes0); win.kld.copy_(kld0)
    win.compose()
    win.reset_gn()
    n = win.run_gn(0, gn['max_iters'], irls_eps=gn['irls_eps'], conv_tol=gn['conv_tol'])
    n += win.run_gn(0, gn['polish_max'], irls_eps=gn['polish_eps'], conv_tol=gn['polish_tol'])
    return n


wins = [make() for _ in range(max(S_list))]
torch.cuda.synchronize()
for w in wins[:2]:
    optimise(*w)
torch.cuda.synchronize()
P = wins[0][0].node_poses().cpu().numpy().astype(np.float64)
err = max(rot_angle(P[k], frames[i].T_wc.astype(np.float64)) for k, i in enumerate(kfi))
print(f"reference-sized window (112 camera unknowns, 28 edges), Gauss-Newton {gn}: keyframe poses end {err:.1e} rad from the ground truth")
for S in S_list:
    streams = [torch.cuda.Stream() for _ in range(S)]
    its = [0] * S

    def work(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            for _ in range(REPS):
                its[i] += optimise(*wins[i])
            streams[i].synchronize()

    for rep in range(2):
        its = [0] * S
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"S = {S:3d} windows side by side (one stream + one host loop each): {S * REPS / dt:8.1f} windows/s, {1e3 * dt / REPS:7.2f} ms per round of S windows, "
          f"{sum(its) / (S * REPS):.1f} iterations per window, {1e6 * dt / max(sum(its), 1) * S:7.1f} us per iteration of one window", flush=True)

# ---- the device-side form: every launch covers all S windows (PoseWindowBatch / sp_window_gn_run_multi), ONE host loop ----
print("S windows per LAUNCH (PoseWindowBatch: window = blockIdx.z in the reduce / Schur / update kernels, one cost launch over the S work lists, one poll for all):")
latency_wins = wins
for S, span in [(S, None) for S in S_list] + [(S, 8192) for S in S_list if S >= 8]:
    if span is not None and (len(wins) < S or wins is latency_wins):
        wins = [make(span) for _ in range(max(S_list))]          # (the same windows with the cost pass's spans sized for throughput)
        print(f"... windows built with span_points = {span} ({wins[0][0].n_spans} workgroups per window's cost pass instead of {latency_wins[0][0].n_spans}):")
    batch = PoseWindowBatch([w[0] for w in wins[:S]])

    def run_all():
        for win, nodes0, kld0 in wins[:S]:
            win.nodes.copy_(nodes0); win.kld.copy_(kld0)
            win.compose()
        batch.reset_gn()
        n = batch.run_gn(0, gn['max_iters'], irls_eps=gn['irls_eps'], conv_tol=gn['conv_tol'])
        return n + batch.run_gn(0, gn['polish_max'], irls_eps=gn['polish_eps'], conv_tol=gn['polish_tol'])

    run_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rounds = 0
    for _ in range(REPS):
        rounds += run_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    P = wins[S - 1][0].node_poses().cpu().numpy().astype(np.float64)
    err = max(rot_angle(P[k], frames[i].T_wc.astype(np.float64)) for k, i in enumerate(kfi))
    print(f"S = {S:3d}: {S * REPS / dt:8.1f} windows/s, {1e3 * dt / REPS:7.2f} ms per batch of S windows, {rounds / REPS:.1f} rounds per batch, {1e6 * dt / max(rounds, 1):7.1f} us per round; "
          f"last window's keyframe poses {err:.1e} rad from the ground truth", flush=True)
