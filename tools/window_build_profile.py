"""Developer tool: what building the three per-keyframe Gauss-Newton windows of the config-3 chain costs on the host (tracker, supplementary-mapping
window, scheduled-mapping window; ms per build + cProfile by own time).   python tools/window_build_profile.py"""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import MonoVO
from super_primitive_amd.odometery import loops
n = 30
seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
vo = MonoVO(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt))
for i in range(1, n):
    vo.step(i)
torch.cuda.synchronize()
print("image", tuple(frames[0].image.shape), "keyframes", len(vo.kfs), "N", [int(k.shape[0]) for k in vo.kf_klds])
aff = vo.kf_affs[-1]
def build_tracker():
    return loops.GnTracker(vo.kfs[-1], vo.kf_klds[-1], vo.kf_poses[-1], frames[n - 1], (0, 3), kf_aff=aff)
K = len(vo.kfs)
rows = [[(s.frame, s.pose, s.aff) for s in vo.supp_opt[k]] for k in range(K - 1)] + [[(frames[n - 2], vo.current_track, vo.current_aff), (frames[n - 1], vo.current_track, vo.current_aff)]]
def build_mapper():
    return loops.GnSuppMapper(vo.kfs, vo.kf_poses, vo.kf_klds, vo.kf_affs, rows, 10, window_size=5)
def build_map():
    return loops._build_map_window(vo.kfs, vo.kf_poses, vo.kf_klds, vo.kf_affs, rows, 500, 1e-4, K == 5, True, True, 1e-8, 'map', dict(loops.MAP_GN_SCHEDULE))
for name, fn in (("tracker", build_tracker), ("supp mapper", build_mapper), ("map window", build_map)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"== {name}: {1e3 * dt:.3f} ms per build")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[6:40]))
