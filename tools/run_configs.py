#!/usr/bin/env python
"""Synthetic stand-ins for the five BASELINE.json configurations (no dataset ships with the build), run through
the HIP path on one GPU.  Prints one line per configuration; numbers are quoted in DESIGN.md.

  1  320x240, 8 segments           -- the reference's CPU-runnable case: oracle (CPU) next to the HIP API loop
  2  640x480, 64 segments, 3 levels -- bench.py (headline)
  3  TUM-shaped MonoVO inner loops  -- 224x288 keyframe, tracking [0,0,300] Adam steps / frame, windowed mapping
  4  VOID-shaped depth completion   -- 480x640, ~1200 sparse-point segments: per-segment median + per-pixel average
  5  many pairs x 128 segments      -- bench.py --segments 128 (per-GPU slice of the 1024-pair / 8-GPU case)
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.core import dense_optim
from super_primitive_amd.image.keyframe import KeyFrame, keyframe_pyramid
from super_primitive_amd.odometery.two_frame_sfm import SfM
from super_primitive_amd.odometery.loops import GnTracker, track_frame, track_frame_fused, track_frame_gn, map_window
from super_primitive_amd.odometery.depth_init import segment_based_depth_reinit
from super_primitive_amd.depth_completion.segment_based_completion import average_visible_segments
from super_primitive_amd.lie.lie_algebra import invertSE3
from super_primitive_amd.optim.pair_batch import PairBatch

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def frames(p):
    return (KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)),
            KeyFrame(t(p.trg_image), t(p.K)))
def sync_time(f, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def rot_err(A, B):
    D = A[:3, :3].astype(np.float64) @ B[:3, :3].astype(np.float64).T
    w = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])      # exact for small angles
    return float(np.arctan2(np.linalg.norm(w), (np.trace(D) - 1) / 2))

# ---- config 1 -------------------------------------------------------------------------------------------
p = synth.make_pair(240, 320, 8, seed=1, init_sigma=0.01)
src, trg = frames(p)
cfg = {"aligment": {"pyramid_min": 0, "pyramid_max": 3, "cost_params": {}}}
for fused, graphed in ((True, False), (False, True), (False, False)):
    sfm = SfM(cfg, src, [trg], [t(p.pose_init)], num_iters=5); sfm.init_optimisation(kld_init=t(p.kld_init)); sfm.run(fused=fused, graphed=graphed)
    sfm = SfM(cfg, src, [trg], [t(p.pose_init)], num_iters=500); sfm.init_optimisation(kld_init=t(p.kld_init))
    dt = sync_time(lambda: sfm.run(fused=fused, graphed=graphed))
    what = 'fused optimiser (3 launches / iteration)' if fused else ('eager statements (photomeric_cost + autograd + torch Adam) replayed from a hipGraph per level' if graphed else 'eager (autograd + torch Adam)')
    print(f"config 1  320x240x8: drop-in SfM driver, {what}: "
          f"{1500/dt:.0f} Adam it/s (3 levels x 500, the reference budget) | loss {float(sfm.losses[0]):.4f} -> {float(sfm.losses[-1]):.4f}")

# ---- config 3: TUM-shaped tracking + mapping ------------------------------------------------------------
H, W, N = 224, 288, 40
p = synth.make_pair(H, W, N, seed=3, init_sigma=0.01, overlap=3)
src, trg = frames(p)
levels = (0, 3)
src_pyr, trg_pyr = keyframe_pyramid(src, *levels), keyframe_pyramid(trg, *levels)
with torch.no_grad():
    pre = [dense_optim.unproject_kf(s, t(p.kld_gt)) for s in src_pyr]
supp_T0 = invertSE3(t(p.pose_init))
args = (pre, trg_pyr, supp_T0, torch.eye(4, device=dev), [0, 0, 300])
track_frame(pre, trg_pyr, supp_T0, torch.eye(4, device=dev), [0, 0, 5])
torch.cuda.synchronize(); t0 = time.perf_counter()
supp_T, _, losses = track_frame(*args, lr=5e-3, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
est = invertSE3(supp_T).cpu().numpy()
print(f"config 3  tracking 224x288x{N}, 300 Adam steps/frame, eager (precomputed dict -> 24 B/point list + autograd + torch Adam): {dt*1e3:.0f} ms/frame "
      f"({300/dt:.0f} it/s), loss {float(losses[0]):.4f} -> {float(losses[-1]):.4f}, rot err {rot_err(est, p.pose_gt):.2e} rad, t err {np.abs(est[:3,3]-p.pose_gt[:3,3]).max():.2e}")
fargs = (src, t(p.kld_gt), trg, supp_T0, torch.eye(4, device=dev), [0, 0, 300], levels)
track_frame_fused(*fargs, lr=5e-3, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
torch.cuda.synchronize(); t0 = time.perf_counter()
supp_T, _, losses = track_frame_fused(*fargs, lr=5e-3, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
est = invertSE3(supp_T).cpu().numpy()
print(f"config 3  tracking, fused optimiser incl. table / pyramid set-up per frame: {dt*1e3:.1f} ms/frame ({300/dt:.0f} it/s), "
      f"rot err {rot_err(est, p.pose_gt):.2e} rad, t err {np.abs(est[:3,3]-p.pose_gt[:3,3]).max():.2e}")
# Gauss-Newton / LM tracking (sp_window_gn_step): 6 pose + 2 affine unknowns, <= 15 iterations
gargs = (src, t(p.kld_gt), trg, supp_T0, torch.eye(4, device=dev), levels)
track_frame_gn(*gargs, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    supp_T, _, losses, its = track_frame_gn(*gargs, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
est = invertSE3(supp_T).cpu().numpy()
print(f"config 3  tracking, Gauss-Newton window optimiser incl. table / pyramid set-up per frame: {dt*1e3:.1f} ms/frame ({1/dt:.0f} frames/s), {its} LM iterations, "
      f"loss {float(losses[0]):.4f} -> {float(losses[-1]):.4f}, rot err {rot_err(est, p.pose_gt):.2e} rad, t err {np.abs(est[:3,3]-p.pose_gt[:3,3]).max():.2e}")
trk = GnTracker(src, t(p.kld_gt), torch.eye(4, device=dev), trg, levels, kf_aff=torch.zeros(2, device=dev))
trk.track(trg, supp_T0, torch.zeros(2, device=dev))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    supp_T, _, losses, its = trk.track(trg, supp_T0, torch.zeros(2, device=dev))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
est = invertSE3(supp_T).cpu().numpy()
print(f"config 3  tracking, Gauss-Newton with ONE window per keyframe (GnTracker: per frame only the target pyramid, the poses and the LM state): {dt*1e3:.2f} ms/frame "
      f"({1/dt:.0f} frames/s), {its} LM iterations, rot err {rot_err(est, p.pose_gt):.2e} rad, t err {np.abs(est[:3,3]-p.pose_gt[:3,3]).max():.2e}")
# the same frame-to-keyframe problem for a whole batch of frames on device (pose + affine, depths fixed)
B = 64
pb = PairBatch([src] * 1, [t(p.trg_image)], [t(p.K)], t(p.pose_init)[None].repeat(B, 1, 1), [t(p.kld_gt)], levels=levels, use_affine=True,
               replicate=B, tile_points=4096)
for _ in range(3): pb.adam_step(0, lr_kld=0.0, lr_pose=5e-3, lr_aff=5e-3)
dt = sync_time(lambda: pb.adam_step(0, lr_kld=0.0, lr_pose=5e-3, lr_aff=5e-3), 300)
print(f"config 3  tracking, {B} frames side by side on device: {B/ (300*dt):.0f} frames/s at 300 steps/frame ({B/dt:.0f} Adam it/s)")
frames_w, est, klds, affs = synth.window_inputs(300, 3, H=H, W=W, N=N)
kfs = [KeyFrame(t(f.image), t(f.K), t(f.logdepth_perseg), t(f.keypoints), t(f.keypoint_regions)) for f in frames_w[0::2]]
supp = [[(KeyFrame(t(frames_w[2 * k + 1].image), t(frames_w[2 * k + 1].K)), t(est[2 * k + 1]), t(affs[2 * k + 1]))] for k in range(3)]
margs = (kfs, [t(est[2 * k]) for k in range(3)], [t(k) for k in klds], [t(affs[2 * k]) for k in range(3)], supp)
for fused in (True, False):
    map_window(*margs, 5, window_size=3, initialised=False, fused=fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = map_window(*margs, 500, window_size=3, initialised=False, fused=fused)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"config 3  windowed mapping, 3 keyframes + 3 supporting frames (10 edges), 500 Adam steps, {'fused' if fused else 'eager'}: "
          f"{dt*1e3:.0f} ms ({500/dt:.0f} it/s), loss {float(out['losses'][0]):.4f} -> {float(out['losses'][-1]):.4f}")

map_window(*margs, 40, window_size=3, initialised=True, optimiser="gn")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    out = map_window(*margs, 40, window_size=3, initialised=True, optimiser="gn")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"config 3  windowed mapping, same window, Gauss-Newton (5 free poses, 80 log-depths, 5 affine pairs; Schur + Cholesky on the device): {dt*1e3:.1f} ms/window "
      f"({1/dt:.0f} windows/s), {out['stopped']} LM iterations ({out['gn']['accepted']} accepted), loss {float(out['losses'][0]):.4f} -> {float(out['losses'][-1]):.6f}")
# the whole chain on a 30-frame sequence (odometery/sequence.py): track -> keyframe criterion -> depth render -> re-initialisation -> mapping
from super_primitive_amd.odometery.sequence import run_sequence
rng = np.random.default_rng(31)
base = 0.6 * np.array([0.05, -0.02, 0.015, 0.01, -0.015, 0.008])
seq = synth.make_sequence(H, W, N, [k * base + 0.003 * rng.standard_normal(6) * (k > 0) for k in range(30)], keyframe_ids=list(range(30)), seed=31, overlap=1)
sframes = [KeyFrame(t(f.image), t(f.K)) for f in seq]
to_kf = lambda i: KeyFrame(t(seq[i].image), t(seq[i].K), t(seq[i].logdepth_perseg), t(seq[i].keypoints), t(seq[i].keypoint_regions))
for engine in ("gn", "adam"):
    run_sequence(sframes[:4], to_kf, t(seq[0].T_wc), t(seq[0].kld_gt), engine=engine)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = run_sequence(sframes, to_kf, t(seq[0].T_wc), t(seq[0].kld_gt), engine=engine, translation_thresh=0.1, window_size=3, map_steps=300)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    P = out["track_poses"].cpu().numpy().astype(np.float64); G = np.stack([f.T_wc for f in seq]).astype(np.float64)
    sec = out["seconds"]
    print(f"config 3  30-frame sequence, {engine}: {29/dt:.0f} frames/s end to end (tracking {29/sec['track']:.0f} frames/s, {out['n_mappings']} mappings at "
          f"{1e3*sec['mapping']/max(out['n_mappings'],1):.1f} ms, keyframes {out['all_kf_ids']}); trajectory error max rot {max(rot_err(a, b) for a, b in zip(P, G)):.1e} rad, "
          f"t {np.abs(P[:, :3, 3] - G[:, :3, 3]).max():.1e}")

# ---- config 4: VOID-shaped depth completion --------------------------------------------------------------
p = synth.make_pair(480, 640, 1200, seed=4, shape="blobs")
src, _ = frames(p)
rng = np.random.default_rng(0)
sparse = np.zeros_like(p.depth); rc = p.meta["kp_rc"]; sparse[rc[:, 0], rc[:, 1]] = p.depth[rc[:, 0], rc[:, 1]]
sp = t(sparse)
def complete():
    kld, vis = segment_based_depth_reinit(sp.clone(), src, mode='median', return_info=True)
    return average_visible_segments(src, kld, vis)
depth, invalid = complete()
dt = sync_time(complete, 20)
ok = ~invalid.cpu().numpy()
err = np.abs(depth.cpu().numpy()[ok] - p.depth[ok]) / p.depth[ok]
from super_primitive_amd.segment_table import table_of
# algorithmic bytes: the re-initialisation reads every table point's pixel word and the sparse depth under it (4 + 4 B),
# the average reads pixel word and log-depth again (4 + 4 B) and read-modify-writes the 12-byte accumulator of every image pixel once
P4 = table_of(src).P
b4 = 16.0 * P4 + 24.0 * 480 * 640
print(f"config 4  VOID-shaped 480x640, {p.N} segments (P = {P4} points = {P4 / (480 * 640):.0f} per pixel): {dt*1e3:.2f} ms/image ({1/dt:.0f} images/s), "
      f"coverage {ok.mean():.2f}, median rel err {np.median(err):.2e}; algorithmic bytes {b4 / 1e6:.0f} MB per image -> {b4 / dt / 1e12:.2f} TB/s = {b4 / dt / 8e12:.3f} of HBM "
      f"(the table fits the 256 MB Infinity Cache only in part; what bounds the pass is the {P4 / (480 * 640):.0f} fixed-point atomics PER PIXEL of the per-pixel average, "
      f"not its bytes)")
print("config 2 / 5: see bench.py (--segments 128 for config 5's per-GPU slice)")
