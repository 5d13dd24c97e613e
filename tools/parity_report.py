#!/usr/bin/env python
"""Measured parity figures on an MI355X -> stdout (committed as profiles/rNN_parity.txt).

Everything is compared against fixtures recorded with the REAL reference (tests/golden, oracle/gen_goldens*.py); the
tests assert about 5x the figures printed here.  North-star bar: pose 1e-4 rad / 1e-4 t, depth 1e-3 relative.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from gpu_util import T, frames_from_golden, frames_from_synth, npy  # noqa: E402
from parity_util import fullsize_pair, pose_depth_errors, rel_max, rot_angle  # noqa: E402

CFG0 = {"mode": "colour", "collect_stats": 0}


def gradients_at_size():
    from super_primitive_amd.core import dense_optim
    from super_primitive_amd.image import keyframe
    print("== single cost evaluation at full size: residual and gradients (max |err| / max |ref|) ==")
    print("   columns: HIP vs reference fp32 | HIP vs fp64 | reference fp32 vs fp64   (fp64 = the oracle restatement in float64)")
    for name in ("g14_config1_converged", "g15_config2_fullsize", "g16_config5_seg128"):
        g = load_golden(name)
        pair = fullsize_pair(g)
        src, trg = frames_from_synth(pair)
        sp, tp = keyframe.keyframe_pyramid(src, 0, 3), keyframe.keyframe_pyramid(trg, 0, 3)
        for li in range(3):
            if f"L{li}_residual" not in g.files:
                continue
            kld0 = g[f"L{li}_in_kld"] if f"L{li}_in_kld" in g.files else pair.kld_init
            pose0 = g[f"L{li}_in_pose"] if f"L{li}_in_pose" in g.files else pair.pose_init
            kld, pose = T(kld0, True), T(pose0, True)
            out = dense_optim.photomeric_cost(sp[li], tp[li], kld, pose, CFG0)
            out["residual"].abs().mean().backward()
            r = float(out["residual"])
            r32, r64 = float(g[f"L{li}_residual"]), float(g[f"L{li}_residual64"])
            print(f"  {name} level {li} ({int(pair.keypoint_regions.sum())} pts): residual rel {abs(r - r32) / r32:.1e} | {abs(r - r64) / r64:.1e} | {abs(r32 - r64) / r64:.1e}")
            for key, got in (("g_kld", npy(kld.grad)), ("g_pose", npy(pose.grad))):
                a, b, c = rel_max(got, g[f"L{li}_{key}"]), rel_max(got, g[f"L{li}_{key}64"]), rel_max(g[f"L{li}_{key}"], g[f"L{li}_{key}64"])
                print(f"      {key:7s} {a:.1e} | {b:.1e} | {c:.1e}")


def small_goldens():
    from super_primitive_amd.core import dense_optim
    print("== gradients on the small reference goldens (max |err| / max |ref|) ==")
    worst = {}
    for name in ["g1_grid_48x64", "g1_blobs_affine_60x80", "g1_pyramid_72x96", "g1_behind_camera_48x64", "g1_odd_45x67"]:
        g = load_golden(name)
        for li in range(int(g["n_levels"])):
            p = f"L{li}_"
            src, trg = frames_from_golden(g, g[p + "lvl_src_image"], g[p + "lvl_trg_image"], g[p + "lvl_K_img"])
            kld, pose = T(g["in_kld"], True), T(g["in_pose"], True)
            aff = (T(g["in_aff_src"], True), T(g["in_aff_trg"], True)) if "in_aff_src" in g else None
            out = dense_optim.photomeric_cost(src, trg, kld, pose, CFG0, affine_comp=aff)
            out["residual"].abs().mean().backward()
            items = [("residual", abs(float(out["residual"]) - float(g[p + "residual"])) / float(g[p + "residual"])),
                     ("g_kld", rel_max(npy(kld.grad), g[p + "g_kld"])), ("g_pose", rel_max(npy(pose.grad), g[p + "g_pose"]))]
            if aff is not None:
                items += [("g_aff_src", rel_max(npy(aff[0].grad), g[p + "g_aff_src"])), ("g_aff_trg", rel_max(npy(aff[1].grad), g[p + "g_aff_trg"]))]
            for k, v in items:
                worst[k] = max(worst.get(k, 0.0), v)
    print("  worst over 5 goldens x levels:", {k: f"{v:.1e}" for k, v in worst.items()})


def trajectories():
    from super_primitive_amd.lie.lie_algebra import invertSE3
    from super_primitive_amd.odometery.loops import map_window, track_frame_fused
    from super_primitive_amd.odometery.two_frame_sfm import SfM
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_window import window_case, window_errors
    print("== K-step Adam trajectories against the reference's (goldens G9) ==")
    g = load_golden("g9a_traj_sfm")
    for fused in (True, False):
        src, trg = frames_from_golden(g)
        sfm = SfM({"aligment": {"pyramid_min": 0, "pyramid_max": 2, "cost_params": {}}}, src, [trg], [T(g["in_pose_init"])], num_iters=int(g["steps"]))
        sfm.init_optimisation(kld_init=T(g["in_kld"]))
        sfm.run(fused=fused)
        L = np.array([float(l) for l in sfm.losses])
        P = npy(sfm.poses()[0])
        print(f"  two-frame SfM 80 steps, {'fused' if fused else 'eager'}: first-loss rel {abs(L[0] - g['losses'][0]) / g['losses'][0]:.1e}, max loss rel "
              f"{np.abs(L / g['losses'] - 1).max():.1e}, rot {rot_angle(P, g['final_pose']):.1e} rad, t {np.abs(P[:3, 3] - g['final_pose'][:3, 3]).max():.1e}, "
              f"kld {np.abs(npy(sfm.keypoint_logdepths()) - g['final_kld']).max():.1e}")
    g = load_golden("g9b_traj_track")
    src, trg = frames_from_golden(g)
    dev = src.image.device
    supp_T, aff, losses = track_frame_fused(src, T(g["in_kld"]), trg, invertSE3(T(g["in_pose_init"])), torch.eye(4, device=dev), [int(g["steps"])], (0, 1),
                                            lr=5e-3, prev_aff=torch.zeros(2, device=dev), curr_aff=torch.zeros(2, device=dev))
    L = np.array([float(l) for l in losses])
    print(f"  tracking 40 steps, fused: max loss rel {np.abs(L / g['losses'] - 1).max():.1e}, rot {rot_angle(npy(supp_T), g['final_supp_T']):.1e} rad, "
          f"t {np.abs(npy(supp_T)[:3, 3] - g['final_supp_T'][:3, 3]).max():.1e}, affine {np.abs(npy(aff) - g['final_aff']).max():.1e}")
    g = load_golden("g9d_traj_window")
    for tag in ("full", "init"):
        for fused in (True, False):
            args, kw = window_case(g, tag)
            out = map_window(*args, fused=fused, **kw)
            L = np.array([float(l) for l in out["losses"]])
            err = window_errors(out, g, tag)
            print(f"  windowed mapping '{tag}' 30 steps, {'fused' if fused else 'eager'}: max loss rel {np.abs(L / g[tag + '_losses'] - 1).max():.1e}, "
                  + ", ".join(f"{k} {v:.1e}" for k, v in err.items()))


def config1_converged():
    from super_primitive_amd.odometery.two_frame_sfm import SfM
    g = load_golden("g14_config1_converged")
    pair = fullsize_pair(g)
    print("== BASELINE config 1 (320x240, 8 segments, 3 levels): reference schedule 3 x 500 Adam + polish 300 @ lr/10 + 300 @ lr/100 ==")
    for fused in (True, False):
        src, trg = frames_from_synth(pair)
        sfm = SfM({"aligment": {"pyramid_min": 0, "pyramid_max": 3, "cost_params": {}}}, src, [trg], [T(pair.pose_init)], num_iters=int(g["iters"]))
        sfm.init_optimisation(kld_init=T(pair.kld_init))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sfm.run(fused=fused)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tag = "fused" if fused else "eager"
        L = np.array([float(l) for l in sfm.losses])
        print(f"  {tag}: 1500 iterations in {t1 - t0:.2f} s ({1500 / (t1 - t0):.0f} it/s); first-loss rel {abs(L[0] - g['losses'][0]) / g['losses'][0]:.1e}; "
              f"loss at 10/100/499: {L[10]:.6f}/{L[100]:.6f}/{L[499]:.6f} (reference {g['losses'][10]:.6f}/{g['losses'][100]:.6f}/{g['losses'][499]:.6f})")
        e = pose_depth_errors(npy(sfm.poses()[0]), npy(sfm.keypoint_logdepths()), g["L2_end_pose"], g["L2_end_kld"])
        print(f"      after the 3 levels (unconverged, lr 1e-2 jitter) vs reference: rot {e[0]:.1e} t {e[1]:.1e} depth {e[2]:.1e} (gauge-aligned)")
        for scale in (0.1, 0.01):
            sfm.run(fused=fused, lr_scale=scale, levels=[2], num_iters=int(g["polish"]))
        P, k = npy(sfm.poses()[0]), npy(sfm.keypoint_logdepths())
        raw = pose_depth_errors(P, k, g["final_pose"], g["final_kld"], gauge=False)
        al = pose_depth_errors(P, k, g["final_pose"], g["final_kld"])
        print(f"      converged + polished vs reference final: raw rot {raw[0]:.1e} t {raw[1]:.1e} depth {raw[2]:.1e} | scale-gauge aligned rot {al[0]:.1e} t {al[1]:.1e} depth {al[2]:.1e}; "
              f"final loss {float(sfm.losses[-1]):.8f} (reference {float(g['final_loss']):.8f})")


def config2_schedule():
    from super_primitive_amd.optim.pair_batch import FRAME_PAIR_SCHEDULE, PairBatch
    from super_primitive_amd.odometery.two_frame_sfm import SfM
    g = load_golden("g15_config2_fullsize")
    pair = fullsize_pair(g)
    print("== BASELINE config 2 (640x480, 64 segments, 3 levels) against the minimiser of the reference cost (golden g15) ==")
    print(f"   reference minimiser vs synthetic ground truth: rot {rot_angle(g['min_pose'], g['pose_gt']):.1e} t {np.abs(g['min_pose'][:3, 3] - g['pose_gt'][:3, 3]).max():.1e} kld {np.abs(g['min_kld'] - g['kld_gt']).max():.1e}")
    for ipl in (10, 20):
        for polish in (0, 10, 40):
            b = PairBatch.from_synth([pair], levels=(0, 3), device="cuda:0", tile_points=2048)
            b.run(ipl, mode="gn", polish_iters=polish, polish_eps=FRAME_PAIR_SCHEDULE["polish_eps"])
            torch.cuda.synchronize()
            raw = pose_depth_errors(npy(b.poses()[0]), npy(b.klds()[0]), g["min_pose"], g["min_kld"], gauge=False)
            al = pose_depth_errors(npy(b.poses()[0]), npy(b.klds()[0]), g["min_pose"], g["min_kld"])
            print(f"  GN 3 x {ipl} + polish {polish:2d} @ eps {FRAME_PAIR_SCHEDULE['polish_eps']:g}: raw rot {raw[0]:.1e} t {raw[1]:.1e} depth {raw[2]:.1e} | gauge-aligned rot {al[0]:.1e} t {al[1]:.1e} depth {al[2]:.1e}; cost {float(b.evaluate(0)[0]):.8f} (reference {float(g['min_final_loss']):.8f})")
    for fused in (True, False):
        src, trg = frames_from_synth(pair)
        n = int(g["traj_steps"])
        sfm = SfM({"aligment": {"pyramid_min": 0, "pyramid_max": 3, "cost_params": {}}}, src, [trg], [T(pair.pose_init)], num_iters=n)
        sfm.init_optimisation(kld_init=T(pair.kld_init))
        sfm.run(fused=fused)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sfm.run(fused=fused, levels=[2], num_iters=300)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        L = np.array([float(l) for l in sfm.losses])[: 3 * n]
        print(f"  Adam API loop {'fused' if fused else 'eager'}: {3 * n} reference steps, max loss rel dev {np.abs(L / g['traj_losses'] - 1).max():.1e}; "
              f"level-0 rate {300 / dt:.0f} it/s ({1e6 * dt / 300:.0f} us/iteration)")


def reference_vs_itself():
    g, alt = load_golden("g14_config1_converged"), load_golden("g14_config1_converged_t1")
    print("== yardstick: the REFERENCE against itself (config 1, same inputs, 8 CPU threads vs 1 thread = another fp32 summation order) ==")
    for tag, what in (("L2_end", "after 3 x 500 Adam iterations"), ("P1_end", "after the polish (converged)")):
        raw = pose_depth_errors(alt[tag + "_pose"], alt[tag + "_kld"], g[tag + "_pose"], g[tag + "_kld"], gauge=False)
        al = pose_depth_errors(alt[tag + "_pose"], alt[tag + "_kld"], g[tag + "_pose"], g[tag + "_kld"])
        print(f"  {what}: raw rot {raw[0]:.1e} t {raw[1]:.1e} depth {raw[2]:.1e} | scale-gauge aligned rot {al[0]:.1e} t {al[1]:.1e} depth {al[2]:.1e}")
    d = np.abs(g["losses"] / alt["losses"] - 1)
    print(f"  loss-curve relative deviation at iteration 1/10/100/200/500/1500: " + "/".join(f"{d[i]:.1e}" for i in (1, 10, 100, 200, 499, 1499)))


def main():
    torch.cuda.set_device(0)
    print("device:", torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["grad", "small", "traj", "self", "c1", "c2"]
    for key, fn in (("grad", gradients_at_size), ("small", small_goldens), ("traj", trajectories), ("self", reference_vs_itself),
                    ("c1", config1_converged), ("c2", config2_schedule)):
        if key in which:
            fn()


if __name__ == "__main__":
    main()
