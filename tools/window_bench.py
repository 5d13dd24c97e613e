"""Mapping windows at the reference's extent (config/tum/odom_desk.yaml: window_size 5, supporting frames with free poses and affine
pairs) through ``map_window``: Gauss-Newton and the fused Adam schedule, time per window / per iteration, errors against the
synthetic ground truth.   python tools/window_bench.py [n_supp ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth  # noqa: E402
from super_primitive_amd.image.keyframe import KeyFrame  # noqa: E402
from super_primitive_amd.odometery.loops import map_window  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rot_angle(A, B):
    R = A[:3, :3].T @ B[:3, :3]
    return float(np.arccos(np.clip(0.5 * (np.trace(R) - 1), -1, 1)))


def build(n_kf, n_supp, n_run, N, seed=300, **kw):
    frames, kfi, si, est, klds, affs = synth.reference_window_inputs(seed, n_kf, n_supp, n_run, N=N, **kw)
    kfs = [KeyFrame(T(frames[i].image), T(frames[i].K), T(frames[i].logdepth_perseg), T(frames[i].keypoints), T(frames[i].keypoint_regions)) for i in kfi]
    supp = [[(KeyFrame(T(frames[j].image), T(frames[j].K)), T(est[j]), T(affs[j])) for j in row] for row in si]
    return frames, kfi, si, kfs, [T(est[i]) for i in kfi], [T(k) for k in klds], [T(affs[i]) for i in kfi], supp


def errors(out, frames, kfi, si):
    pr = max(rot_angle(out['kf_poses'][k].cpu().numpy().astype(np.float64), frames[i].T_wc.astype(np.float64)) for k, i in enumerate(kfi))
    pt = max(float(np.abs(out['kf_poses'][k].cpu().numpy()[:3, 3] - frames[i].T_wc[:3, 3]).max()) for k, i in enumerate(kfi))
    sr = max([rot_angle(out['supp_poses'][k][j].cpu().numpy().astype(np.float64), frames[i].T_wc.astype(np.float64)) for k, row in enumerate(si) for j, i in enumerate(row)] or [0])
    st = max([float(np.abs(out['supp_poses'][k][j].cpu().numpy()[:3, 3] - frames[i].T_wc[:3, 3]).max()) for k, row in enumerate(si) for j, i in enumerate(row)] or [0])
    dd = max(float(np.abs(np.expm1(out['klds'][k].cpu().numpy() - frames[i].kld_gt)).max()) for k, i in enumerate(kfi))
    return pr, pt, sr, st, dd


bga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bgb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def main():
    cases = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for N in (40, 100):
        for n_supp in cases:
            frames, kfi, si, kfs, poses, klds, affs, supp = build(5, n_supp, 2, N)
            n_free = 4 + sum(len(r) for r in si)
            for opt, iters in (("gn", 25), ("adam", 500)):
                for rep in range(2):
                    if os.environ.get("SP_BG"):              # a background load on a side stream keeps the shader clock up (diagnostic)
                        side = torch.cuda.Stream()
                        with torch.cuda.stream(side):
                            for _ in range(int(os.environ["SP_BG"])):
                                bgc = bga @ bgb
                    torch.cuda.synchronize() if not os.environ.get("SP_BG") else time.sleep(0.05)
                    t0 = time.perf_counter()
                    out = map_window(kfs, poses, klds, affs, supp, iters, window_size=5, optimiser=opt, gn_schedule=dict(profile=True) if opt == "gn" else None)
                    torch.cuda.current_stream().synchronize(); dt = time.perf_counter() - t0
                    torch.cuda.synchronize()
                e = errors(out, frames, kfi, si)
                L = [float(l) for l in out['losses']]
                print(f"N={N} 5 KFs x {n_supp} supp + 2 running: {n_free} free nodes = {8 * n_free} camera unknowns, {opt}: {1e3 * dt:.1f} ms per window, "
                      f"{len(L)} iterations ({1e6 * dt / len(L):.0f} us each), loss {L[0]:.6f} -> {L[-1]:.6f}; vs ground truth: kf {e[0]:.1e} rad {e[1]:.1e} t, "
                      f"supp {e[2]:.1e} rad {e[3]:.1e} t, depth {e[4]:.1e}", flush=True)
                if 'gn_profile' in out:
                    print("      update kernel, last step [us]: " + ", ".join(f"{k} {v:.1f}" for k, v in out['gn_profile'].items()), flush=True)


if __name__ == "__main__":
    main()
