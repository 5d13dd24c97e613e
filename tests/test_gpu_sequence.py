"""-m gpu: BASELINE configs[2] as a SEQUENCE: 30 frames of 224x288 with 40 segments, the MonoVO chain
track -> keyframe criterion -> depth render -> per-segment re-initialisation -> windowed mapping run end to end
(``odometery/sequence.py``; reference ``odometery/odometery.py:986-1075``) on both optimisers, against the synthetic ground-truth
trajectory."""
import numpy as np
import pytest
import torch

from gpu_util import T, npy
from parity_util import rot_angle

pytestmark = pytest.mark.gpu


def make_sequence_inputs(n=30, H=224, W=288, N=40, seed=31):
    from super_primitive_amd import synth
    from super_primitive_amd.image.keyframe import KeyFrame
    rng = np.random.default_rng(seed)
    base = 0.6 * np.array([0.05, -0.02, 0.015, 0.01, -0.015, 0.008])
    # a smooth trajectory with a little jitter per frame (a multiplicative jitter on k * base would grow with k and end far outside
    # what a constant-velocity prior can bridge on this 14-px-period texture)
    twists = [k * base + 0.003 * rng.standard_normal(6) * (k > 0) for k in range(n)]
    seq = synth.make_sequence(H, W, N, twists, keyframe_ids=list(range(n)), seed=seed, overlap=1)
    frames = [KeyFrame(T(f.image), T(f.K)) for f in seq]
    to_kf = lambda i: KeyFrame(T(seq[i].image), T(seq[i].K), T(seq[i].logdepth_perseg), T(seq[i].keypoints), T(seq[i].keypoint_regions))
    return seq, frames, to_kf


@pytest.mark.parametrize("engine", ["gn", "adam"])
def test_config3_sequence_trajectory_against_ground_truth(engine):
    from super_primitive_amd.odometery.sequence import run_sequence
    seq, frames, to_kf = make_sequence_inputs()
    run_sequence(frames[:4], to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine=engine)        # (first-use costs out of the timings)
    log = []
    out = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine=engine, translation_thresh=0.1, window_size=3, map_steps=300, log=log)
    for i, ev, d in log:
        if ev == 'keyframe':
            e = npy(d['kld']) - seq[i].kld_gt
            print(f"   frame {i}: new keyframe, criterion {[round(float(v), 4) for v in d['criterion']]}, visible segments {d['visible']}, render valid {d['valid_ratio']:.3f}, "
                  f"log-depth error at creation: median {np.median(e):+.2e}, max |.| {np.abs(e).max():.2e}")
        else:
            errs = [float(np.abs(npy(k) - seq[j].kld_gt).max()) for j, k in zip(d['kf_ids'], d['klds'])]
            perr = [float(np.abs(npy(p)[:3, 3] - seq[j].T_wc[:3, 3]).max()) for j, p in zip(d['kf_ids'], d['kf_poses'])]
            print(f"   frame {i}: mapping over keyframes {d['kf_ids']}: {d['n']} iterations, loss {d['losses'][0]:.6f} -> {d['losses'][1]:.6f}; log-depth errors after {['%.1e' % v for v in errs]}, "
                  f"keyframe translation errors {['%.1e' % v for v in perr]}")
    P = npy(out["track_poses"]).astype(np.float64)
    G = np.stack([f.T_wc for f in seq]).astype(np.float64)
    # one global scale (the first keyframe's depths fix it to ~1; drift through the rendered keyframes is what alignment removes)
    s = float((P[:, :3, 3] * G[:, :3, 3]).sum() / max((P[:, :3, 3] ** 2).sum(), 1e-30))
    rot = max(rot_angle(a, b) for a, b in zip(P, G))
    tt = float(np.abs(s * P[:, :3, 3] - G[:, :3, 3]).max())
    n = len(frames) - 1
    sec = out["seconds"]
    print(f"\nconfig 3 sequence, {engine}: {n} frames tracked, keyframes at {out['all_kf_ids']}, {out['n_mappings']} mappings; trajectory vs ground truth: "
          f"rot {rot:.2e} rad, t {tt:.2e} (scale {s:.5f}); tracking {n / sec['track']:.0f} frames/s, keyframe work {1e3 * sec['keyframe'] / n:.2f} ms/frame, "
          f"mapping {1e3 * sec['mapping'] / max(out['n_mappings'], 1):.1f} ms/window")
    per = [(i, rot_angle(P[i], G[i]), float(np.abs(P[i, :3, 3] - G[i, :3, 3]).max())) for i in range(len(P))]
    print("   per-frame (rot, t) error: " + " ".join(f"{i}:{r:.1e}/{t:.1e}" for i, r, t in per))
    print("   keyframe log-depth errors: " + " ".join(f"{i}:{float(np.abs(npy(k) - seq[i].kld_gt).max()):.1e}" for i, k in zip(out["kf_ids"], out["kf_klds"])))
    assert len(out["all_kf_ids"]) >= 3 and out["n_mappings"] >= 1
    assert abs(s - 1.0) < 5e-3
    # Gauss-Newton converges every frame: 1e-3 with a wide margin.  The reference's own tracking schedule ([0, 0, 300] Adam steps at lr
    # 5e-3, no decay) keeps jittering ~1e-3 around the optimum (test_config3_tracking_300_steps_at_size; golden g17's end state is
    # 8e-4 rad from a rerun of itself): its per-frame errors are that jitter, not drift
    bar = 1e-3 if engine == "gn" else 3e-3
    assert rot <= bar and tt <= bar
    # the keyframes' depths (re-initialised from a render, then mapped) against the ground truth
    for i, kld in zip(out["kf_ids"], out["kf_klds"]):
        np.testing.assert_allclose(npy(kld), seq[i].kld_gt, atol=5e-3)
