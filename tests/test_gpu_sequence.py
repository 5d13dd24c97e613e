"""-m gpu: BASELINE configs[2] as a SEQUENCE: 224x288 frames with 40 segments through the MonoVO chain of ``odometery/sequence.py`` (the
reference's ``Odometery.run``, ``odometery/odometery.py:1018-1075``: track -> supplementary mapping -> scheduled mapping -> keyframe
criterion -> depth render -> per-segment re-initialisation, with the reference's supporting-frame selection) on both optimisers:
against the synthetic ground truth, against golden g21 (the same chain through the imported reference functions), and over 64 frames
with the reference's window size so that the window slides."""
import numpy as np
import pytest
import torch

from gpu_util import T, npy
from parity_util import rot_angle

pytestmark = pytest.mark.gpu


def make_sequence_inputs(n=30, H=224, W=288, N=40, seed=31, rot_scale=1.0):
    from super_primitive_amd import synth
    from super_primitive_amd.image.keyframe import KeyFrame
    rng = np.random.default_rng(seed)
    base = 0.6 * np.array([0.05, -0.02, 0.015, 0.01 * rot_scale, -0.015 * rot_scale, 0.008 * rot_scale])
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
    # Gauss-Newton converges every step: 1e-3 with a wide margin, scale kept.  The reference's own schedules ([0, 0, 300] Adam tracking steps
    # at lr 5e-3 without decay, 10 supplementary mapping steps on the latest keyframe's depths after every frame) keep jittering ~1e-3 per
    # frame (test_config3_tracking_300_steps_at_size; golden g17's end state is 8e-4 rad from a rerun of itself) and the supplementary
    # mapping turns that jitter into a slow drift of the monocular scale -- which golden g21 shows for the reference chain itself
    # (test_config3_sequence_follows_the_reference_chain); the similarity alignment removes it
    bar = (1e-3, 1e-3, 5e-3, 5e-3) if engine == "gn" else (6e-3, 2.5e-2, 0.15, 3e-2)
    assert rot <= bar[0] and tt <= bar[1] and abs(s - 1.0) <= bar[2], (rot, tt, s)
    # the keyframes' depths (re-initialised from a render, then mapped) against the ground truth, in the trajectory's scale
    for i, kld in zip(out["kf_ids"], out["kf_klds"]):
        np.testing.assert_allclose(npy(kld) + np.log(s), seq[i].kld_gt, atol=bar[3])

def _aligned_errors(P, G):
    s = float((P[:, :3, 3] * G[:, :3, 3]).sum() / max((P[:, :3, 3] ** 2).sum(), 1e-30))
    return s, max(rot_angle(a, b) for a, b in zip(P, G)), float(np.abs(s * P[:, :3, 3] - G[:, :3, 3]).max())


@pytest.mark.parametrize("engine", ["gn", "adam"])
def test_config3_sequence_with_the_reference_window_size_slides_the_window(engine):
    """config/tum/odom_desk.yaml's own extent -- window_size 5, supp_every_n 3 (at most two supporting frames per
    keyframe, odometery.py:1327-1360, + the two running ones), continual_steps 10, affine compensation, all supporting poses free -- over 64
    frames: 7-8 keyframes, so the window fills (oldest depths frozen, 14 free nodes = 112 camera unknowns in the Gauss-Newton solver) and
    slides.  Round 3 raised ValueError around frame 40 here (more than 128 unknowns: it appended every third frame as a supporting frame)."""
    from super_primitive_amd.odometery.sequence import run_sequence
    n = 64
    seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
    log = []
    out = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine=engine, translation_thresh=0.095, window_size=5, log=log,
                       depth_of=lambda i: T(seq[i].kld_gt))
    P = npy(out["track_poses"]).astype(np.float64)
    G = np.stack([f.T_wc for f in seq]).astype(np.float64)
    s, rot, tt = _aligned_errors(P, G)
    sec = out["seconds"]
    maps = [d for _, ev, d in log if ev == 'mapping']
    print(f"\nconfig 3, {n} frames, window_size 5, {engine}: keyframes {out['all_kf_ids']}, window {out['kf_ids']}, supporting frames per keyframe {out['supp_ids']}, "
          f"{out['n_mappings']} scheduled + {out['n_supp_mappings']} supplementary mappings; nodes per scheduled mapping {[len(d['kf_ids']) + sum(d['n_supp']) for d in maps]}; "
          f"trajectory vs ground truth rot {rot:.2e} rad, t {tt:.2e} (scale {s:.5f}); {(n - 1) / sum(sec.values()):.0f} frames/s end to end "
          f"(tracking {1e3 * sec['track'] / (n - 1):.2f}, supplementary mapping {1e3 * sec['supp_mapping'] / (n - 1):.2f}, keyframe work {1e3 * sec['keyframe'] / (n - 1):.2f} ms/frame; "
          f"scheduled mapping {1e3 * sec['mapping'] / max(out['n_mappings'], 1):.1f} ms/window)")
    for d in maps[-2:]:
        print(f"   mapping over keyframes {d['kf_ids']} + supporting {d['n_supp']}: {d['n']} iterations, loss {d['losses'][0]:.6f} -> {d['losses'][1]:.6f} {d.get('gn') or ''}")
    assert len(out["all_kf_ids"]) >= 7 and len(out["kf_ids"]) == 5 and out["kf_ids"] == out["all_kf_ids"][-5:]          # the window slid
    assert all(len(r) <= 2 for r in out["supp_ids"]) and sum(len(r) == 2 for r in out["supp_ids"][:-1]) >= 3          # the reference's selection
    assert out["n_mappings"] >= 6 and out["n_supp_mappings"] == n - 1
    full = [d for d in maps if len(d['kf_ids']) == 5]
    assert full and max(len(d['kf_ids']) + sum(d['n_supp']) for d in full) >= 14                                       # 5 keyframes + 8 supporting + 2 running - ...
    if engine == "gn":
        assert all(not d['gn']['too_many_unknowns'] for d in maps)
    # Gauss-Newton converges every step of the chain; the reference's Adam schedules jitter ~1e-3 per frame (lr 5e-3, no decay) and the
    # supplementary mapping lets that jitter leak into the latest keyframe's depths: a slow scale drift, removed by the alignment
    bar = (1.5e-3, 2e-3, 0.01) if engine == "gn" else (6e-3, 2.5e-2, 0.15)
    assert rot <= bar[0] and tt <= bar[1] and abs(s - 1) <= bar[2], (rot, tt, s)


def test_config3_sequence_follows_the_reference_chain():
    """Golden g21 (oracle/gen_goldens_sequence.py): the same chain, 24 frames, every numerical step by the imported reference functions
    (300 Adam tracking steps per frame, 10 supplementary + 500 scheduled mapping steps) on the CPU.  The fused Adam engine must take the
    same decisions -- keyframes at the same frames, the same supporting frames -- and stay within the reference's own tracking jitter of its
    poses; the Gauss-Newton engine the same decisions."""
    from conftest import load_golden
    from super_primitive_amd.odometery.sequence import run_sequence
    g = load_golden("g21_config3_sequence_chain")
    n = int(g["n_frames"])
    H, W, N = (int(v) for v in g["HWN"])
    seq, frames, to_kf = make_sequence_inputs(n, H, W, N, seed=int(g["seed"]))
    np.testing.assert_array_equal(np.stack([f.T_wc for f in seq]), g["gt_poses"])
    ref_kf_frames = [i for i in range(1, n) if bool(g[f"f{i}_new_kf"])]
    ref_track = np.stack([g[f"f{i}_tracked_pose"] for i in range(1, n)]).astype(np.float64)
    for engine in ("adam", "gn"):
        log = []
        out = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine=engine, translation_thresh=0.095, window_size=5, log=log,
                           depth_of=lambda i: T(seq[i].kld_gt))
        assert out["all_kf_ids"] == [0] + ref_kf_frames, (engine, out["all_kf_ids"], ref_kf_frames)
        assert [",".join(str(t) for t in row) for row in out["supp_ids"]] == [str(v) for v in g["final_supp_ids"]], (engine, out["supp_ids"])
        P = npy(out["track_poses"]).astype(np.float64)[1:]
        gt = np.stack([f.T_wc for f in seq]).astype(np.float64)[1:]
        # The monocular scale is a gauge of this chain: the supplementary mapping after frame 1 re-fits the first keyframe's depths against
        # ONE frame 0.03 away, so the ~5e-4 jitter of that frame's tracked pose (lr 5e-3 Adam, not converged) moves all depths by
        # (pose error / baseline) ~ a few per cent -- the reference chain's own first keyframe ends 1-2 % off its ground truth -- and every later
        # pose and depth inherits that factor.  Two faithful runs therefore agree up to one global scale: align it (the reference evaluates
        # its trajectories with scale correction too) and compare the rest.
        s_ref = float((P[:, :3, 3] * ref_track[:, :3, 3]).sum() / (P[:, :3, 3] ** 2).sum())
        rot = max(rot_angle(a, b) for a, b in zip(P, ref_track))
        tt = float(np.abs(s_ref * P[:, :3, 3] - ref_track[:, :3, 3]).max())
        kld_err = max(float(np.abs(npy(k) + np.log(s_ref) - w).max()) for k, w in zip(out["kf_klds"], g["final_kf_klds"]))
        kf_t = float(np.abs(s_ref * npy(out["kf_poses"])[:, :3, 3] - g["final_kf_poses"][:, :3, 3]).max())
        s_gt = lambda X: float((X[:, :3, 3] * gt[:, :3, 3]).sum() / (X[:, :3, 3] ** 2).sum())
        ref_vs_gt = (max(rot_angle(a, b) for a, b in zip(ref_track, gt)), float(np.abs(s_gt(ref_track) * ref_track[:, :3, 3] - gt[:, :3, 3]).max()))
        print(f"\nchain of {n} frames, {engine}: keyframes {out['all_kf_ids']} (reference: {[0] + ref_kf_frames}), supporting frames {out['supp_ids']}; scale vs the reference "
              f"chain {s_ref:.4f} (the reference chain vs ground truth: {s_gt(ref_track):.4f}, this run: {s_gt(P):.4f}); tracked poses vs the reference chain's, scale aligned: "
              f"rot {rot:.2e} rad, t {tt:.2e}; final keyframe log-depths {kld_err:.2e}, keyframe translations {kf_t:.2e} (the reference chain itself vs ground truth: "
              f"rot {ref_vs_gt[0]:.2e}, t {ref_vs_gt[1]:.2e})")
        # the reference's tracking schedule does not converge (lr 5e-3 Adam, 300 steps: ~1e-3 of jitter per frame, golden g17's rerun spread),
        # so two faithful runs of the chain differ by that jitter; Gauss-Newton converges and sits at the centre of it
        assert abs(s_ref - 1) <= 0.15 and rot <= 4e-3 and tt <= 1e-2 and kld_err <= 3e-2 and kf_t <= 1e-2


def _load_chain_state(vo, g, tag, frames, kf_of):
    """Put the MonoVO into the reference chain's recorded state ``<tag>`` (oracle/gen_goldens_sequence.py ``snapshot``)."""
    from super_primitive_amd.odometery.sequence import _Supp
    ids = [int(i) for i in g[f"{tag}_kf_ids"]]
    vo.kf_ids, vo.kfs = list(ids), [kf_of(i) for i in ids]
    vo.kf_poses = [T(p) for p in g[f"{tag}_kf_poses"]]
    vo.kf_klds = [T(k) for k in g[f"{tag}_kf_klds"]]
    vo.kf_affs = [T(a) for a in g[f"{tag}_kf_affs"]]
    row = lambda key: [_Supp(frames[int(ts)], T(p), T(a), int(ts)) for ts, p, a in zip(g[f"{tag}_{key}_ts"], g[f"{tag}_{key}_poses"], g[f"{tag}_{key}_affs"])]
    flat, vo.supp_opt, q = row("supp"), [], 0
    for c in g[f"{tag}_supp_counts"]:
        vo.supp_opt.append(flat[q: q + int(c)]); q += int(c)
    vo.tracked, vo.curr_supp = row("tracked"), row("curr")
    vo.current_track, vo.current_aff = T(g[f"{tag}_current_track"]), T(g[f"{tag}_current_aff"])
    fl = g[f"{tag}_flags"]
    vo.current_ts, vo.initialised, vo.mapping_scheduled = int(fl[0]), bool(fl[1]), bool(fl[2])
    vo.tracker, vo.supp_mapper = None, None


def test_config3_chain_stage_by_stage_from_the_references_own_states():
    """VERDICT r05 item 5 -- TEACHER-FORCED chain parity at the north-star bar.  Golden g21 holds the reference chain's complete state
    before and after every stage of every frame; here EVERY stage is restarted from the reference's own recorded input (Adam moments start
    at zero in every stage of the reference too: a fresh optimiser per tracking / mapping call, odometery.py:300-312,576-648) and its
    output compared with the reference's: tracking (300 Adam steps), supplementary mapping (10), scheduled mapping (the reference's own
    iteration count), keyframe decision + creation (depth render, criterion, per-segment median).  Free-running, the chain inherits the
    jitter of every un-converged stage before it (the 4e-3 / 1e-2 / 3e-2 of the test above); stage by stage it must hold
    1e-4 rad / 1e-4 t / 1e-3 depth -- with ONE qualification, measured, not assumed: the reference's tracking stage does not converge (300
    Adam steps at lr 5e-3 on an L1 cost jitter by ~3e-4 rad), and the REFERENCE ITSELF, re-run from its own recorded input with another
    thread count, ends 1.5e-4 rad / 1.3e-4 t from its first run (golden ``spread_track``, oracle/gen_goldens_sequence.py ``stage_spread``).
    Adam's first steps move the pose by its learning rate, 5e-3 rad per step, whatever the gradient's size: the iterate overshoots and
    rings, and a last-bit difference in one gradient flips a step's sign a few iterations later -- even the first 50 steps of the
    reference's tracker differ by 3e-4 rad between two of its own runs (``spread_track50``).  So the tracker is held to max(bar, 3 x the
    reference's own spread) over the full stage, the tracker to max(bar, 5 x spread) over its first 50 steps and over the full stage; the mapping stages (whose loops the reference reproduces bit
    for bit across thread counts: spread 0) and the keyframe stage are held to the bar."""
    from conftest import load_golden
    from super_primitive_amd.odometery.sequence import MonoVO
    g = load_golden("g21_config3_sequence_chain")
    if "f1_s0_kf_ids" not in g or "spread_track" not in g:
        pytest.skip("golden g21 without stage snapshots (regenerate: python oracle/gen_goldens_sequence.py)")
    n = int(g["n_frames"])
    H, W, N = (int(v) for v in g["HWN"])
    seq, frames, to_kf = make_sequence_inputs(n, H, W, N, seed=int(g["seed"]))
    cache = {}
    kf_of = lambda i: cache.setdefault(i, to_kf(i))
    vo = MonoVO(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="adam", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt))
    BAR_R, BAR_T, BAR_D = 1e-4, 1e-4, 1e-3
    sp_track, sp_map = g["spread_track"], g["spread_map"]
    # (the recorded spread is the largest deviation seen over 23 stages of ONE second run of a chaotic iteration -- a sample maximum, not a
    #  bound: the tracker, whose spread is 3-15 x the bar, gets 5 x it; the mapping stages, whose spread is below the bar, 3 x)
    tol_track = (max(BAR_R, 5 * float(sp_track[0])), max(BAR_T, 5 * float(sp_track[1])), max(5e-4, 5 * float(sp_track[2])))
    sp50 = g["spread_track50"]
    tol_track50 = (max(BAR_R, 5 * float(sp50[0])), max(BAR_T, 5 * float(sp50[1])), max(2e-4, 5 * float(sp50[2])))
    tol_map = (max(BAR_R, 3 * float(sp_map[0])), max(BAR_T, 3 * float(sp_map[1])), max(BAR_D, 3 * float(sp_map[2])), max(5e-4, 3 * float(sp_map[3])))
    print(f"\nthe reference against itself per stage (threads {g['spread_threads'].tolist()}): tracking {sp_track}, its first 50 steps {g['spread_track50']}, supplementary mapping "
          f"{g['spread_supp']}, scheduled mapping {sp_map}")
    worst = dict(track=[0.0, 0.0, 0.0], track50=[0.0, 0.0, 0.0], supp=[0.0], map=[0.0, 0.0, 0.0, 0.0], keyframe=[0.0, 0.0])
    n_stage = dict(track=0, supp=0, map=0, keyframe=0)
    pose_err = lambda A, B: (rot_angle(np.asarray(A, np.float64), np.asarray(B, np.float64)), float(np.abs(np.asarray(A, np.float64)[:3, 3] - np.asarray(B, np.float64)[:3, 3]).max()))
    for i in range(1, n):
        # ---- tracking: s0 -> its first 50 steps (the bar), s0 -> s1 (the bar or 3 x the reference's own spread)
        _load_chain_state(vo, g, f"f{i}_s0", frames, kf_of)
        vo.c["track_steps"] = (0, 0, 50)
        vo.track_frame(i)
        vo.c["track_steps"] = (0, 0, 300)
        r, t = pose_err(npy(vo.current_track), g[f"f{i}_track50_pose"])
        a = float(np.abs(npy(vo.current_aff) - g[f"f{i}_track50_aff"]).max())
        worst["track50"] = [max(x, y) for x, y in zip(worst["track50"], (r, t, a))]
        assert r <= tol_track50[0] and t <= tol_track50[1] and a <= tol_track50[2], (i, "track, first 50 steps", r, t, a, tol_track50)
        _load_chain_state(vo, g, f"f{i}_s0", frames, kf_of)
        vo.track_frame(i)
        r, t = pose_err(npy(vo.current_track), g[f"f{i}_s1_current_track"])
        a = float(np.abs(npy(vo.current_aff) - g[f"f{i}_s1_current_aff"]).max())
        worst["track"] = [max(x, y) for x, y in zip(worst["track"], (r, t, a))]
        n_stage["track"] += 1
        assert r <= tol_track[0] and t <= tol_track[1] and a <= tol_track[2], (i, "track", r, t, a, tol_track)
        last = "s1"
        # ---- supplementary mapping: s1 -> s2
        if f"f{i}_s2_kf_ids" in g:
            _load_chain_state(vo, g, f"f{i}_s1", frames, kf_of)
            vo.mapping(vo.c["continual_steps"], mode="supp")
            d = float(np.abs(np.expm1(npy(vo.kf_klds[-1]) - g[f"f{i}_s2_kf_klds"][-1])).max())
            worst["supp"][0] = max(worst["supp"][0], d)
            n_stage["supp"] += 1
            assert d <= BAR_D, (i, "supp", d)
            np.testing.assert_allclose(npy(vo.current_track), g[f"f{i}_s2_current_track"], atol=1e-6)      # (update_track_pose: the tracked pose)
            last = "s2"
        # ---- scheduled mapping: s2 -> s3, over the reference's own number of iterations (its early stop is noise-sensitive: two
        #      regenerations of this golden on 2 / 4 threads stop 61 iterations apart)
        if f"f{i}_s3_kf_ids" in g:
            _load_chain_state(vo, g, f"f{i}_s2", frames, kf_of)
            n_it = len(g[f"f{i}_map_losses"])
            vo.c["map_rel_tol"] = 0.0
            vo.mapping(n_it, mode="map")
            vo.c["map_rel_tol"] = 1e-8
            pe = [pose_err(npy(p), q) for p, q in zip(vo.kf_poses, g[f"f{i}_s3_kf_poses"])]
            se = [pose_err(npy(s.pose), q) for s, q in zip([s for row in vo.supp_opt for s in row], g[f"f{i}_s3_supp_poses"])]
            d = max(float(np.abs(np.expm1(npy(k) - w)).max()) for k, w in zip(vo.kf_klds, g[f"f{i}_s3_kf_klds"]))
            fa = max(float(np.abs(npy(a_) - w).max()) for a_, w in zip(vo.kf_affs, g[f"f{i}_s3_kf_affs"]))
            rr, tt = max(e[0] for e in pe + se), max(e[1] for e in pe + se)
            worst["map"] = [max(x, y) for x, y in zip(worst["map"], (rr, tt, d, fa))]
            n_stage["map"] += 1
            assert rr <= tol_map[0] and tt <= tol_map[1] and d <= tol_map[2] and fa <= tol_map[3], (i, "map", n_it, rr, tt, d, fa, tol_map)
            last = "s3"
        # ---- keyframe decision / creation: -> s4
        _load_chain_state(vo, g, f"f{i}_{last}", frames, kf_of)
        if last == "s3":                                      # (step() resets the pools after a scheduled mapping, odometery.py:1052-1055)
            vo.mapping_scheduled = False
            vo.reset_tracked_poses(); vo.reset_running_supp_kfs()
        new_kf, crit = vo.keyframe_stage(i)
        assert bool(new_kf) == bool(g[f"f{i}_new_kf"]), (i, new_kf)
        ref_crit = g[f"f{i}_criterion"]                         # (validity ratio, scale, translation difference)
        np.testing.assert_allclose([crit[0], crit[1], crit[2]], ref_crit, rtol=2e-4, atol=2e-5)
        n_stage["keyframe"] += 1
        worst["keyframe"][0] = max(worst["keyframe"][0], float(np.abs((np.asarray(crit[:3]) - ref_crit) / np.maximum(np.abs(ref_crit), 1e-3)).max()))
        assert vo.kf_ids == [int(v) for v in g[f"f{i}_s4_kf_ids"]] and [len(r) for r in vo.supp_opt] == [int(c) for c in g[f"f{i}_s4_supp_counts"]]
        assert [s.ts for row in vo.supp_opt for s in row] == [int(v) for v in g[f"f{i}_s4_supp_ts"]]
        if new_kf:
            d = float(np.abs(np.expm1(npy(vo.kf_klds[-1]) - g[f"f{i}_kf_kld"])).max())
            worst["keyframe"][1] = max(worst["keyframe"][1], d)
            assert d <= BAR_D, (i, "keyframe depths", d)
            assert vo.mapping_scheduled
    print(f"\nteacher-forced chain, {n - 1} frames: stages compared {n_stage}; worst deviation from the reference's own stage output: tracking, first 50 steps, rot "
          f"{worst['track50'][0]:.1e} rad, t {worst['track50'][1]:.1e}, affine {worst['track50'][2]:.1e}; tracking, all 300 steps, rot {worst['track'][0]:.1e} rad, "
          f"t {worst['track'][1]:.1e}, affine {worst['track'][2]:.1e} (asserted {tol_track}); supplementary mapping depth {worst['supp'][0]:.1e}; scheduled mapping rot {worst['map'][0]:.1e}, t {worst['map'][1]:.1e}, "
          f"depth {worst['map'][2]:.1e}, affine {worst['map'][3]:.1e}; keyframe criterion (relative) {worst['keyframe'][0]:.1e}, new keyframe depths {worst['keyframe'][1]:.1e}")
    assert n_stage["track"] == n - 1 and n_stage["supp"] == n - 1 and n_stage["map"] >= 2 and n_stage["keyframe"] == n - 1


@pytest.mark.parametrize("engine", ["gn", "adam"])
def test_config3_sequence_with_mono_initialisation(engine):
    """The reference's ``mono_init: True`` start (config/tum/odom_desk.yaml; odometery.py:136-139,1003-1007,1066-1071,578-581): the first
    keyframe gets UNIT depths at its keypoints, ``init_frames`` frames are tracked against it, the frame after becomes the second keyframe
    (unit depths again) and mapping(mode='init') -- two keyframes, no supporting frames, pose rate 1e-2, ``init_steps`` = 1000 iterations, no
    early stop -- recovers both depth maps and the relative pose up to the one monocular scale.  From there the chain runs as usual."""
    from super_primitive_amd.odometery.sequence import run_sequence
    n = 24
    seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.5)
    log = []
    out = run_sequence(frames, to_kf, T(seq[0].T_wc), torch.zeros(seq[0].kld_gt.shape[0], device="cuda"), engine=engine, translation_thresh=0.095,
                       window_size=5, mono_init=True, init_frames=7, log=log)
    P = npy(out["track_poses"]).astype(np.float64)
    G = np.stack([f.T_wc for f in seq]).astype(np.float64)
    s, _, _ = _aligned_errors(P[8:], G[8:])
    rot = max(rot_angle(a, b) for a, b in zip(P[8:], G[8:]))
    tt = float(np.abs(s * P[8:, :3, 3] - G[8:, :3, 3]).max())
    init = [d for _, ev, d in log if ev == 'mapping' and d['mode'] == 'init']
    kld_err = max(float(np.abs(npy(k) + np.log(s) - seq[i].kld_gt).max()) for i, k in zip(out["kf_ids"], out["kf_klds"]))
    print(f"\nmono initialisation, {engine}: keyframes {out['all_kf_ids']}, init mapping {init[0]['n']} iterations, loss {init[0]['losses'][0]:.5f} -> {init[0]['losses'][1]:.5f}; "
          f"monocular scale {s:.3f} (unit depths for a plane ~3 away); after the initialisation, scale aligned: rot {rot:.2e} rad, t {tt:.2e}, keyframe log-depths {kld_err:.2e}")
    assert out["all_kf_ids"][:2] == [0, 7] and out["n_init_mappings"] == 1 and len(init) == 1 and init[0]['kf_ids'] == [0, 7]
    assert 2.0 < s < 4.0
    bar = (5e-4, 5e-3, 1e-2) if engine == "gn" else (6e-3, 1.5e-2, 3e-2)
    assert rot <= bar[0] and tt <= bar[1] and kld_err <= bar[2], (rot, tt, kld_err)


class _LazySequence:
    """Frames of a long synthetic sequence made on demand (``synth.make_sequence`` one frame at a time: the plane and its texture are a
    function of the seed alone) and dropped again: ``frames[i]`` = supporting-frame-like KeyFrame (image + K), ``keyframe(i)`` = with the
    segment data; the host copies of the last few frames only are kept."""

    def __init__(self, twists, H=224, W=288, N=40, seed=31):
        self.twists, self.H, self.W, self.N, self.seed = twists, H, W, N, seed
        self._host = {}

    def __len__(self):
        return len(self.twists)

    def host(self, i, keyframe=False):
        from super_primitive_amd import synth
        key = (i, keyframe)
        if key not in self._host:
            if len(self._host) > 16:
                self._host.pop(next(iter(self._host)))
            self._host[key] = synth.make_sequence(self.H, self.W, self.N, [self.twists[i]], keyframe_ids=[0] if keyframe else [], seed=self.seed, overlap=1)[0]
        return self._host[key]

    def __getitem__(self, i):
        from super_primitive_amd.image.keyframe import KeyFrame
        f = self.host(i)
        return KeyFrame(T(f.image), T(f.K))

    def keyframe(self, i):
        from super_primitive_amd.image.keyframe import KeyFrame
        f = self.host(i, True)
        return KeyFrame(T(f.image), T(f.K), T(f.logdepth_perseg), T(f.keypoints), T(f.keypoint_regions))


def test_config3_at_sequence_length():
    """BASELINE configs[2] names the FULL fr1/desk sequence (config/tum/odom_desk.yaml:6; ~600 frames).  640 frames of
    the synthetic plane on a bounded (Lissajous) trajectory at the fr1/desk frame-to-frame motion through the whole MonoVO chain with the
    reference's extent (window 5, two supporting frames per keyframe, supplementary mapping after every frame), Gauss-Newton engine:
    drift stays bounded -- every tracked pose against the ground truth after ONE similarity alignment, and frame to frame -- the device
    memory pool does not grow after frame 100 (per-keyframe caches, windows and trackers are released with their keyframes), and the
    stage timings do not creep."""
    import time
    from super_primitive_amd import synth
    from super_primitive_amd.odometery.sequence import MonoVO
    n = 640
    k = np.arange(n, dtype=np.float64)
    tw = np.stack([0.8 * np.sin(2 * np.pi * k / 200.0), 0.5 * np.sin(2 * np.pi * k / 140.0 + 1.0) - 0.5 * np.sin(1.0), 0.3 * np.sin(2 * np.pi * k / 260.0),
                   0.05 * np.sin(2 * np.pi * k / 170.0), 0.06 * np.sin(2 * np.pi * k / 230.0), 0.04 * np.sin(2 * np.pi * k / 190.0)], axis=1)
    seq = _LazySequence([t for t in tw])
    f0 = seq.host(0, True)
    assert np.abs(tw[0]).max() == 0.0
    vo = MonoVO(seq, seq.keyframe, T(f0.T_wc), T(f0.kld_gt), engine="gn", translation_thresh=0.095, window_size=5,
                depth_of=lambda i: T(seq.host(i, True).kld_gt))
    mem, stage = {}, {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(1, n):
        vo.step(i)
        if i in (100, 300, n - 1):
            torch.cuda.synchronize()
            mem[i] = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
            stage[i] = dict(vo.secs, wall=time.perf_counter() - t0, keyframes=len(vo.all_kf_ids))
    out = vo.result()
    P = npy(out["track_poses"]).astype(np.float64)
    G = np.stack([synth.se3_exp_np(t) for t in tw])
    s, rot, tt = _aligned_errors(P, G)
    # frame-to-frame: the relative motion of consecutive tracked poses against the ground truth's (scale aligned)
    rel = lambda X: [np.linalg.inv(X[i]) @ X[i + 1] for i in range(len(X) - 1)]
    rp, rg = rel(P), rel(G)
    rel_rot = max(rot_angle(a, b) for a, b in zip(rp, rg))
    rel_t = max(float(np.abs(s * a[:3, 3] - b[:3, 3]).max()) for a, b in zip(rp, rg))
    per = lambda a, b: {key: 1e3 * (stage[b][key] - stage[a][key]) / (b - a) for key in ("track", "supp_mapping", "keyframe")}
    early, late = per(100, 300), per(300, n - 1)
    n_map = out["n_mappings"]
    print(f"\nconfig 3 at sequence length: {n} frames, {len(out['all_kf_ids'])} keyframes, {n_map} scheduled + {out['n_supp_mappings']} supplementary mappings, "
          f"{(n - 1) / stage[n - 1]['wall']:.0f} frames/s end to end (wall, frame synthesis included), {(n - 1) / sum(vo.secs.values()):.0f} frames/s in the chain; vs ground truth "
          f"after one similarity alignment: rot {rot:.2e} rad, t {tt:.2e} (scale {s:.4f}); frame to frame: rot {rel_rot:.2e}, t {rel_t:.2e}; "
          f"ms per frame (track / supplementary mapping / keyframe work) frames 100-300: {early['track']:.2f} / {early['supp_mapping']:.2f} / {early['keyframe']:.2f}, "
          f"frames 300-{n - 1}: {late['track']:.2f} / {late['supp_mapping']:.2f} / {late['keyframe']:.2f}; scheduled mapping {1e3 * vo.secs['mapping'] / max(n_map, 1):.1f} ms per window; "
          f"device memory (allocated / reserved MB) at frame 100: {mem[100][0] / 1e6:.0f} / {mem[100][1] / 1e6:.0f}, 300: {mem[300][0] / 1e6:.0f} / {mem[300][1] / 1e6:.0f}, "
          f"{n - 1}: {mem[n - 1][0] / 1e6:.0f} / {mem[n - 1][1] / 1e6:.0f}; the frontend's share (to_keyframe: synthesis and upload of the keyframes, not the chain's) "
          f"{1e3 * vo.frontend_secs / max(len(out['all_kf_ids']) - 1, 1):.1f} ms per keyframe")
    assert len(out["all_kf_ids"]) >= 40 and len(out["kf_ids"]) == 5 and n_map >= 35
    assert rel_rot <= 1e-3 and rel_t <= 2e-3, (rel_rot, rel_t)                      # every frame tracked
    assert rot <= 2e-2 and tt <= 6e-2 and abs(s - 1.0) <= 0.1, (rot, tt, s)         # drift over 640 frames and ~60 keyframe hand-overs: bounded
    # no growth after frame 100: live tensors flat; the caching allocator's pool may still round up a little (its state depends on what ran before)
    assert mem[n - 1][0] <= mem[100][0] * 1.1 + 32e6 and mem[n - 1][1] <= mem[100][1] + 256e6, mem
    # (wall-clock comparisons flake on a loaded host -- the stage times are printed above; what is ASSERTED of "the chain does
    #  not slow down as the sequence grows" is a generous bound: a stage that grew with the sequence would be 3-6 x slower by frame 600)
    assert late["track"] <= 3.0 * early["track"] + 1.0 and late["supp_mapping"] <= 3.0 * early["supp_mapping"] + 1.0


def test_persistent_supplementary_mapping_window_equals_a_rebuilt_one():
    """``loops.GnSuppMapper`` -- ONE window per latest keyframe for the supplementary mapping after every tracked
    frame, the two running supporting frames moved through its slots in place -- against the window rebuilt every frame
    (``persistent_supp=False``): the same chain -- the same keyframe decisions and supporting frames, tracked poses, keyframe poses and
    log-depths equal to the round-off of the source colours -- at a fraction of the per-frame cost."""
    from super_primitive_amd.odometery.sequence import run_sequence
    n = 40
    seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
    run_sequence(frames[:4], to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn")
    outs = {}
    for persistent in (False, True):
        outs[persistent] = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn", translation_thresh=0.095, window_size=5,
                                        depth_of=lambda i: T(seq[i].kld_gt), persistent_supp=persistent)
    a, b = outs[False], outs[True]
    assert a["all_kf_ids"] == b["all_kf_ids"] and a["supp_ids"] == b["supp_ids"] and a["n_supp_mappings"] == b["n_supp_mappings"] == n - 1
    # (not bitwise: the source colours of a window are sampled where the reference samples them -- at the re-projection of the keyframe's
    #  own points, whose last bit depends on the depths, core/dense_optim.py:143-162 -- ONCE, when the window is built: the rebuilt window
    #  samples with this frame's depths, the persistent one with those of the keyframe's second frame.  Round-off of the colours, 1e-7)
    dp = float((a["track_poses"] - b["track_poses"]).abs().max()), float((a["kf_poses"] - b["kf_poses"]).abs().max())
    dk = max(float((x - y).abs().max()) for x, y in zip(a["kf_klds"], b["kf_klds"]))
    assert dp[0] <= 5e-6 and dp[1] <= 5e-6 and dk <= 5e-5, (dp, dk)
    sa, sb = a["seconds"], b["seconds"]
    print(f"\nsupplementary mapping per frame: rebuilt {1e3 * sa['supp_mapping'] / (n - 1):.2f} ms, persistent {1e3 * sb['supp_mapping'] / (n - 1):.2f} ms; "
          f"chain {(n - 1) / sum(sa.values()):.0f} -> {(n - 1) / sum(sb.values()):.0f} frames/s; largest differences: tracked poses {dp[0]:.1e}, keyframe poses {dp[1]:.1e}, "
          f"log-depths {dk:.1e}")
    # (the timing is a printed diagnostic, not an assertion: ADVICE r05; what the persistent window must do is give the same chain)


@pytest.mark.parametrize("cfg", [dict(), dict(affine_compensation=False), dict(continual_steps=0), dict(window_size=3, supp_every_n=2)],
                         ids=["reference extent", "no affine compensation", "no supplementary mapping", "window of 3"])
def test_one_call_per_frame_gives_the_chain_of_the_python_steps(cfg):
    """``sp_chain_step`` (include/sp_hip.h; odometery/chain.py) -- tracking, the supplementary mapping against the two
    running supporting frames and the keyframe criterion of one frame as ONE foreign call with the state on the device -- against the
    step-by-step Python driver (``native_step=False``: ``GnTracker.track`` -> ``GnSuppMapper`` -> ``is_kf``, odometery/odometery.py:1018-1075):
    the same launches in the same order on the same data, so the same chain -- keyframe decisions, supporting frames, iteration for
    iteration -- with poses and depths equal to the round-off of inv(T) T' (one kernel here, three torch launches there)."""
    from super_primitive_amd.odometery.sequence import run_sequence
    n = 40
    seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
    run_sequence(frames[:4], to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn")
    outs = {}
    for native in (False, True):
        outs[native] = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn", depth_of=lambda i: T(seq[i].kld_gt), native_step=native,
                                    **dict(dict(translation_thresh=0.095, window_size=5), **cfg))
    a, b = outs[False], outs[True]
    assert a["all_kf_ids"] == b["all_kf_ids"] and a["supp_ids"] == b["supp_ids"], (a["all_kf_ids"], b["all_kf_ids"])
    assert a["n_supp_mappings"] == b["n_supp_mappings"] == (0 if cfg.get("continual_steps", 10) == 0 else n - 1) and a["n_mappings"] == b["n_mappings"]
    assert len(a["all_kf_ids"]) >= 3
    dp = float((a["track_poses"] - b["track_poses"]).abs().max()), float((a["kf_poses"] - b["kf_poses"]).abs().max())
    dk = max(float((x - y).abs().max()) for x, y in zip(a["kf_klds"], b["kf_klds"]))
    sa, sb = a["seconds"], b["seconds"]
    print(f"\nchain of {n} frames: python steps {(n - 1) / sum(sa.values()):.0f} frames/s, one call per frame {(n - 1) / sum(sb.values()):.0f} frames/s; "
          f"largest differences: tracked poses {dp[0]:.1e}, keyframe poses {dp[1]:.1e}, log-depths {dk:.1e}")
    assert dp[0] <= 5e-6 and dp[1] <= 5e-6 and dk <= 5e-5, (dp, dk)
