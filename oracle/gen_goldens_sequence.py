"""Golden g21: the reference's MonoVO CHAIN on a short synthetic sequence, every numerical step done by the REAL reference functions.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference; minutes of CPU):

    python oracle/gen_goldens_sequence.py [n_frames]

``odometery/odometery.py`` itself cannot be imported here (it pulls the SAM / cupy frontend, lietorch and hard-codes 'cuda:0', SURVEY.md
section 8(c)), so its driver loop ``Odometery.run`` (:1018-1075) and the bookkeeping it calls are restated below, statement for statement,
around the imported reference functions:

    track_frame (:323-449)           image.keyframe.keyframe_pyramid, core.dense_optim.unproject_kf / photomeric_cost_precomputed,
                                     lie.lie_algebra.invertSE3 / renormalise_se3, torch.optim.Adam (setup_tracking_opt :300-312)
    mapping (:687-967)               gen_goldens.reference_mapping_loop = :576-648 + :756-915 around core.dense_optim_batch.photomeric_cost_batch;
                                     modes 'map' and 'supp'
    is_kf (:986-1016)                core.depth_render.estimate_depth_kf_native, odometery.kf_criteria.translation_difference
    init_keyframe (:124-196)         odometery.depth_init.segment_based_depth_reinit (median) on that render; the first two keyframes take
                                     the ground-truth depth at their keypoints (the ``mono_init: False`` branch, :140-163)
    supporting frames                collect_tracking_frames (:1327-1360), tracked_poses_to_supp (:1271-1289),
                                     flush_tracked_poses_to_supp (:1314-1325), update_track_pose (:969-983)

lietorch's Exp is the oracle's (parity unpinned at that boundary).  Config = config/tum/odom_desk.yaml (track steps [0, 0, 300] at lr 5e-3,
mapping steps 500 / continual_steps 10, supp_every_n 3, window_size 5, affine compensation, depth_validity_ratio 0.6) except
``translation_thresh`` (0.095: the synthetic camera moves 0.03 per frame in front of a plane at depth 3, so keyframes come every ~9 frames).
The frames are ``tests/test_gpu_sequence.py::make_sequence_inputs`` (regenerated there from the recorded arguments).

Round 6, TEACHER FORCING: the golden also holds the chain's COMPLETE state before and after every stage of every
frame (``f<i>_s<k>_*``: s0 = before tracking, s1 = after tracking, s2 = after the supplementary mapping, s3 = after the scheduled
mapping when there was one, s4 = after the keyframe decision / creation; ``snapshot`` below) so that a test can restart ANY stage from
the reference's own recorded input and compare that stage's output at the north-star bar -- a free-running comparison of 24 frames
inherits the jitter of every un-converged Adam stage before it (300 steps at lr 5e-3), a single stage does not.
Two more things are recorded for that test.  ``f<i>_track50_pose / _aff``: the tracker's state after the first 50 of its 300 steps -- a
horizon over which two faithful runs have not yet decorrelated (the reference's tracking does not converge: lr 5e-3 Adam on an L1 cost
jitters by ~3e-4 rad, and 300 steps amplify any change of summation order to that amplitude).  ``spread_*``: THE REFERENCE AGAINST
ITSELF, stage by stage -- after the chain, every stage is run once more from its own recorded input with a different number of threads
(another reduction order) and the largest deviation from the recorded output is kept per stage kind: the yardstick for what "the same
stage output" can mean over a full stage.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_goldens import OUT, T, import_reference, reference_mapping_loop  # noqa: E402
from super_primitive_amd import synth  # noqa: E402
from oracle import photometric_oracle as orc  # noqa: E402

CFG = {"mode": "colour", "collect_stats": 0}
C = dict(track_steps=(0, 0, 300), track_lr=5e-3, map_steps=500, continual_steps=10, map_lr_pose=1e-4, window_size=5, supp_every_n=3,
         depth_validity_ratio=0.60, translation_thresh=0.095)
SEQ = dict(H=224, W=288, N=40, seed=31)


def sequence_twists(n, seed):
    """The trajectory of tests/test_gpu_sequence.py::make_sequence_inputs."""
    rng = np.random.default_rng(seed)
    base = 0.6 * np.array([0.05, -0.02, 0.015, 0.01, -0.015, 0.008])
    return [k * base + 0.003 * rng.standard_normal(6) * (k > 0) for k in range(n)]


class Supp:
    def __init__(self, frame, pose, aff, ts):
        self.frame, self.pose, self.aff, self.ts = frame, pose, aff, ts


class Chain:
    def __init__(self, ref, seq):
        self.ref, self.seq = ref, seq
        self.frames = [ref.kf.KeyFrame(T(f.image), T(f.K)) for f in seq]
        self.kfs, self.kf_ids, self.kf_poses, self.kf_klds, self.kf_affs, self.supp_opt = [], [], [], [], [], []
        self.tracked, self.curr_supp = [], []
        self.initialised, self.mapping_scheduled = True, False              # mono_init False: initialised with the first keyframe (:163)
        self.current_aff = torch.zeros(2)
        self.log = []
        self.add_kf(0, T(seq[0].T_wc), T(seq[0].kld_gt), self.current_aff.clone())
        self.update_track_pose("init")

    def to_kf(self, i):
        f = self.seq[i]
        return self.ref.kf.KeyFrame(T(f.image), T(f.K), T(f.logdepth_perseg), T(f.keypoints), torch.from_numpy(f.keypoint_regions.copy()))

    def add_kf(self, i, pose, kld, aff):
        self.kfs.append(self.to_kf(i)); self.kf_ids.append(i); self.kf_poses.append(pose); self.kf_klds.append(kld); self.kf_affs.append(aff)
        self.supp_opt.append([])

    def collect_tracking_frames(self, last=False):                          # :1327-1360
        n = len(self.tracked)
        ids = [n - 1, n - 2] if last else [i * (n - 1) // C["supp_every_n"] + 1 for i in range(1, C["supp_every_n"])]
        return [Supp(self.tracked[i].frame, self.tracked[i].pose, self.tracked[i].aff, self.tracked[i].ts) for i in sorted(set(ids)) if 0 <= i < n]

    def update_track_pose(self, mode):                                      # :969-983
        if len(self.curr_supp) == 0 or self.kf_ids[-1] > self.curr_supp[-1].ts:
            assert mode != "supp"
            self.current_track, self.current_aff, self.current_ts = self.kf_poses[-1].clone(), self.kf_affs[-1].clone(), self.kf_ids[-1]
        else:
            s = self.curr_supp[-1]
            self.current_track, self.current_aff, self.current_ts = s.pose.clone(), s.aff.clone(), s.ts

    def track_frame(self, i):                                               # :323-449 (motion prior off, :324)
        ref = self.ref
        supp_kf, supp_T = self.frames[i], self.current_track.clone()
        prev_kf, prev_pose, prev_kld, prev_aff = self.kfs[-1], self.kf_poses[-1].clone(), self.kf_klds[-1].clone(), self.kf_affs[-1].clone()
        delta = torch.nn.Parameter(torch.zeros(1, 6))
        aff = torch.nn.Parameter(self.current_aff.clone())
        opt = torch.optim.Adam([{"params": [delta], "lr": C["track_lr"]}, {"params": [aff], "lr": 5e-3}], lr=5e-3)
        with torch.no_grad():
            supp_pyr = ref.kf.keyframe_pyramid(supp_kf, 0, 3, geo_down=False)
            prev_pyr = ref.kf.keyframe_pyramid(prev_kf, 0, 3, geo_down=False)
            pre = [ref.do.unproject_kf(k, prev_kld) for k in prev_pyr]
        loss = None
        self.track50 = None
        for level, n in enumerate(C["track_steps"]):
            for it in range(n):
                if level == len(C["track_steps"]) - 1 and it == 50:
                    self.track50 = (ref.la.renormalise_se3(supp_T).detach().clone(), aff.detach().clone())
                pose = orc.se3_exp(delta)[0] @ torch.linalg.inv(supp_T) @ prev_pose
                out = ref.do.photomeric_cost_precomputed(pre[level], supp_pyr[level], pose=pose, affine_comp=(prev_aff, aff), cost_config=CFG)
                loss = torch.mean(out["residual"])
                loss.backward()
                opt.step()
                opt.zero_grad(set_to_none=True)
                with torch.no_grad():
                    supp_T = supp_T @ torch.linalg.inv(orc.se3_exp(delta.detach())[0])
                    delta.data = torch.zeros_like(delta.data)
        supp_T = ref.la.renormalise_se3(supp_T).detach().clone()
        self.current_track, self.current_aff, self.current_ts = supp_T.clone(), aff.detach().clone(), i
        self.tracked.append(Supp(supp_kf, supp_T.clone(), aff.detach().clone(), i))
        return float(loss)

    def mapping(self, num_iters, mode):                                     # :687-967
        if mode == "init":
            self.curr_supp, self.tracked = [], []
        elif not self.initialised:
            self.curr_supp, self.tracked = [], []
        else:
            self.curr_supp = self.collect_tracking_frames(last=True)        # tracked_poses_to_supp, :1271-1289
        K = len(self.kfs)
        rows = [(self.curr_supp if k == K - 1 else self.supp_opt[k]) if self.initialised else [] for k in range(K)]
        supp = [[(s.frame, s.pose, s.aff) for s in row] for row in rows]
        out = reference_mapping_loop(self.ref, self.kfs, self.kf_poses, self.kf_klds, self.kf_affs, supp, num_iters, C["map_lr_pose"], C["window_size"],
                                     True, self.initialised, mode=mode)
        self.kf_poses = [T(p) for p in out["kf_poses"]]
        self.kf_klds = [T(np.asarray(k, dtype=np.float32)) for k in out["klds"]]
        self.kf_affs = [T(a) for a in out["affs"]]
        # :949-960 -- ``for indx in range(len(self.supp_kfs_opt[src_id]))``: the latest keyframe's list is empty while it is mapped (:484-485),
        # so its running supporting frames are optimised but NOT written back (curr_supp keeps the tracked poses)
        q = 0
        for k, row in enumerate(rows):
            for j, s in enumerate(row):
                if self.initialised and j < len(self.supp_opt[k]):
                    s.pose, s.aff = T(out["supp_poses"][q]), T(out["supp_affs"][q])
                q += 1
        self.update_track_pose(mode)
        self.initialised = True
        return out

    def is_kf(self):                                                        # :986-1016
        ref = self.ref
        pose = self.current_track.clone()
        est = ref.dr.estimate_depth_kf_native(self.kfs[-1], self.kf_klds[-1], torch.linalg.inv(pose) @ self.kf_poses[-1])
        valid = est > 1e-6
        ratio = float(valid.sum() / valid.nelement())
        diff, scale = ref.kc.translation_difference(pose, self.kf_poses[-1], est)
        return (ratio < C["depth_validity_ratio"] or float(diff) > C["translation_thresh"]), est, (ratio, float(scale), float(diff))

    def init_keyframe(self, i, est):                                        # :124-196
        kf = self.to_kf(i)
        if len(self.kfs) < 2:
            kld = T(self.seq[i].kld_gt)                                     # log of the ground-truth depth at the keypoints (:141-163)
            vis = np.ones(kld.shape[0], bool)
        else:
            with torch.no_grad():
                kld, vis = self.ref.di.segment_based_depth_reinit(est.clone(), kf, mode="median", return_info=True)
            vis = vis.numpy()
        self.add_kf(i, self.current_track.clone(), kld.clone(), self.current_aff.clone())
        if len(self.kfs) > C["window_size"]:
            for lst in (self.kfs, self.kf_ids, self.kf_poses, self.kf_klds, self.kf_affs, self.supp_opt):
                lst.pop(0)
        return kld, vis

    def snapshot(self, rec, tag):
        """The chain's complete state (everything a stage reads or writes) under ``<tag>_*``: keyframes in the window, their supporting
        frames (odometery.py ``supp_kfs_opt``), the tracked pool, the running supporting frames, the tracker's hand-over."""
        def frames(key, row):
            rec[f"{tag}_{key}_ts"] = np.array([s.ts for s in row], dtype=np.int64)
            rec[f"{tag}_{key}_poses"] = np.stack([s.pose.numpy() for s in row]).astype(np.float32) if row else np.zeros((0, 4, 4), np.float32)
            rec[f"{tag}_{key}_affs"] = np.stack([s.aff.numpy() for s in row]).astype(np.float32) if row else np.zeros((0, 2), np.float32)
        rec[f"{tag}_kf_ids"] = np.array(self.kf_ids, dtype=np.int64)
        rec[f"{tag}_kf_poses"] = np.stack([p.numpy() for p in self.kf_poses]).astype(np.float32)
        rec[f"{tag}_kf_klds"] = np.stack([k.detach().numpy() for k in self.kf_klds]).astype(np.float32)
        rec[f"{tag}_kf_affs"] = np.stack([a.numpy() for a in self.kf_affs]).astype(np.float32)
        rec[f"{tag}_supp_counts"] = np.array([len(row) for row in self.supp_opt], dtype=np.int64)
        frames("supp", [s for row in self.supp_opt for s in row])
        frames("tracked", self.tracked)
        frames("curr", self.curr_supp)
        rec[f"{tag}_current_track"] = self.current_track.numpy().astype(np.float32).copy()
        rec[f"{tag}_current_aff"] = self.current_aff.numpy().astype(np.float32).copy()
        rec[f"{tag}_flags"] = np.array([self.current_ts, int(self.initialised), int(self.mapping_scheduled)], dtype=np.int64)

    def step(self, i, rec):                                                 # the body of Odometery.run, :1027-1075
        t0 = time.time()
        self.snapshot(rec, "s0")
        rec["track_loss"] = self.track_frame(i)
        rec["track50_pose"], rec["track50_aff"] = self.track50[0].numpy().copy(), self.track50[1].numpy().copy()
        self.snapshot(rec, "s1")
        rec["tracked_pose"] = self.current_track.numpy().copy(); rec["tracked_aff"] = self.current_aff.numpy().copy()
        if self.initialised and C["continual_steps"] > 0:
            out = self.mapping(C["continual_steps"], "supp")
            rec["supp_losses"] = out["losses"]
            rec["supp_kld_last"] = self.kf_klds[-1].numpy().copy()
            self.snapshot(rec, "s2")
        if self.mapping_scheduled and len(self.curr_supp) >= 2:
            out = self.mapping(C["map_steps"], "map")
            self.mapping_scheduled = False
            self.tracked, self.curr_supp = [], []
            rec["map"] = dict(kf_ids=np.array(self.kf_ids), kf_poses=out["kf_poses"], klds=np.stack([np.asarray(k, np.float32) for k in out["klds"]]),
                              affs=out["affs"], supp_poses=out["supp_poses"], supp_affs=out["supp_affs"], losses=out["losses"], stopped=out["stopped"])
            self.snapshot(rec, "s3")
        assert self.current_ts == i
        new_kf, est, crit = self.is_kf()
        rec["criterion"] = np.array(crit)
        rec["new_kf"] = bool(new_kf)
        rec["pose_after"] = self.current_track.numpy().copy()
        if new_kf:
            assert len(self.supp_opt[-1]) == 0
            self.supp_opt[-1] = self.collect_tracking_frames(last=False)    # flush_tracked_poses_to_supp, :1314-1325
            rec["flushed_supp_ids"] = np.array([s.ts for s in self.supp_opt[-1]])
            kld, vis = self.init_keyframe(i, est)
            rec["kf_kld"] = kld.numpy().copy(); rec["kf_visible"] = vis
            self.tracked, self.curr_supp = [], []
            self.mapping_scheduled = True
        self.snapshot(rec, "s4")
        print(f"  frame {i}: track loss {rec['track_loss']:.6f}, criterion {crit}, {'NEW KEYFRAME ' if new_kf else ''}{'mapped ' if 'map' in rec else ''}"
              f"keyframes {self.kf_ids} ({time.time() - t0:.0f} s)", flush=True)


def restore(ch, save, tag):
    """The inverse of ``Chain.snapshot``: put the chain into the recorded state ``<tag>``."""
    ids = [int(i) for i in save[f"{tag}_kf_ids"]]
    ch.kf_ids, ch.kfs = ids, [ch.to_kf(i) for i in ids]
    ch.kf_poses = [T(p) for p in save[f"{tag}_kf_poses"]]
    ch.kf_klds = [T(k) for k in save[f"{tag}_kf_klds"]]
    ch.kf_affs = [T(a) for a in save[f"{tag}_kf_affs"]]
    row = lambda key: [Supp(ch.frames[int(ts)], T(p), T(a), int(ts)) for ts, p, a in zip(save[f"{tag}_{key}_ts"], save[f"{tag}_{key}_poses"], save[f"{tag}_{key}_affs"])]
    flat, ch.supp_opt, q = row("supp"), [], 0
    for c in save[f"{tag}_supp_counts"]:
        ch.supp_opt.append(flat[q: q + int(c)]); q += int(c)
    ch.tracked, ch.curr_supp = row("tracked"), row("curr")
    ch.current_track, ch.current_aff = T(save[f"{tag}_current_track"]), T(save[f"{tag}_current_aff"])
    fl = save[f"{tag}_flags"]
    ch.current_ts, ch.initialised, ch.mapping_scheduled = int(fl[0]), bool(fl[1]), bool(fl[2])


def rot_angle(A, B):
    R = A[:3, :3].T.astype(np.float64) @ B[:3, :3].astype(np.float64)
    return float(np.arctan2(0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]), 0.5 * (np.trace(R) - 1)))


def stage_spread(ch, save, n):
    """Every stage once more from its own recorded input (the caller has changed the thread count): the largest deviation from the recorded
    output, per stage kind.  track: (rot, t, affine) full stage and after 50 steps; supp: max |d log-depth| of the latest keyframe; map:
    (rot, t, log-depth, affine) over keyframes and supporting frames."""
    sp = dict(track=np.zeros(3), track50=np.zeros(3), supp=np.zeros(1), map=np.zeros(4))
    pe = lambda A, B: (rot_angle(np.asarray(A), np.asarray(B)), float(np.abs(np.asarray(A, np.float64)[:3, 3] - np.asarray(B, np.float64)[:3, 3]).max()))
    for i in range(1, n):
        restore(ch, save, f"f{i}_s0")
        ch.track_frame(i)
        r, t = pe(ch.current_track.numpy(), save[f"f{i}_s1_current_track"])
        sp["track"] = np.maximum(sp["track"], (r, t, float(np.abs(ch.current_aff.numpy() - save[f"f{i}_s1_current_aff"]).max())))
        r, t = pe(ch.track50[0].numpy(), save[f"f{i}_track50_pose"])
        sp["track50"] = np.maximum(sp["track50"], (r, t, float(np.abs(ch.track50[1].numpy() - save[f"f{i}_track50_aff"]).max())))
        if f"f{i}_s2_kf_ids" in save:
            restore(ch, save, f"f{i}_s1")
            ch.mapping(C["continual_steps"], "supp")
            sp["supp"] = np.maximum(sp["supp"], float(np.abs(ch.kf_klds[-1].numpy() - save[f"f{i}_s2_kf_klds"][-1]).max()))
        if f"f{i}_s3_kf_ids" in save:
            restore(ch, save, f"f{i}_s2")
            ch.mapping(C["map_steps"], "map")
            e = [pe(p.numpy(), q) for p, q in zip(ch.kf_poses, save[f"f{i}_s3_kf_poses"])]
            e += [pe(s_.pose.numpy(), q) for s_, q in zip([s_ for row in ch.supp_opt for s_ in row], save[f"f{i}_s3_supp_poses"])]
            d = max(float(np.abs(k.numpy() - w).max()) for k, w in zip(ch.kf_klds, save[f"f{i}_s3_kf_klds"]))
            fa = max(float(np.abs(a.numpy() - w).max()) for a, w in zip(ch.kf_affs, save[f"f{i}_s3_kf_affs"]))
            sp["map"] = np.maximum(sp["map"], (max(x[0] for x in e), max(x[1] for x in e), d, fa))
        print(f"  spread after frame {i}: track {sp['track']}, first 50 steps {sp['track50']}, supp {sp['supp']}, map {sp['map']}", flush=True)
    return sp


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("SP_GOLDEN_THREADS", "4")))
    ref = import_reference()
    seq = synth.make_sequence(SEQ["H"], SEQ["W"], SEQ["N"], sequence_twists(n, SEQ["seed"]), keyframe_ids=list(range(n)), seed=SEQ["seed"], overlap=1)
    ch = Chain(ref, seq)
    save = dict(n_frames=np.array(n), HWN=np.array([SEQ["H"], SEQ["W"], SEQ["N"]]), seed=np.array(SEQ["seed"]),
                config=np.array(str({k: (list(v) if isinstance(v, tuple) else v) for k, v in C.items()})))
    t0 = time.time()
    for i in range(1, n):
        rec = {}
        ch.step(i, rec)
        for k, v in rec.items():
            if k == "map":
                save.update({f"f{i}_map_{kk}": vv for kk, vv in v.items()})
            else:
                save[f"f{i}_{k}"] = np.asarray(v)
    save.update(final_kf_ids=np.array(ch.kf_ids), final_kf_poses=np.stack([p.numpy() for p in ch.kf_poses]),
                final_kf_klds=np.stack([k.numpy() for k in ch.kf_klds]), final_kf_affs=np.stack([a.numpy() for a in ch.kf_affs]),
                final_supp_ids=np.array([",".join(str(s.ts) for s in row) for row in ch.supp_opt]),
                gt_poses=np.stack([f.T_wc for f in seq]))
    # the reference against itself, stage by stage, under another reduction order
    threads = torch.get_num_threads()
    torch.set_num_threads(1 if threads > 1 else 2)
    sp = stage_spread(ch, save, n)
    torch.set_num_threads(threads)
    save.update(spread_track=sp["track"], spread_track50=sp["track50"], spread_supp=sp["supp"], spread_map=sp["map"],
                spread_threads=np.array([threads, 1 if threads > 1 else 2]))
    np.savez_compressed(os.path.join(OUT, "g21_config3_sequence_chain.npz"), **save)
    print(f"g21_config3_sequence_chain: {n} frames, keyframes {ch.kf_ids}, {time.time() - t0:.0f} s; the reference against itself per stage: {sp}", flush=True)


if __name__ == "__main__":
    main()
