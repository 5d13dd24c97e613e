"""BASELINE configs[2] as a SEQUENCE: the chain of the reference's MonoVO driver loop (``odometery/odometery.py:1018-1075``) that
sits on the hot path, with the reference's own bookkeeping of keyframes, supporting frames and the tracked pool:

    track_frame             every frame against the latest keyframe (``:323-449``; no motion prior: ``:324`` switches it off)
    mapping(mode='supp')    ``continual_steps`` iterations after every tracked frame: only the latest keyframe's depths move
                            (``:1038-1042``, parameter groups ``:616-619,634-635``, connectivity ``:467-469``)
    mapping(mode='map')     the scheduled mapping once the new keyframe has two running supporting frames (``:1046-1055``); all
                            keyframe poses but the first, all supporting-frame poses, all affine pairs, all depths but the oldest
                            keyframe's in a full window
    mapping(mode='init')    mono initialisation: two keyframes, unit depths, pose rate 1e-2, no early stop (``:1066-1071,578-581``)
    is_kf                   rendered depth of the latest keyframe -> validity ratio / scaled translation (``:986-1016``)
    init_keyframe           new keyframe's depths from that render by the per-segment median (``:124-196``)
    supporting frames       at most ``supp_every_n - 1`` evenly spaced frames of the tracked pool become the supporting frames of a
                            keyframe when its successor is created (``collect_tracking_frames(last=False)``, ``:1327-1360``); the
                            latest keyframe is supported by the last two tracked frames (``last=True``); ``update_track_pose``
                            (``:969-983``) hands the newest frame's pose back to the tracker -- its TRACKED pose: the write-back after
                            a mapping (``:949-960``) skips the latest keyframe's running supporting frames

on the HIP kernels, with nothing of the reference's frontend, GUI queues, checkpoints or dataset loaders (out of scope, SURVEY.md
section 2).  Every step is one of the drop-in functions the reference's driver calls (``core.depth_render.estimate_depth_kf_native``,
``odometery.kf_criteria``, ``odometery.depth_init.segment_based_depth_reinit``) or one of the inner loops of ``odometery/loops.py``;
``engine`` picks the optimiser of those loops: ``'adam'`` = the reference's schedules on the fused engine (tracking ``[0, 0, 300]``
steps, mapping ``steps`` / ``continual_steps`` / ``init_steps`` iterations), ``'gn'`` = Gauss-Newton / LM (``GnTracker``,
``map_window(optimiser='gn')``).  Without ``mono_init`` the first two keyframes take their depths from given values (the
reference's ground-truth-depth branch, ``:140-163``), which fixes the scale of the trajectory."""
from __future__ import annotations

import time

import torch

from ..core.depth_render import estimate_depth_kf_native
from ..lie.lie_algebra import invertSE3
from .depth_init import segment_based_depth_reinit
from .kf_criteria import keyframe_criterion
from .loops import GnSuppMapper, GnTracker, map_window, track_frame_fused, track_frame_gn  # noqa: F401

def _sync():
    """Wait for THIS thread's stream (the timers' fences): several sequences may run side by side, each on its own stream and host thread
    (tools/chain_throughput.py) -- a device-wide synchronisation would make every one of them wait for all the others."""
    torch.cuda.current_stream().synchronize()


# config/tum/odom_desk.yaml (aligment.track / aligment.mapping / kf / window_size)
DEFAULTS = dict(track_steps=(0, 0, 300), track_levels=(0, 3), track_lr=5e-3, map_steps=500, continual_steps=10, init_steps=1000,
                map_lr_pose=1e-4, window_size=5, supp_every_n=3, depth_validity_ratio=0.60, translation_thresh=0.2,
                affine_compensation=True, mono_init=False, init_frames=7, motion_prior=False, persistent_supp=True, map_rel_tol=1e-8,
                native_step=True)


class _Supp:
    """A supporting frame with its parameters (``SupportingKF`` + ``ParamsSupportingKF``, odometery.py:56-86)."""
    __slots__ = ("frame", "pose", "aff", "ts")

    def __init__(self, frame, pose, aff, ts):
        self.frame, self.pose, self.aff, self.ts = frame, pose, aff, ts


class MonoVO:
    """State and steps of ``Odometery`` (odometery/odometery.py) that belong to the hot path; method names follow the reference."""

    def __init__(self, frames, to_keyframe, pose0, kld0, engine="gn", log=None, depth_of=None, **cfg):
        self.c = dict(DEFAULTS, **cfg)
        self.frames, self.to_keyframe, self.engine, self.log = frames, to_keyframe, engine, log
        self.depth_of = depth_of              # frame index -> keypoint log-depths (the ground-truth-depth branch for the second keyframe)
        self.dev = pose0.device
        self.affine = bool(self.c['affine_compensation'])
        self.kfs, self.kf_ids, self.kf_poses, self.kf_klds, self.kf_affs, self.supp_opt = [], [], [], [], [], []
        self.all_kf_ids = []
        self.reset_tracked_poses()
        self.reset_running_supp_kfs()
        self.initialised = not self.c['mono_init']
        self.mapping_scheduled = False
        self.n_map = dict(supp=0, map=0, init=0)
        self.secs = dict(track=0.0, keyframe=0.0, mapping=0.0, supp_mapping=0.0)
        self.frontend_secs = 0.0              # (spent in ``to_keyframe`` while the chain ran)
        self.tracker = None                   # (Gauss-Newton engine: one window per keyframe, re-used for every frame tracked against it)
        self.supp_mapper = None               # (... and one window per latest keyframe for the supplementary mapping after every frame)
        self.current_aff = torch.zeros(2, device=self.dev)
        # engine 'gn': one foreign call per frame (sp_chain_step; odometery/chain.py) instead of the Python steps below -- same stages, same
        # arithmetic, the state on the device.  ``native_step=False`` keeps the step-by-step path (the two are compared in tests/test_gpu_sequence.py)
        self.chain = None
        self.native = bool(self.c['native_step']) and engine == "gn" and not self.c['motion_prior'] and self.dev.type == "cuda"
        self.add_kf(to_keyframe(0), pose0.clone(), kld0.clone(), 0, self.current_aff.clone())
        self.update_track_pose('init')
        self.track = [pose0.clone()]

    # ---- bookkeeping (odometery.py:1223-1390) ---------------------------------------------------------------------
    def add_kf(self, kf, pose, kld, ts, aff):
        self.kfs.append(kf); self.kf_ids.append(ts); self.kf_poses.append(pose); self.kf_klds.append(kld); self.kf_affs.append(aff)
        self.supp_opt.append([])
        self.all_kf_ids.append(ts)
        self.tracker = None
        self.supp_mapper = None

    def pop_kf(self, i):
        self.supp_mapper = None
        for lst in (self.kfs, self.kf_ids, self.kf_poses, self.kf_klds, self.kf_affs, self.supp_opt):
            lst.pop(i)

    def reset_tracked_poses(self):
        self.tracked = []

    def reset_running_supp_kfs(self):
        self.curr_supp = []

    def collect_tracking_frames(self, last=False):
        """odometery.py:1327-1360: ``last``: the two newest tracked frames; else ``supp_every_n - 1`` evenly spaced ones."""
        n = len(self.tracked)
        if last:
            ids = [n - 1, n - 2]
        else:
            each_n = int(self.c['supp_every_n'])
            ids = [i * (n - 1) // each_n + 1 for i in range(1, each_n)]
        return [_Supp(self.tracked[i].frame, self.tracked[i].pose, self.tracked[i].aff, self.tracked[i].ts) for i in sorted(set(ids)) if 0 <= i < n]

    def tracked_poses_to_supp(self):
        """odometery.py:1271-1289."""
        if not self.initialised:
            self.reset_tracked_poses()
            self.reset_running_supp_kfs()
            return
        self.curr_supp = self.collect_tracking_frames(last=True)

    def flush_tracked_poses_to_supp(self):
        """odometery.py:1314-1325 (called before the new keyframe is added: the pool supports the keyframe it was tracked against)."""
        assert len(self.supp_opt[-1]) == 0
        self.supp_opt[-1] = self.collect_tracking_frames(last=False)

    def update_track_pose(self, mode):
        """odometery.py:969-983."""
        if len(self.curr_supp) == 0 or self.kf_ids[-1] > self.curr_supp[-1].ts:
            assert mode != 'supp'
            self.current_track = self.kf_poses[-1].detach().clone()
            if self.affine:
                self.current_aff = self.kf_affs[-1].detach().clone()
            self.current_ts = self.kf_ids[-1]
        else:
            self.current_track = self.curr_supp[-1].pose.detach().clone()
            if self.affine:
                self.current_aff = self.curr_supp[-1].aff.detach().clone()
            self.current_ts = self.curr_supp[-1].ts

    # ---- tracking (odometery.py:323-449) --------------------------------------------------------------------------
    def track_frame(self, i):
        f, c = self.frames[i], self.c
        supp_T = self.current_track
        if c['motion_prior'] and len(self.tracked) >= 2:            # apply_motion_prior, :314-321 (the reference switches it off, :324)
            supp_T = (self.current_track @ invertSE3(self.tracked[-2].pose)) @ supp_T
        aff_kf = self.kf_affs[-1] if self.affine else None
        _sync(); t0 = time.perf_counter()
        if self.engine == "gn":
            if self.tracker is None:
                self.tracker = GnTracker(self.kfs[-1], self.kf_klds[-1], self.kf_poses[-1], f, c['track_levels'], kf_aff=aff_kf)
            T, aff, _, _ = self.tracker.track(f, supp_T, self.current_aff if self.affine else None)
        else:
            T, aff, _ = track_frame_fused(self.kfs[-1], self.kf_klds[-1], f, supp_T, self.kf_poses[-1], list(c['track_steps']), c['track_levels'],
                                          lr=c['track_lr'], prev_aff=aff_kf, curr_aff=self.current_aff if self.affine else None)
        _sync(); self.secs['track'] += time.perf_counter() - t0
        self.current_track = T.detach().clone()
        if self.affine:
            self.current_aff = aff.detach().clone()
        self.current_ts = i
        self.tracked.append(_Supp(f, self.current_track.clone(), self.current_aff.clone(), i))
        self.track.append(self.current_track.clone())

    # ---- mapping (odometery.py:687-967) ---------------------------------------------------------------------------
    def mapping(self, num_iters, mode='map'):
        assert mode in ('init', 'map', 'supp')
        c = self.c
        if mode == 'init':
            self.reset_running_supp_kfs()
            self.reset_tracked_poses()
        else:
            self.tracked_poses_to_supp()
        K = len(self.kfs)
        # get_supp_kf_poses_pairs (:481-521): no supporting frames at all before the system is initialised
        rows = [(self.curr_supp if k == K - 1 else self.supp_opt[k]) if self.initialised else [] for k in range(K)]
        supp = [[(s.frame, s.pose, s.aff) for s in row] for row in rows]
        lr_pose = 1e-2 if (mode == 'init' and c['mono_init']) else c['map_lr_pose']
        if mode == 'supp' and self.engine == "gn" and c['persistent_supp'] and self.initialised and len(self.curr_supp) in (1, 2):
            # the supplementary mapping between two keyframes is the same window every frame but for the running supporting frames (two; one
            # right after a keyframe or a scheduled mapping): ONE window per latest keyframe, re-pointed in place (loops.GnSuppMapper) --
            # same arithmetic, same result as the branch below
            _sync(); t0 = time.perf_counter()
            if self.supp_mapper is None:
                self.supp_mapper = GnSuppMapper(self.kfs, self.kf_poses, self.kf_klds, self.kf_affs if self.affine else None, supp, num_iters,
                                                window_size=c['window_size'])
            kld, _, _ = self.supp_mapper([(s.frame, s.pose, s.aff) for s in self.curr_supp])
            _sync(); self.secs['supp_mapping'] += time.perf_counter() - t0
            self.kf_klds[-1] = kld
            self.n_map[mode] += 1
            if self.tracker is not None:
                self.tracker.update_keyframe(self.kf_klds[-1])
            self.update_track_pose(mode)
            return
        # (whatever follows moves poses / depths the persistent window holds: a scheduled mapping leaves its GRAPH as it is -- the window is
        #  refreshed with the mapped values below; anything else drops it)
        keep_mapper = self.supp_mapper if mode == 'map' and c['persistent_supp'] else None
        self.supp_mapper = None
        _sync(); t0 = time.perf_counter()
        out = map_window(self.kfs, self.kf_poses, self.kf_klds, self.kf_affs if self.affine else None, supp, num_iters, lr_pose=lr_pose,
                         window_size=c['window_size'], initialised=self.initialised, optimiser="gn" if self.engine == "gn" else "adam", mode=mode,
                         rel_tol=c['map_rel_tol'])
        _sync(); self.secs['supp_mapping' if mode == 'supp' else 'mapping'] += time.perf_counter() - t0
        self.kf_poses = [p.clone() for p in out['kf_poses']]
        self.kf_klds = [k.clone() for k in out['klds']]
        if self.affine:
            self.kf_affs = [a.clone() for a in out['affs']]
        # write-back of the supporting frames, odometery.py:949-960: the loop runs over ``range(len(self.supp_kfs_opt[src_id]))``, and the
        # latest keyframe's entry of that list is empty while it is being mapped (asserted, :484-485) -- so the RUNNING supporting frames of
        # the latest keyframe take part in the optimisation but keep their tracked poses and affine pairs afterwards, and
        # ``update_track_pose`` hands the tracker the un-mapped pose of the newest frame.  Mirrored exactly.
        if self.initialised:
            for k in range(K):
                for j, s in enumerate(self.supp_opt[k]):
                    s.pose = out['supp_poses'][k][j].clone()
                    if self.affine:
                        s.aff = out['supp_affs'][k][j].clone()
        self.n_map[mode] += 1
        if keep_mapper is not None and keep_mapper.K == K:
            keep_mapper.refresh(self.kf_poses, self.kf_klds, self.kf_affs if self.affine else None,
                                [[(s.frame, s.pose, s.aff) for s in self.supp_opt[k]] for k in range(K)])
            self.supp_mapper = keep_mapper
        if self.tracker is not None:                              # the latest keyframe moved
            self.tracker.update_keyframe(self.kf_klds[-1], self.kf_poses[-1], self.kf_affs[-1] if self.affine else None)
        if self.log is not None and mode != 'supp':
            self.log.append((self.current_ts, 'mapping', dict(mode=mode, kf_ids=list(self.kf_ids), klds=[k.clone() for k in self.kf_klds],
                                                               kf_poses=[p.clone() for p in self.kf_poses], n_supp=[len(r) for r in rows],
                                                               losses=[float(out['losses'][0]), float(out['losses'][-1])], n=len(out['losses']),
                                                               gn=out.get('gn'))))
        self.update_track_pose(mode)
        self.initialised = True

    # ---- keyframe decision and creation (odometery.py:986-1016, 124-196) ---------------------------------------------
    def estimate_depth_latest_kf(self, pose):
        return estimate_depth_kf_native(self.kfs[-1], self.kf_klds[-1], invertSE3(pose) @ self.kf_poses[-1])

    def is_kf(self, i):
        c = self.c
        if not self.initialised:
            return i == c['init_frames'], None
        est = self.estimate_depth_latest_kf(self.current_track)
        crit = keyframe_criterion(self.current_track, self.kf_poses[-1], est).tolist()      # [validity ratio, scale, translation diff, rotation deg]
        return (crit[0] < c['depth_validity_ratio'] or crit[2] > c['translation_thresh']), (est, crit)

    def init_keyframe(self, i, info):
        # (``to_keyframe`` is the FRONTEND -- segmentation and per-segment depth shapes, out of the hot path's scope: its time is kept apart
        #  from the chain's 'keyframe' stage, which the caller's timer would otherwise charge it to)
        _sync(); t0 = time.perf_counter()
        kf = self.to_keyframe(i)
        _sync(); dt = time.perf_counter() - t0
        self.frontend_secs += dt
        self.secs['keyframe'] -= dt
        if len(self.kfs) < 2 and self.c['mono_init']:
            kld = torch.zeros(kf.keypoints.shape[0], device=self.dev)                        # log(1), :136-139
            vis, crit, valid = None, None, None
        elif len(self.kfs) < 2 and self.depth_of is not None:
            kld = self.depth_of(i).to(self.dev)                                              # ground-truth depth at the keypoints, :141-163
            vis, crit, valid = None, info[1] if info else None, None
        else:
            est, crit = info if info is not None else (self.estimate_depth_latest_kf(self.current_track), None)
            kld, vis = segment_based_depth_reinit(est.clone(), kf, mode='median', return_info=True)
            valid = float((est > 1e-6).float().mean())
        if self.log is not None:
            self.log.append((i, 'keyframe', dict(criterion=crit, kld=kld.clone(), visible=None if vis is None else int(vis.sum()), valid_ratio=valid)))
        self.add_kf(kf, self.current_track.detach().clone(), kld, i, self.current_aff.detach().clone())
        if self.c['window_size'] is not None and len(self.kfs) > self.c['window_size']:
            self.pop_kf(0)

    # ---- the driver loop (odometery.py:1018-1075) -----------------------------------------------------------------------
    def step(self, i):
        c = self.c
        if self.native and self.initialised:
            return self._step_native(i)
        self.track_frame(i)
        if self.initialised and c['continual_steps'] > 0:
            self.mapping(c['continual_steps'], mode='supp')
        if self.mapping_scheduled and len(self.curr_supp) >= 2:
            self.mapping(c['map_steps'], mode='map')
            self.mapping_scheduled = False
            self.reset_tracked_poses()
            self.reset_running_supp_kfs()
        assert self.current_ts == i
        self.keyframe_stage(i)

    def _step_native(self, i):
        """``step`` with the per-frame stages in ONE foreign call (``sp_chain_step``): tracking, the supplementary mapping against the two
        running supporting frames and the keyframe criterion, all on the device.  What stays in Python is what happens once per keyframe:
        building the windows, the scheduled mapping, the new keyframe."""
        from .chain import CRITERION, SUPP, TRACK, ChainStep
        c, f = self.c, self.frames[i]
        _sync(); t0 = time.perf_counter()
        if self.chain is None:
            H, W = f.image.shape[-2:]
            self.chain = ChainStep(len(self.frames), H, W, c['track_levels'][1], self.dev)
        ch = self.chain
        aff_kf = self.kf_affs[-1] if self.affine else None
        if self.tracker is None:
            self.tracker = GnTracker(self.kfs[-1], self.kf_klds[-1], self.kf_poses[-1], f, c['track_levels'], kf_aff=aff_kf)
        if ch.tracker is not self.tracker:
            ch.bind_tracker(self.tracker, self.kfs[-1], self.affine)
        want_supp = c['continual_steps'] > 0
        native_supp = want_supp and c['persistent_supp']
        images, prev = 0, (self.tracked[-1] if self.tracked else None)      # (no ``prev``: this frame is the keyframe's only running supporting frame)
        if native_supp:
            if self.supp_mapper is None:
                # the window of the supplementary mapping, built with the running frames in its slots (their poses are overwritten by the call)
                K = len(self.kfs)
                rows = [[(s.frame, s.pose, s.aff) for s in self.supp_opt[k]] for k in range(K - 1)]
                rows.append(([(prev.frame, prev.pose, prev.aff)] if prev is not None else []) + [(f, self.current_track, self.current_aff)])
                self.supp_mapper = GnSuppMapper(self.kfs, self.kf_poses, self.kf_klds, self.kf_affs if self.affine else None, rows, c['continual_steps'],
                                                window_size=c['window_size'])
            m = self.supp_mapper
            if ch.mapper is not m:
                ch.bind_mapper(m)
            if prev is None:
                images = 0 if m.frames[1] is f else 2                # (slot 0's edge is switched off: whatever frame sits there)
                m.frames[1] = f
            else:
                if m.frames[0] is prev.frame and m.frames[1] is f:
                    images = 0
                elif m.frames[1] is prev.frame:
                    images = 3                                       # yesterday's newest frame moves to slot 0, this frame into slot 1
                else:
                    m.win.set_target_image(m.slots[0], prev.frame.image)
                    images = 2
                m.frames = [prev.frame, f]
        map_now = self.mapping_scheduled                                  # (mapping(mode='map') may move the keyframe before the criterion looks at it)
        stages = TRACK | (SUPP if native_supp else 0) | (CRITERION if (native_supp or not want_supp) and not map_now else 0)
        _, _, crit = ch.run(stages, i=i, image=f.image, start_pose=self.current_track, start_aff=self.current_aff if self.affine else None,
                            prev=(prev.ts if prev is not None else i) if native_supp else None, supp_images=images, supp_one=prev is None)
        self.secs['track'] += time.perf_counter() - t0
        # ---- the bookkeeping of track_frame (:323-449) ...
        self.current_track = ch.hist_pose[i]
        if self.affine:
            self.current_aff = ch.hist_aff[i]
        self.current_ts = i
        self.tracked.append(_Supp(f, self.current_track, self.current_aff, i))
        self.track.append(self.current_track)
        # ---- ... and of mapping(mode='supp') (:1038-1042)
        if native_supp:
            self.tracked_poses_to_supp()
            self.kf_klds[-1] = self.supp_mapper.win.kld.clone()
            self.n_map['supp'] += 1
            self.update_track_pose('supp')
        elif want_supp:
            self.mapping(c['continual_steps'], mode='supp')
        if self.mapping_scheduled and len(self.curr_supp) >= 2:
            self.mapping(c['map_steps'], mode='map')
            self.mapping_scheduled = False
            self.reset_tracked_poses()
            self.reset_running_supp_kfs()
        assert self.current_ts == i
        t0 = time.perf_counter()
        if crit is None:
            if ch.tracker is not self.tracker:                            # (cannot happen: mapping keeps the tracker; guards the binding)
                ch.bind_tracker(self.tracker, self.kfs[-1], self.affine)
            _, _, crit = ch.run(CRITERION, pose=self.current_track)
        new_kf = crit[0] < c['depth_validity_ratio'] or crit[2] > c['translation_thresh']
        if new_kf:
            self.flush_tracked_poses_to_supp()
            self.init_keyframe(i, (ch.depth, crit))
            self.reset_tracked_poses()
            self.reset_running_supp_kfs()
            _sync()
            self.mapping_scheduled = True
        self.secs['keyframe'] += time.perf_counter() - t0
        return new_kf, crit

    def keyframe_stage(self, i):
        """The tail of one pass of the driver loop (odometery.py:1056-1075): keyframe decision, creation, what it schedules.  Returns
        (new keyframe?, the criterion's values or None)."""
        c = self.c
        _sync(); t0 = time.perf_counter()
        new_kf, info = self.is_kf(i)
        if new_kf:
            self.flush_tracked_poses_to_supp()
            self.init_keyframe(i, info)
            self.reset_tracked_poses()
            self.reset_running_supp_kfs()
        _sync(); self.secs['keyframe'] += time.perf_counter() - t0
        if new_kf:
            if not self.initialised:
                self.mapping(c['init_steps'], mode='init')
            else:
                self.mapping_scheduled = True
        return new_kf, (info[1] if info is not None else None)

    def run(self):
        for i in range(1, len(self.frames)):
            self.step(i)
        return self.result()

    def result(self):
        return dict(track_poses=torch.stack(self.track), kf_ids=list(self.kf_ids), all_kf_ids=list(self.all_kf_ids), kf_poses=torch.stack(self.kf_poses),
                    kf_klds=self.kf_klds, kf_affs=self.kf_affs, supp_ids=[[s.ts for s in row] for row in self.supp_opt],
                    n_mappings=self.n_map['map'], n_supp_mappings=self.n_map['supp'], n_init_mappings=self.n_map['init'], seconds=self.secs, frontend_seconds=self.frontend_secs)


def run_sequence(frames, to_keyframe, pose0, kld0, engine="gn", **cfg):
    """frames: list of supporting-frame-like objects (``image`` (3,H,W), ``K``), frame 0 is the first keyframe;
    to_keyframe(i) -> KeyFrame of frame i (the frontend's job in the reference: segments + per-segment log-depth shapes);
    pose0: camera-to-world of frame 0; kld0: keypoint log-depths of the first keyframe; ``depth_of(i)``: optional keypoint log-depths
    for the second keyframe (the reference's ground-truth-depth initialisation); other keywords override ``DEFAULTS``.
    Returns dict(track_poses (n,4,4) camera-to-world as tracked, kf_ids, kf_poses, kf_klds, n_mappings, seconds dict, ...)."""
    return MonoVO(frames, to_keyframe, pose0, kld0, engine=engine, **cfg).run()
