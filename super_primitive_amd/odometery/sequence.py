"""BASELINE configs[2] as a SEQUENCE: the chain of the reference's MonoVO driver loop (``odometery/odometery.py:1018-1075``) that
sits on the hot path -- track every frame against the latest keyframe (``:323-428``), keyframe decision from the rendered depth
of the latest keyframe (``is_kf``, ``:986-1016``; ``odometery/kf_criteria.py``), a new keyframe's depths from that render by the
per-segment median (``init_keyframe``, ``:124-196``: ``estimate_depth_latest_kf`` -> ``segment_based_depth_reinit``), windowed
mapping once the new keyframe has supporting frames (``mapping``, ``:687-937``) -- on the HIP kernels, with nothing of the
reference's frontend, GUI queues, checkpoints or dataset loaders (out of scope).

Every step is one of the drop-in functions the reference's driver calls (``core.depth_render.estimate_depth_kf_native``,
``odometery.kf_criteria``, ``odometery.depth_init.segment_based_depth_reinit``) or one of the two inner loops of
``odometery/loops.py``; ``engine`` picks the optimiser of those loops: ``'adam'`` = the reference's schedule on the fused engine
(tracking ``[0, 0, 300]`` steps, mapping ``steps`` iterations), ``'gn'`` = Gauss-Newton / LM (``track_frame_gn``,
``map_window(optimiser='gn')``).  The first keyframe is initialised from given depths (the reference's ``mono_init: False``
branch, ``:140-163``), which fixes the scale of the trajectory."""
from __future__ import annotations

import time

import torch

from ..core.depth_render import estimate_depth_kf_native
from ..lie.lie_algebra import invertSE3
from .depth_init import segment_based_depth_reinit
from .kf_criteria import keyframe_criterion
from .loops import GnTracker, map_window, track_frame_fused, track_frame_gn  # noqa: F401

DEFAULTS = dict(track_steps=(0, 0, 300), track_levels=(0, 3), track_lr=5e-3, map_steps=500, map_lr_pose=1e-4, window_size=5,
                supp_every_n=3, depth_validity_ratio=0.60, translation_thresh=0.2, affine_compensation=True)


def run_sequence(frames, to_keyframe, pose0, kld0, engine="gn", **cfg):
    """frames: list of supporting-frame-like objects (``image`` (3,H,W), ``K``), frame 0 is the first keyframe;
    to_keyframe(i) -> KeyFrame of frame i (the frontend's job in the reference: segments + per-segment log-depth shapes);
    pose0: camera-to-world of frame 0; kld0: keypoint log-depths of the first keyframe.
    Returns dict(track_poses (n,4,4) camera-to-world as tracked, kf_ids, kf_poses, kf_klds, n_mappings, seconds dict)."""
    log = cfg.pop('log', None)              # optional list: (frame, event, payload) records for diagnostics
    c = dict(DEFAULTS, **cfg)
    dev = pose0.device
    affine = c['affine_compensation']
    zero2 = lambda: torch.zeros(2, device=dev)
    kfs, kf_ids, kf_poses, kf_klds, kf_affs, supp = [to_keyframe(0)], [0], [pose0.clone()], [kld0.clone()], [zero2()], [[]]
    track = [pose0.clone()]
    all_kf_ids = [0]
    cur_T, cur_aff = pose0.clone(), zero2()
    since_kf, scheduled, n_map = 0, False, 0
    tracker = None                       # (Gauss-Newton engine: one window per keyframe, re-used for every frame tracked against it)
    secs = dict(track=0.0, keyframe=0.0, mapping=0.0)
    sync = torch.cuda.synchronize
    for i in range(1, len(frames)):
        f = frames[i]
        # ---- tracking against the latest keyframe; constant-velocity prior (apply_motion_prior, odometery.py:314-321) ----
        init_T = cur_T if len(track) < 2 else (cur_T @ invertSE3(track[-2])) @ cur_T
        sync(); t0 = time.perf_counter()
        if engine == "gn":
            if tracker is None:
                tracker = GnTracker(kfs[-1], kf_klds[-1], kf_poses[-1], f, c['track_levels'], kf_aff=kf_affs[-1] if affine else None)
            cur_T, aff, _, _ = tracker.track(f, init_T, cur_aff if affine else None)
        else:
            cur_T, aff, _ = track_frame_fused(kfs[-1], kf_klds[-1], f, init_T, kf_poses[-1], list(c['track_steps']), c['track_levels'],
                                              lr=c['track_lr'], prev_aff=kf_affs[-1] if affine else None, curr_aff=cur_aff if affine else None)
        sync(); secs['track'] += time.perf_counter() - t0
        if affine:
            cur_aff = aff
        track.append(cur_T.clone())
        since_kf += 1
        if since_kf % c['supp_every_n'] == 0:                 # every n-th tracked frame supports the latest keyframe
            supp[-1].append((f, cur_T.clone(), cur_aff.clone()))
        # ---- scheduled mapping once the new keyframe has two supporting frames (odometery.py:1046-1055) ----
        if scheduled and len(supp[-1]) >= 2:
            sync(); t0 = time.perf_counter()
            out = map_window(kfs, kf_poses, kf_klds, kf_affs if affine else None, supp, c['map_steps'], lr_pose=c['map_lr_pose'],
                             window_size=c['window_size'], initialised=True, optimiser="gn" if engine == "gn" else "adam")
            sync(); secs['mapping'] += time.perf_counter() - t0
            kf_poses = [p.clone() for p in out['kf_poses']]
            kf_klds = [k.clone() for k in out['klds']]
            if affine:
                kf_affs = [a.clone() for a in out['affs']]
            supp = [[(fr, out['supp_poses'][k][j].clone(), (out['supp_affs'][k][j].clone() if affine else a)) for j, (fr, _, a) in enumerate(row)]
                    for k, row in enumerate(supp)]
            scheduled, n_map = False, n_map + 1
            if tracker is not None:          # the latest keyframe moved
                tracker.update_keyframe(kf_klds[-1], kf_poses[-1], kf_affs[-1] if affine else None)
            if log is not None:
                log.append((i, 'mapping', dict(kf_ids=list(kf_ids), klds=[k.clone() for k in kf_klds], kf_poses=[p.clone() for p in kf_poses],
                                               losses=[float(out['losses'][0]), float(out['losses'][-1])], n=len(out['losses']))))
        # ---- keyframe decision on the latest keyframe's depth rendered into the current pose (is_kf) ----
        sync(); t0 = time.perf_counter()
        est_depth = estimate_depth_kf_native(kfs[-1], kf_klds[-1], invertSE3(cur_T) @ kf_poses[-1])
        crit = keyframe_criterion(cur_T, kf_poses[-1], est_depth).tolist()        # [validity ratio, scale, translation diff, rotation deg]
        if crit[0] < c['depth_validity_ratio'] or crit[2] > c['translation_thresh']:
            kf = to_keyframe(i)
            kld, vis = segment_based_depth_reinit(est_depth.clone(), kf, mode='median', return_info=True)
            if log is not None:
                log.append((i, 'keyframe', dict(criterion=crit, kld=kld.clone(), visible=int(vis.sum()), valid_ratio=float((est_depth > 1e-6).float().mean()))))
            kfs.append(kf); kf_ids.append(i); kf_poses.append(cur_T.clone()); kf_klds.append(kld); kf_affs.append(cur_aff.clone()); supp.append([])
            if len(kfs) > c['window_size']:
                for lst in (kfs, kf_ids, kf_poses, kf_klds, kf_affs, supp):
                    lst.pop(0)
            all_kf_ids.append(i)
            since_kf, scheduled, tracker = 0, True, None
        sync(); secs['keyframe'] += time.perf_counter() - t0
    return dict(track_poses=torch.stack(track), kf_ids=kf_ids, all_kf_ids=all_kf_ids, kf_poses=torch.stack(kf_poses), kf_klds=kf_klds, n_mappings=n_map, seconds=secs)
