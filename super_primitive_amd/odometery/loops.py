"""The two inner-loop shapes of the reference's MonoVO driver (``odometery/odometery.py``) on the HIP cost:
frame-to-keyframe tracking (``:300-312,323-407``) and windowed mapping of one source keyframe against a batch of
targets (``:576-648,687-915``).  Keyframe management, supporting-frame bookkeeping, GUI queues and checkpoints
stay a driver concern and are out of scope; these functions are the part that sits on the hot path
and are what BASELINE config 3 exercises."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..core import dense_optim, dense_optim_batch
from ..lie.lie_algebra import invertSE3, renormalise_se3
from ..lie.lietorch_utils import zero_out_lietorch_tensor
from ..lie.se3 import SE3, LieGroupParameter, se3_exp_matrix

CFG = {'mode': 'colour', 'collect_stats': 0}


def track_frame(kf_precomputed_levels, supp_levels, supp_T, prev_pose, steps, lr=5e-3, prev_aff=None, curr_aff=None):
    """Adam on a zero-reset pose tangent (+ affine) against precomputed source points
    (odometery.py:300-312 setup_tracking_opt, :375-407 the loop).

    kf_precomputed_levels / supp_levels: per pyramid level (coarse -> fine) the ``unproject_kf`` dict and the
    supporting KeyFrame; steps: iterations per level (config ``track.steps``).  Returns (supp_T, curr_aff, losses)."""
    dev = supp_T.device
    delta = LieGroupParameter(SE3.Identity(1, device=dev))
    params = [{'params': [delta], 'lr': lr}]
    affine = prev_aff is not None
    if affine:
        curr_aff = nn.Parameter(curr_aff.detach().clone())
        params.append({'params': [curr_aff], 'lr': 5e-3})
    optim = torch.optim.Adam(params, lr=5e-3)
    supp_T = supp_T.detach().clone()
    losses = []
    for level, n_steps in enumerate(steps):
        for _ in range(n_steps):
            pose = delta.retr().matrix()[0] @ invertSE3(supp_T) @ prev_pose
            aff = (prev_aff, curr_aff) if affine else None
            out = dense_optim.photomeric_cost_precomputed(kf_precomputed_levels[level], supp_levels[level], pose, CFG,
                                                          affine_comp=aff)
            loss = torch.mean(out['residual'])
            losses.append(loss.detach())
            loss.backward()
            optim.step()
            optim.zero_grad()
            with torch.no_grad():
                supp_T = supp_T @ invertSE3(se3_exp_matrix(delta.detach().as_subclass(torch.Tensor))[0])
                zero_out_lietorch_tensor(delta)          # Adam moments persist (odometery.py:400-403)
    supp_T = renormalise_se3(supp_T.contiguous())
    return supp_T, (curr_aff.detach() if affine else None), losses


def map_source_against_targets(src_kf, trg_images, trg_Ks, kld, poses, steps, lr_kld=1e-2, lr_pose=1e-2, lr_aff=1e-5,
                               aff_src=None, affs=None, rel_tol=1e-8):
    """One source keyframe, B targets: Adam on kld + per-target delta poses + affines with fold-in,
    renormalisation and tangent reset every iteration, relative-loss early stop (odometery.py:756-915 restricted
    to a single source keyframe).  Returns (kld, poses (B,4,4), affs, losses)."""
    dev = kld.device
    B = poses.shape[0]
    kld = nn.Parameter(kld.detach().clone())
    deltas = [LieGroupParameter(SE3.Identity(1, device=dev)) for _ in range(B)]
    poses = [p.detach().clone() for p in poses]
    affine = affs is not None
    groups = [{'params': [kld], 'lr': lr_kld}, {'params': deltas, 'lr': lr_pose}]
    if affine:
        affs = [nn.Parameter(a.detach().clone()) for a in affs]
        groups.append({'params': affs, 'lr': lr_aff})
    optim = torch.optim.Adam(groups, lr=1e-3)
    losses = []
    prev = None
    for _ in range(steps):
        P = torch.stack([d.retr().matrix()[0] @ p for d, p in zip(deltas, poses)])
        aff = (aff_src, torch.stack(list(affs))) if affine else None
        out = dense_optim_batch.photomeric_cost_batch(src_kf, trg_images, trg_Ks, kld, P, CFG, affine_comp=aff)
        loss = torch.mean(out['residual'])
        losses.append(loss.detach())
        loss.backward()
        optim.step()
        optim.zero_grad()
        with torch.no_grad():
            for i in range(B):
                step = se3_exp_matrix(deltas[i].detach().as_subclass(torch.Tensor))[0]
                poses[i] = renormalise_se3((step @ poses[i]).contiguous())
                zero_out_lietorch_tensor(deltas[i])
        cur = float(losses[-1])
        if prev is not None and abs(prev - cur) / max(abs(prev), 1e-30) < rel_tol:
            break
        prev = cur
    return kld.detach(), torch.stack(poses), ([a.detach() for a in affs] if affine else None), losses


# ---------------------------------------------------------------------------------------------------------------------
# Fused engines (optim/window.py): same loops, 3 launches per iteration, no autograd graph, no torch optimiser
# ---------------------------------------------------------------------------------------------------------------------
def track_frame_fused(kf, kld, supp_frame, supp_T, prev_pose, steps, levels, lr=5e-3, prev_aff=None, curr_aff=None, polish=()):
    """``track_frame`` on the fused optimiser.  kf: the latest KeyFrame (full resolution), kld: its keypoint log-depths
    (fixed while tracking); supp_frame: the frame being tracked (image + K); steps: iterations per pyramid level, coarse ->
    fine (config ``track.steps``); levels = (pyramid_min, pyramid_max).  ``polish``: optional extra phases
    ((lr scale, iterations), ...) on the finest level, each with a fresh Adam -- not part of the reference's schedule, used to
    obtain a converged result (lr 5e-3 Adam keeps jittering ~1e-3 around the optimum).  Returns (supp_T, curr_aff, losses)."""
    from ..optim.window import KIND_WINDOW, PoseWindow
    affine = prev_aff is not None
    nodes = [dict(T=prev_pose, kind=KIND_WINDOW, aff=prev_aff if affine else None),
             dict(T=supp_T, kind=KIND_WINDOW, lr_pose=lr, lr_aff=5e-3 if affine else 0.0, aff=curr_aff if affine else None,
                  image=supp_frame.image, K=supp_frame.K)]
    win = PoseWindow([dict(kf=kf, kld=kld, lr=0.0, node=0)], nodes, [(0, 1, 1.0, dense_optim.Z_MIN_SINGLE)], levels,
                     abs_loss=False, use_affine=affine, max_iters=max(1, sum(steps) + sum(n for _, n in polish)))
    order = list(reversed(win.level_ids))
    for li, n in enumerate(steps):
        if n > 0:
            win.run(order[li], n)
    scale0 = 1.0
    for scale, n in polish:
        win.reset_optimiser(scale / scale0)
        scale0 = scale
        win.run(order[-1], n)
    T = renormalise_se3(win.node_poses()[1].contiguous())
    return T, (win.node_affines()[1] if affine else None), list(win.losses().unbind(0))


# Gauss-Newton schedules of the window optimiser (optim/window.py run_gn; sp_window_gn_step): per pyramid level, coarse -> fine, LM
# iterations until an accepted step buys less than ``conv_tol`` of the loss, then a polish at the finest level with the IRLS epsilon
# at ``polish_eps`` (the default epsilon smooths |r| like a Huber kernel and leaves the fixed point ~1e-4 from the L1 minimiser).
TRACK_GN_SCHEDULE = dict(phases=((1, 4), (0, 6)), conv_tol=2e-3, irls_eps=1e-3, polish_max=5, polish_eps=1e-5, polish_tol=1e-4)      # phases: (pyramid level, max iterations), coarse -> fine
MAP_GN_SCHEDULE = dict(max_iters=25, conv_tol=1e-3, irls_eps=1e-3, polish_max=12, polish_eps=1e-5, polish_tol=1e-5)


def track_frame_gn(kf, kld, supp_frame, supp_T, prev_pose, levels, prev_aff=None, curr_aff=None, schedule=None):
    """Frame-to-keyframe tracking (odometery/odometery.py:300-312,375-407) by Gauss-Newton / LM instead of 300 Adam steps: 6 pose
    + 2 affine unknowns of the tracked frame against the latest keyframe's points (its depths fixed), coarse to fine over the
    phases of the schedule (pyramid levels inside ``levels`` = (pyramid_min, pyramid_max); at most 15 iterations by default).  Same parameterisation as the reference's loop -- relative pose
    Exp(d) inv(T_supp) T_prev, fold-in T_supp <- T_supp inv(Exp(d)) after every step, renormalise_se3 at the end -- so the result
    is directly comparable with ``track_frame`` / ``track_frame_fused`` run to convergence.
    Returns (supp_T, curr_aff, losses, iterations)."""
    from ..optim.window import KIND_WINDOW, PoseWindow
    sch = dict(TRACK_GN_SCHEDULE, **(schedule or {}))
    affine = prev_aff is not None
    nodes = [dict(T=prev_pose, kind=KIND_WINDOW, aff=prev_aff if affine else None),
             dict(T=supp_T, kind=KIND_WINDOW, lr_pose=1.0, lr_aff=1.0 if affine else 0.0, aff=curr_aff if affine else None,
                  image=supp_frame.image, K=supp_frame.K)]
    phases = [(int(l), int(n)) for l, n in sch['phases'] if levels[0] <= int(l) < levels[1]]
    win = PoseWindow([dict(kf=kf, kld=kld, lr=0.0, node=0)], nodes, [(0, 1, 1.0, dense_optim.Z_MIN_SINGLE)], levels,
                     abs_loss=False, use_affine=affine, max_iters=sum(n for _, n in phases) + sch['polish_max'] + 8)
    its = 0
    for level, n in phases:
        its += win.run_gn(level, n, irls_eps=sch['irls_eps'], conv_tol=sch['conv_tol'])
    if sch['polish_max'] > 0:
        its += win.run_gn(win.level_ids[0], sch['polish_max'], irls_eps=sch['polish_eps'], conv_tol=sch['polish_tol'])
    T = renormalise_se3(win.node_poses()[1].contiguous())
    return T, (win.node_affines()[1] if affine else None), list(win.gn_losses().unbind(0)), its


class GnTracker:
    """Frame-to-keyframe tracking by Gauss-Newton with ONE window per keyframe: the keyframe's padded tables, per-level source
    samples, work list and descriptors are built once (they are a property of the keyframe and its depths -- the reference also
    prepares ``kf_precomputed`` per level before the loop, ``odometery/odometery.py:365-369``); every tracked frame only replaces
    the target image's pyramid in place, the two poses and the LM state.  ``track`` = ``track_frame_gn`` (same schedule, same
    result), at a fraction of the per-frame cost (tools/run_configs.py)."""

    def __init__(self, kf, kld, kf_pose, frame, levels, kf_aff=None, schedule=None):
        from ..optim.window import KIND_WINDOW, PoseWindow
        self.sch = dict(TRACK_GN_SCHEDULE, **(schedule or {}))
        self.affine = kf_aff is not None
        dev = kf.image.device
        z2 = torch.zeros(2, device=dev)
        nodes = [dict(T=kf_pose, kind=KIND_WINDOW, aff=kf_aff if self.affine else None),
                 dict(T=kf_pose, kind=KIND_WINDOW, lr_pose=1.0, lr_aff=1.0 if self.affine else 0.0, aff=z2 if self.affine else None,
                      image=frame.image, K=frame.K)]
        self.phases = [(int(l), int(n)) for l, n in self.sch['phases'] if levels[0] <= int(l) < levels[1]]
        self.win = PoseWindow([dict(kf=kf, kld=kld, lr=0.0, node=0)], nodes, [(0, 1, 1.0, dense_optim.Z_MIN_SINGLE)], levels, abs_loss=False,
                              use_affine=self.affine, max_iters=sum(n for _, n in self.phases) + self.sch['polish_max'] + 8, share_sources=True)
        self._first_image = frame.image

    def update_keyframe(self, kld=None, kf_pose=None, kf_aff=None):
        """After a mapping pass moved the keyframe: new depths / pose / affine pair (the tables do not depend on them)."""
        if kld is not None:
            self.win.set_klds([kld])
        if kf_pose is not None:
            self.win.set_nodes({0: dict(T=kf_pose, aff=kf_aff)})

    def track(self, frame, supp_T, curr_aff=None):
        """Returns (supp_T, curr_aff, losses, iterations) like ``track_frame_gn``."""
        win, sch = self.win, self.sch
        if frame.image is not self._first_image:
            win.set_target_image(1, frame.image)
        self._first_image = None
        win.set_nodes({1: dict(T=supp_T, aff=curr_aff if self.affine else None)})
        win.reset_gn()
        its = 0
        for level, n in self.phases:
            its += win.run_gn(level, n, irls_eps=sch['irls_eps'], conv_tol=sch['conv_tol'])
        if sch['polish_max'] > 0:
            its += win.run_gn(win.level_ids[0], sch['polish_max'], irls_eps=sch['polish_eps'], conv_tol=sch['polish_tol'])
        T = renormalise_se3(win.node_poses()[1].contiguous())
        return T, (win.node_affines()[1] if self.affine else None), list(win.gn_losses().unbind(0)), its


def window_connectivity(n_kfs, mode='map'):
    """Neighbouring keyframes only (odometery.py:451-479); in 'supp' mode only the latest keyframe is a source (:467-469)."""
    return {s: [t for t in (s - 1, s + 1) if 0 <= t < n_kfs] for s in range(n_kfs) if not (mode == 'supp' and s != n_kfs - 1)}


def _window_targets(s, n_kfs, supp):
    """Targets of source keyframe s, in the reference's order (odometery.py:770-823): its neighbouring keyframes, then
    its own supporting frames, then those of the previous keyframe.  Entries: ('kf', t) or ('supp', k, j)."""
    out = [('kf', t) for t in (s - 1, s + 1) if 0 <= t < n_kfs]
    for ss in ([s] + ([s - 1] if s > 0 else [])):
        out += [('supp', ss, j) for j in range(len(supp[ss]))]
    return out


def _free_parts(K, mode, frozen0):
    """Which parameters the reference's optimiser holds (odometery.py:576-648): (pose of keyframe k free, depths of keyframe k free,
    supporting frames free).  'supp': only the latest keyframe's depths (:616-619,634-635 and no pose / affine groups, :586-588,
    :628-629,:544)."""
    if mode == 'supp':
        return [False] * K, [k == K - 1 for k in range(K)], False
    return [k > 0 for k in range(K)], [not (k == 0 and frozen0) for k in range(K)], True


def map_window(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose=1e-4, window_size=5, initialised=True,
               fused=True, rel_tol=1e-8, optimiser="adam", gn_schedule=None, mode='map'):
    """Windowed mapping over several source keyframes (odometery/odometery.py:576-648 parameter groups, :756-915 loop,
    ``opt_supporting`` on).

    kfs: KeyFrames, oldest first; kf_poses: their camera-to-world (4,4); kf_klds: (N_k,) log-depths; kf_affs: (2,) affine
    pairs or None (no affine compensation); supp[k]: supporting frames attached to keyframe k as (KeyFrame, pose, affine).
    Semantics kept: relative pose ``D_trg inv(T_trg) T_src inv(D_src)`` (:793,817); the first keyframe's pose and affine are
    fixed (:589-592,625); the oldest keyframe's depths are frozen once the window is full (:594-603); Adam groups -- log-depths
    1e-2, poses ``lr_pose`` (1e-4, or 1e-2 at mono-init), affines 1e-5; loss = sum_src mean_targets(residual) (:845-850);
    every iteration every pose is folded in ``T <- T inv(Exp(D))``, renormalised and its tangent zeroed (:861-882); when
    ``initialised`` the loop stops once the relative loss change is < ``rel_tol`` (:907-915).
    ``mode``: 'map' / 'init' as above; 'supp' = the supplementary mapping after every tracked frame (:1038-1042): only the latest
    keyframe is a source (:467-469) and only ITS log-depths are optimised (:616-619) -- no pose, no affine pair moves.
    ``optimiser='gn'``: the same window -- same unknowns, same fixed / frozen parts, same fold-in -- optimised by Gauss-Newton / LM
    (``sp_window_gn_step``; MAP_GN_SCHEDULE, ``num_iters`` caps the main phase) instead of Adam.
    Returns dict(kf_poses (K,4,4), klds [K], affs (K,2)|None, supp_poses [[...]], supp_affs [[...]], losses, stopped)."""
    assert mode in ('map', 'init', 'supp')
    K = len(kfs)
    affine = kf_affs is not None
    frozen0 = K == window_size
    if optimiser == "gn":
        return _map_window_fused(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode,
                                 gn=dict(MAP_GN_SCHEDULE, **(gn_schedule or {})))
    if fused:
        return _map_window_fused(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode)
    return _map_window_eager(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode)


class GnSuppMapper:
    """The SUPPLEMENTARY mapping after every tracked frame (``mapping(mode='supp')``, odometery/odometery.py:1038-1042: only the latest
    keyframe is a source, only ITS log-depths move) by Gauss-Newton with ONE window per latest keyframe.  Between two keyframes the
    window of that mapping is the same graph every frame -- the latest keyframe against its predecessor, their stored supporting frames
    and the TWO RUNNING supporting frames (the last two tracked frames, ``collect_tracking_frames(last=True)``) -- and only the two
    running frames change: their images move through two slots of the window in place (the one that stays is copied from slot to slot,
    the new one is packed), their poses and affine pairs are overwritten, the LM state is reset.  Tables, source samples, work list,
    descriptors and packed targets of everything else are built once per keyframe instead of once per frame (1 ms of interpreter time
    per frame).  Same launches and arithmetic as ``map_window(..., mode='supp', optimiser='gn')`` on the same
    inputs; the results agree to the round-off of the source colours (sampled once per window, at the re-projection of the keyframe's
    points under the depths the window was built with; tests/test_gpu_sequence.py)."""

    def __init__(self, kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, window_size=5, gn_schedule=None):
        """``supp[-1]``: the latest keyframe's running supporting frames -- two, or ONE (the first frame after a keyframe or after a scheduled
        mapping has emptied the pool, odometery.py:1046-1055): the window is built with two slots either way; with one running frame both
        slots hold it and the first slot's edge carries weight ZERO, the others 1 / (targets - 1) -- the sums of the one-frame window
        with an exact zero added (``edges_one``)."""
        assert len(supp[-1]) in (1, 2), "built once the latest keyframe has a running supporting frame"
        self.K = K = len(kfs)
        self.affine = kf_affs is not None
        self.gn = dict(MAP_GN_SCHEDULE, **(gn_schedule or {}))
        self.num_iters = int(num_iters)
        rows = list(supp[:-1]) + [list(supp[-1]) * (3 - len(supp[-1]))]
        self.win, self.supp_node, src_ids = _build_map_window(kfs, kf_poses, kf_klds, kf_affs, rows, num_iters, 1e-4, K == window_size, self.affine, True,
                                                              1e-8, 'supp', self.gn)
        assert src_ids == [K - 1]
        self.slots = [self.supp_node[(K - 1, 0)], self.supp_node[(K - 1, 1)]]
        self.frames = [rows[-1][0][0], rows[-1][1][0]]               # the frame objects whose images sit in the slots
        # the edge records for ONE running frame: same graph, the first slot's edge switched off
        win = self.win
        E = win.n_edges
        w = win.edges.clone().view(torch.float32).reshape(E, 4)
        idle = [e for e, (_, node, _, _) in enumerate(win.edge_list) if node == self.slots[0]]
        assert len(idle) == 1 and E >= 2
        w[:, 3] = 1.0 / (E - 1)
        w[idle[0], 3] = 0.0
        self.edges_two, self.edges_one = win.edges, w.reshape(-1).view(torch.uint8)

    def refresh(self, kf_poses, kf_klds, kf_affs, supp):
        """After a mapping that moved keyframes / stored supporting frames (``mapping(mode='map')``): their poses, affine pairs and the
        latest keyframe's depths into the window (the graph -- which frames support which keyframe -- is unchanged until the next keyframe)."""
        up = {k: dict(T=kf_poses[k], aff=kf_affs[k] if self.affine else None) for k in range(self.K)}
        for (k, j), node in self.supp_node.items():
            if k < self.K - 1:
                up[node] = dict(T=supp[k][j][1], aff=supp[k][j][2] if self.affine else None)
        self.win.set_nodes(up)
        self.win.set_klds([kf_klds[-1]])

    def __call__(self, running):
        """running: the latest keyframe's running supporting frames [(frame, pose, aff)], older first (two, or one).  Returns (the latest
        keyframe's new log-depths, losses, iterations)."""
        win, gn = self.win, self.gn
        assert len(running) in (1, 2)
        if len(running) == 1:
            running = [running[0], running[0]]
            win.edges = self.edges_one
        else:
            win.edges = self.edges_two
        if running[0][0] is self.frames[1] and running[1][0] is not self.frames[1]:
            win.copy_target_image(self.slots[0], self.slots[1])      # yesterday's newest frame is today's older one
            self.frames[0] = self.frames[1]
        for j in range(2):
            if running[j][0] is not self.frames[j]:
                win.set_target_image(self.slots[j], running[j][0].image)
                self.frames[j] = running[j][0]
        win.set_nodes({self.slots[j]: dict(T=running[j][1], aff=running[j][2] if self.affine else None) for j in range(2)})
        win.reset_gn()
        n = win.run_gn(0, min(self.num_iters, gn['max_iters']), irls_eps=gn['irls_eps'], conv_tol=gn['conv_tol'])
        if gn['polish_max'] > 0 and self.num_iters > gn['max_iters'] // 2:
            n += win.run_gn(0, gn['polish_max'], irls_eps=gn['polish_eps'], conv_tol=gn['polish_tol'])
        return win.klds()[0], win.gn_losses(), n


def _build_map_window(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode, gn, span_points=None):
    """The PoseWindow of one mapping (nodes = keyframes then supporting frames, sources = the keyframes that are sources in ``mode``,
    one edge per photometric term): (window or None when there is nothing to match, {(k, j): node of supporting frame j of keyframe k},
    source keyframe ids)."""
    from ..optim.window import KIND_WINDOW, PoseWindow
    K = len(kfs)
    free_pose, free_kld, free_supp = _free_parts(K, mode, frozen0)
    lr_aff = 1e-5 if affine else 0.0
    nodes = [dict(T=kf_poses[k], kind=KIND_WINDOW, lr_pose=lr_pose if free_pose[k] else 0.0, lr_aff=lr_aff if free_pose[k] else 0.0,
                  aff=kf_affs[k] if affine else None, renorm=True, image=kfs[k].image, K=kfs[k].K) for k in range(K)]
    supp_node = {}
    for k in range(K):
        for j, (f, pose, aff) in enumerate(supp[k]):
            supp_node[(k, j)] = len(nodes)
            nodes.append(dict(T=pose, kind=KIND_WINDOW, lr_pose=lr_pose if free_supp else 0.0, lr_aff=lr_aff if free_supp else 0.0,
                              aff=aff if affine else None, renorm=True, image=f.image, K=f.K))
    src_ids = sorted(window_connectivity(K, mode))                 # keyframes that are sources; block index = position in this list
    sources = [dict(kf=kfs[k], kld=kf_klds[k], lr=1e-2 if free_kld[k] else 0.0, node=k) for k in src_ids]
    edges = []
    for b, s in enumerate(src_ids):
        trg = _window_targets(s, K, supp)
        for t in trg:
            node = t[1] if t[0] == 'kf' else supp_node[(t[1], t[2])]
            edges.append((b, node, 1.0 / len(trg), dense_optim.Z_MIN_BATCH))
    if not edges:
        return None, supp_node, src_ids
    # (``span_points``: points per workgroup of the window's cost pass.  Default: sized for ONE window's latency -- ~2300 workgroups however
    #  few the points; windows optimised side by side, ``PoseWindowBatch``, want the throughput size, 4096-16384)
    # (the only images such a window replaces in place are those of the latest keyframe's running supporting frames in 'supp' mode --
    #  GnSuppMapper's slots; every other target reads the packed image cached on its frame)
    private = [supp_node[(K - 1, j)] for j in range(len(supp[K - 1]))] if mode == 'supp' else []
    win = PoseWindow(sources, nodes, edges, (0, 1), abs_loss=False, rel_tol=rel_tol if initialised else 0.0, use_affine=affine,
                     max_iters=max(1, num_iters) + (gn['polish_max'] + 8 if gn else 0), span_points=span_points, private_targets=private,
                     share_sources=gn is not None)
    return win, supp_node, src_ids


def _map_window_fused(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode='map', gn=None):
    K = len(kfs)
    win, supp_node, src_ids = _build_map_window(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode, gn)
    det = lambda x: x.detach().clone()
    if win is None:                                                # a single keyframe without supporting frames: nothing to match
        return dict(kf_poses=torch.stack([det(p) for p in kf_poses]), klds=[det(k) for k in kf_klds],
                    affs=torch.stack([det(a) for a in kf_affs]) if affine else None, supp_poses=[[] for _ in range(K)],
                    supp_affs=[[] for _ in range(K)] if affine else None, losses=[torch.zeros((), device=kf_poses[0].device)], stopped=-1)
    if gn:
        n = win.run_gn(0, min(num_iters, gn['max_iters']), irls_eps=gn['irls_eps'], conv_tol=gn['conv_tol'])
        if gn['polish_max'] > 0 and num_iters > gn['max_iters'] // 2:      # (a short budget -- the supplementary mapping -- gets no polish)
            n += win.run_gn(0, gn['polish_max'], irls_eps=gn['polish_eps'], conv_tol=gn['polish_tol'])
        poses, affs, losses = win.node_poses(), win.node_affines(), win.gn_losses()
        stopped, extra = n, dict(gn=win.gn_stats())
        if gn.get('profile', False):                               # (phase time stamps of the last update kernel: a read-back, diagnostics only)
            extra['gn_profile'] = win.gn_profile()
    else:
        win.run(0, num_iters)
        poses, affs, losses = win.node_poses(), win.node_affines(), win.losses()
        stopped, extra = (win.iterations() - 1 if win.converged() else -1), {}
    wk = win.klds()
    klds = [det(k) for k in kf_klds]
    for b, s in enumerate(src_ids):
        klds[s] = wk[b]
    return dict(kf_poses=poses[:K], klds=klds, affs=affs[:K] if affine else None,
                supp_poses=[[poses[supp_node[(k, j)]] for j in range(len(supp[k]))] for k in range(K)],
                supp_affs=[[affs[supp_node[(k, j)]] for j in range(len(supp[k]))] for k in range(K)] if affine else None,
                losses=list(losses.unbind(0)), stopped=stopped, **extra)


def _map_window_eager(kfs, kf_poses, kf_klds, kf_affs, supp, num_iters, lr_pose, frozen0, affine, initialised, rel_tol, mode='map'):
    """The same loop as eager PyTorch around ``photomeric_cost_batch`` (autograd + torch.optim.Adam), statement for
    statement the reference's; kept as the engine-independent check of the fused one."""
    K = len(kfs)
    dev = kf_poses[0].device
    free_pose, free_kld, free_supp = _free_parts(K, mode, frozen0)
    kf_poses = [p.detach().clone() for p in kf_poses]
    d_kf = [LieGroupParameter(SE3.Identity(1, device=dev)) if free_pose[k] else None for k in range(K)]
    klds = [nn.Parameter(k.detach().clone()) if free_kld[i] else k.detach().clone() for i, k in enumerate(kf_klds)]
    affs = [nn.Parameter(a.detach().clone()) if free_pose[i] else a.detach().clone() for i, a in enumerate(kf_affs)] if affine else [None] * K
    s_pose = [[p.detach().clone() for _, p, _ in supp[k]] for k in range(K)]
    s_delta = [[LieGroupParameter(SE3.Identity(1, device=dev)) if free_supp else None for _ in supp[k]] for k in range(K)]
    s_aff = [[nn.Parameter(a.detach().clone()) if free_supp else a.detach().clone() for _, _, a in supp[k]] for k in range(K)] if affine else None
    groups = [{'params': [k for k in klds if isinstance(k, nn.Parameter)], 'lr': 1e-2},
              {'params': [d for d in d_kf if d is not None], 'lr': lr_pose}]
    if affine and mode != 'supp':
        groups.append({'params': [a for a in affs if isinstance(a, nn.Parameter)], 'lr': 1e-5})
    if free_supp:
        groups.append({'params': [d for row in s_delta for d in row], 'lr': lr_pose})
        if affine:
            groups.append({'params': [a for row in s_aff for a in row], 'lr': 1e-5})
    optim = torch.optim.Adam(groups, lr=1e-3)
    eye = torch.eye(4, device=dev)
    mat = lambda d: eye if d is None else d.retr().matrix()[0]
    exp0 = lambda d: eye if d is None else se3_exp_matrix(d.detach().as_subclass(torch.Tensor))[0]
    losses, prev, stopped = [], float('inf'), -1
    for it in range(num_iters):
        res = []
        for s in window_connectivity(K, mode):
            src_delta = mat(d_kf[s])
            imgs, Ks, Ps, As = [], [], [], []
            for t in _window_targets(s, K, supp):
                if t[0] == 'kf':
                    f, T_t, D_t, a_t = kfs[t[1]], kf_poses[t[1]], d_kf[t[1]], affs[t[1]]
                else:
                    f, T_t, D_t = supp[t[1]][t[2]][0], s_pose[t[1]][t[2]], s_delta[t[1]][t[2]]
                    a_t = s_aff[t[1]][t[2]] if affine else None
                imgs.append(f.image); Ks.append(f.K); As.append(a_t)
                Ps.append(mat(D_t) @ invertSE3(T_t) @ kf_poses[s] @ torch.linalg.inv(src_delta))
            if not imgs:
                continue
            out = dense_optim_batch.photomeric_cost_batch(kfs[s], torch.stack(imgs), torch.stack(Ks), klds[s], torch.stack(Ps), CFG,
                                                          affine_comp=(affs[s], torch.stack(As)) if affine else None)
            res.append(out['residual'].mean())
        if not res:
            break
        loss = torch.sum(torch.stack(res))
        losses.append(loss.detach())
        loss.backward()
        optim.step()
        optim.zero_grad()
        with torch.no_grad():
            for i in range(K):
                kf_poses[i] = renormalise_se3((kf_poses[i] @ invertSE3(exp0(d_kf[i]))).contiguous())
                if d_kf[i] is not None:
                    zero_out_lietorch_tensor(d_kf[i])
            for k in range(K):
                for j in range(len(s_pose[k])):
                    s_pose[k][j] = renormalise_se3((s_pose[k][j] @ invertSE3(exp0(s_delta[k][j]))).contiguous())
                    if s_delta[k][j] is not None:
                        zero_out_lietorch_tensor(s_delta[k][j])
        if initialised:
            cur = float(losses[-1])
            if abs(cur - prev) / prev < rel_tol:
                stopped = it
                break
            prev = cur
    det = lambda x: x.detach().clone()
    return dict(kf_poses=torch.stack(kf_poses), klds=[det(k) for k in klds], affs=torch.stack([det(a) for a in affs]) if affine else None,
                supp_poses=s_pose, supp_affs=[[det(a) for a in row] for row in s_aff] if affine else None, losses=losses,
                stopped=stopped)
