"""Two-frame SfM optimiser -- the loop of the reference's ``odometery/two_frame_sfm.py:95-207`` on the HIP cost.

The reference class is an ``mp.Process`` that also loads a dataset, runs the SAM / normals frontend and feeds an
Open3D GUI; those parts are out of scope.  What is kept, with the same names and semantics:

* ``init_supporting_frame`` -- pose parameter = ``LieGroupParameter`` around a given initial pose (the reference
  perturbs the ground truth with ``SE3.Random(sigma=0.05)``, ``:77-84``); ``pose_to_mat = retr().matrix()[0]``;
* ``init_optimisation`` -- log-depths ``log(2 + 2*rand(N))`` as ``nn.Parameter`` (``:103-109``);
* ``instatiate_optimisation`` -- ``Adam([{kld, lr 1e-3}, {poses, lr 1e-2}], lr=1e-3)`` (``:116-123``);
* ``run`` -- coarse-to-fine over ``keyframe_pyramid(pyramid_min, pyramid_max)``, ``num_iters`` per level, per support
  frame ``photomeric_cost`` with ``collect_stats`` as configured, ``loss = sum_f mean|residual|``, **no update on
  the very first iteration** (``count > 0``, ``:203``), optional per-iteration stats callback in place of the GUI
  queue (``:175-183``).

``run()`` has two engines with the same semantics.  The default (no stats consumer attached) is the FUSED one
(``optim/window.py``: cost of every support frame in one launch + ``sp_window_step`` -- Adam with torch's arithmetic on
the log-depths and on the persistent pose tangents, the gradient of ``Exp(a) X`` by dual numbers; 3 launches per
iteration, no autograd graph).  ``fused=False`` -- or a ``stats_callback`` / ``collect_stats > 0`` -- runs the eager loop:
``photomeric_cost`` + autograd + ``torch.optim.Adam``, line for line the reference's.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from ..core import dense_optim
from ..image import keyframe
from ..lie.se3 import SE3, LieGroupParameter
from ..tool.etc import dict_cpu


class SfM:
    def __init__(self, config, src_keyframe, support_frames, init_poses, num_iters=500, opt_pose=True,
                 stats_callback=None, collect_stats=0):
        """config: dict with ``aligment.{pyramid_min,pyramid_max,cost_params}``; support_frames: list of supporting
        KeyFrames (image + K); init_poses: list of (4,4) initial target<-source matrices."""
        self.config = config
        self.src_keyframe = src_keyframe
        self.opt_pose = opt_pose
        self.num_iters = num_iters
        self.stats_callback = stats_callback
        self.collect_stats = collect_stats
        self.supp_frames = [self.init_supporting_frame(f, T) for f, T in zip(support_frames, init_poses)]
        self.losses = []
        self._stepped = False          # the very first iteration of the first run() makes no update (two_frame_sfm.py:203)
        self._window = None
        self._lr_scale = 1.0

    def init_supporting_frame(self, frame, pose_init):
        if self.opt_pose:
            current_T = LieGroupParameter(SE3(pose_init.detach().clone()[None].float()))
            pose_to_mat = lambda x: x.retr().matrix()[0]
        else:
            current_T = pose_init.detach().clone().float()
            pose_to_mat = lambda x: x
        return frame, current_T, pose_to_mat

    def init_optimisation(self, kld_init=None, generator=None):
        N = self.src_keyframe.keypoints.shape[0]
        dev = self.src_keyframe.image.device
        if kld_init is None:
            kld_init = torch.log(2.0 + 2 * torch.rand(N, device=dev, generator=generator))
        self.src_depth_keypoints_opt = nn.Parameter(kld_init.detach().clone().float().to(dev))
        self.instatiate_optimisation()

    def _drop_window(self):
        """The fused engine's window owns copies of the log-depths, tangents and Adam moments: anything that replaces or
        edits the parameters outside ``_run_fused`` invalidates it (the next fused run rebuilds it from the parameters)."""
        self._window = None

    def instatiate_optimisation(self):
        self._drop_window()
        self.adam_params = [{'params': self.src_depth_keypoints_opt, 'lr': 1e-3},
                            {'params': [pose for _, pose, _ in self.supp_frames if isinstance(pose, LieGroupParameter)], 'lr': 1e-2}]
        self.optim = torch.optim.Adam(self.adam_params, lr=1e-3)

    def run(self, fused=None, lr_scale=1.0, levels=None, num_iters=None, graphed=False):
        """The reference loop.  ``fused``: None = fused unless per-iteration statistics are wanted (do not switch engines
        between calls on one object: each keeps its own Adam moments).  ``lr_scale`` scales both learning rates -- a
        change starts a fresh Adam -- and ``levels`` restricts the pyramid levels visited (indices into the
        coarse-to-fine pyramid); both exist for the convergence polish of the parity tests (lr/10, lr/100 on the finest
        level, like oracle/gen_goldens_fullsize.py) and default to the reference's behaviour."""
        if fused is None:
            fused = self.stats_callback is None and self.collect_stats == 0
        if fused:
            return self._run_fused(lr_scale, levels, num_iters)
        return self._run_eager(lr_scale, levels, num_iters, graphed=graphed)

    def _run_fused(self, lr_scale, levels, num_iters):
        from ..optim.window import KIND_DIRECT, PoseWindow
        al = self.config['aligment']
        n_levels = al['pyramid_max'] - al['pyramid_min']
        iters = num_iters or self.num_iters
        win = self._window
        if win is not None and not self._window_in_sync(win):
            win = self._window = None                                # parameters were edited behind the window's back
        if win is None:
            nodes = []
            for frame, current_T, _ in self.supp_frames:
                if isinstance(current_T, LieGroupParameter):
                    nodes.append(dict(T=current_T.group.mat[0], kind=KIND_DIRECT, lr_pose=1e-2 * lr_scale, image=frame.image, K=frame.K))
                else:
                    nodes.append(dict(T=current_T, kind=KIND_DIRECT, lr_pose=0.0, image=frame.image, K=frame.K))
            win = PoseWindow([dict(kf=self.src_keyframe, kld=self.src_depth_keypoints_opt.detach(), lr=1e-3 * lr_scale, node=-1)],
                             nodes, [(0, f, 1.0, dense_optim.Z_MIN_SINGLE) for f in range(len(nodes))],
                             (al['pyramid_min'], al['pyramid_max']), abs_loss=True, skip_first=not self._stepped,
                             max_iters=max(8192, 4 * iters * n_levels))
            tang = [T.detach().as_subclass(torch.Tensor)[0] if isinstance(T, LieGroupParameter) else torch.zeros(6, device=win.device)
                    for _, T, _ in self.supp_frames]
            if any(bool((t != 0).any()) for t in tang):
                win.set_tangents(torch.stack(tang))
            self._window, self._fused_seen = win, 0
        elif lr_scale != self._lr_scale:
            win.reset_optimiser(lr_scale / self._lr_scale)           # a fresh Adam at the new rates
        self._lr_scale = lr_scale
        order = list(reversed(win.level_ids))                       # coarse -> fine, like keyframe_pyramid's list
        for li in (range(n_levels) if levels is None else levels):
            win.run(order[li], iters)
        self._stepped = True
        hist = win.losses()
        self.losses.extend(hist[self._fused_seen:].unbind(0))
        self._fused_seen = int(hist.shape[0])
        with torch.no_grad():
            self.src_depth_keypoints_opt.data.copy_(win.klds()[0])
            tang = win.node_tangents()
            for f, (_, current_T, _) in enumerate(self.supp_frames):
                if isinstance(current_T, LieGroupParameter):
                    current_T.data.copy_(tang[f][None])
        return self

    def _window_in_sync(self, win):
        """True when the window's log-depths and tangents are still the values of the parameters (they are written back at the
        end of every fused run; an in-place edit of a parameter in between shows up here)."""
        with torch.no_grad():
            if not torch.equal(win.klds()[0], self.src_depth_keypoints_opt.detach().to(win.device)):
                return False
            tang = win.node_tangents()
            for f, (_, current_T, _) in enumerate(self.supp_frames):
                if isinstance(current_T, LieGroupParameter) and not torch.equal(tang[f], current_T.detach().as_subclass(torch.Tensor)[0].to(win.device)):
                    return False
        return True

    def _run_eager(self, lr_scale=1.0, levels=None, num_iters=None, graphed=False):
        """``graphed``: the SAME statements -- ``photomeric_cost`` per support frame, ``loss.backward()``, ``torch.optim.Adam.step()``,
        ``zero_grad()`` -- recorded once per pyramid level into a hipGraph (``tool/graph_loop.GraphedStep``) and replayed: the
        interpreter, the autograd engine and the optimiser's Python run at capture time only (470-640 us -> well under 150 us per
        iteration; same arithmetic, Adam's step counter on the device).  Not with a ``stats_callback`` (a per-iteration host copy)."""
        al = self.config['aligment']
        self._drop_window()                                          # this engine moves the parameters: the fused window's copies go stale
        if lr_scale != self._lr_scale:                               # a fresh Adam at the new rates
            self.adam_params = [{'params': self.src_depth_keypoints_opt, 'lr': 1e-3 * lr_scale},
                                {'params': [pose for _, pose, _ in self.supp_frames if isinstance(pose, LieGroupParameter)],
                                 'lr': 1e-2 * lr_scale}]
            self.optim = torch.optim.Adam(self.adam_params, lr=1e-3)
            self._lr_scale = lr_scale
        num_iters = num_iters or self.num_iters
        src_pyr = keyframe.keyframe_pyramid(self.src_keyframe, al['pyramid_min'], al['pyramid_max'])
        supp_pyrs = [keyframe.keyframe_pyramid(f, al['pyramid_min'], al['pyramid_max']) for f, _, _ in self.supp_frames]
        cost_params = copy.deepcopy(al.get('cost_params', {}))
        cost_params['mode'] = 'colour'
        cost_params['collect_stats'] = self.collect_stats
        count = 1 if self._stepped else 0
        for level in (range(len(src_pyr)) if levels is None else levels):
            src_l = src_pyr[level]
            supp_l = [p[level] for p in supp_pyrs]
            def iteration(update=True):
                outs = []
                for fid, frame_l in enumerate(supp_l):
                    _, current_T, pose_to_mat = self.supp_frames[fid]
                    outs.append(dense_optim.photomeric_cost(src_l, frame_l, self.src_depth_keypoints_opt,
                                                            pose=pose_to_mat(current_T), cost_config=cost_params))
                if self.stats_callback is not None:
                    self.stats_callback([dict_cpu(o) for o in outs], level)
                loss = torch.sum(torch.stack([torch.mean(torch.abs(o['residual'])) for o in outs]))
                if update:
                    loss.backward()
                    self.optim.step()
                    self.optim.zero_grad()
                return loss.detach()

            left = num_iters
            if count == 0 and left > 0:                      # the very first iteration makes no update (two_frame_sfm.py:203)
                self.losses.append(iteration(update=False))
                count, left = 1, left - 1
            if graphed and self.stats_callback is None and left > 3:
                from ..tool.graph_loop import GraphedStep
                step = GraphedStep(iteration, [self.optim], warmup=2)
                self.losses.extend(step.warmup_outputs)
                for _ in range(left - 2):
                    self.losses.append(step.replay().clone())
                count += left
                continue
            for _ in range(left):
                self.losses.append(iteration())
                count += 1
        self._stepped = True
        return self

    def run_on_device(self, mode="adam", iters_per_level=None, use_graph=True, **kw):
        """The same coarse-to-fine schedule with the whole loop on the GPU (``optim.PairBatch``: 2 launches per
        iteration, no autograd graph, no torch optimizer) -- ~30x the iteration rate of ``run()``.

        ``mode='adam'``: Adam with the reset-tangent pose parameterisation of the reference's tracking / mapping
        loops (``odometery/odometery.py:394-403``); it differs from ``run()``'s accumulated tangent only in the
        second-order terms of Exp and in not skipping the very first update.  ``mode='gn'``: Gauss-Newton / LM on the
        same L1 cost (IRLS) -- without ``iters_per_level`` the CONVERGING per-pair schedule of ``PairBatch.run_scheduled`` with its
        verdict and further attempts (at 640 x 480 and three levels: ``REFERENCE_START_SCHEDULE``, the schedule ``frame_pairs_per_sec`` is
        quoted on, third attempt = the reference's own Adam), whose outcome is kept in ``device_status`` / ``device_attempts`` /
        ``device_diag`` (``converged()``; a flagged pair raises a ``RuntimeWarning``: a wrong pose never comes back silently -- the
        reference only asserts finiteness, core/dense_optim.py:311,321,340-343).  ``verdict=dict(cost_bound=...)`` hands the run the one
        yardstick a pair ALONE cannot have: what a converged pair's cost is (a tracker knows it from its previous frames).  With
        ``iters_per_level`` a fixed number of LM iterations per level, no verdict.  One supporting frame only (several frames share the
        log-depths and are not independent pairs).  Results are written back so ``poses()`` / ``keypoint_logdepths()`` work as after
        ``run()``."""
        from .. import _lib
        from ..optim.pair_batch import (FRAME_PAIR_SCHEDULE, REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, REFERENCE_START_SCHEDULE,
                                        PairBatch)
        if len(self.supp_frames) != 1:
            raise NotImplementedError("run_on_device handles one supporting frame; use run() for several")
        al = self.config['aligment']
        self._drop_window()
        frame, current_T, pose_to_mat = self.supp_frames[0]
        with torch.no_grad():
            pose0 = pose_to_mat(current_T).detach().clone()
        levels = (al['pyramid_min'], al['pyramid_max'])
        self.device_status = self.device_attempts = self.device_diag = None
        if mode == "gn" and iters_per_level is None:
            H, W = frame.image.shape[-2:]
            if levels == REFERENCE_START_LEVELS and H * W >= 480 * 640 // 2:
                build, sched = dict(point_stride=REFERENCE_START_POINT_STRIDE, granule=64), dict(REFERENCE_START_SCHEDULE)
            else:               # (small frames / other pyramids: every level on all its points, two Gauss-Newton attempts)
                build = dict(tile_points=2048)
                sched = dict(FRAME_PAIR_SCHEDULE, use_coarse=False, pose_first_iters=15, pose_first_eps=1e-2, retry_pose_first=((levels[1] - 1, 30),))
            batch = PairBatch([self.src_keyframe], [frame.image], [frame.K], pose0[None], [self.src_depth_keypoints_opt.detach()], levels=levels, **build)
            batch.run_scheduled(**dict(sched, **kw))
            self.device_status, self.device_attempts = int(batch.status[0]), int(batch.attempts[0]) + 1
            self.device_diag = batch.diag[0].detach().cpu()
            if self.device_status & _lib.SP_STATUS_FAILED:
                import warnings
                warnings.warn(f"SfM.run_on_device: the pair failed its verdict after {self.device_attempts} attempt(s) (status {self.device_status:#x}, "
                              "include/sp_hip.h SP_STATUS_*): the returned pose / depths are the end state of the last attempt", RuntimeWarning)
        else:
            batch = PairBatch([self.src_keyframe], [frame.image], [frame.K], pose0[None], [self.src_depth_keypoints_opt.detach()], levels=levels,
                              tile_points=2048)
            batch.run(iters_per_level or self.num_iters, mode=mode, use_graph=use_graph, **kw)
        self.losses.append(batch.evaluate(al['pyramid_min'])[0].detach())
        with torch.no_grad():
            self.src_depth_keypoints_opt.data.copy_(batch.klds()[0])
            final = batch.poses()[0].clone()
            if self.opt_pose:
                from ..lie.se3 import SE3, LieGroupParameter
                self.supp_frames[0] = (frame, LieGroupParameter(SE3(final[None])), pose_to_mat)
            else:
                self.supp_frames[0] = (frame, final, pose_to_mat)
        return self

    # -- results -------------------------------------------------------------------------------------
    def converged(self):
        """After ``run_on_device(mode='gn')``: True = the pair passed its verdict, False = flagged, None = no verdict was taken."""
        from .. import _lib
        st = getattr(self, "device_status", None)
        return None if st is None else not (st & _lib.SP_STATUS_FAILED)

    def poses(self):
        with torch.no_grad():
            return [fn(T).detach().clone() for _, T, fn in self.supp_frames]

    def keypoint_logdepths(self):
        return self.src_depth_keypoints_opt.detach().clone()
