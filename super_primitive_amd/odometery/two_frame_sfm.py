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

    def instatiate_optimisation(self):
        self.adam_params = [{'params': self.src_depth_keypoints_opt, 'lr': 1e-3},
                            {'params': [pose for _, pose, _ in self.supp_frames if isinstance(pose, LieGroupParameter)], 'lr': 1e-2}]
        self.optim = torch.optim.Adam(self.adam_params, lr=1e-3)

    def run(self):
        al = self.config['aligment']
        src_pyr = keyframe.keyframe_pyramid(self.src_keyframe, al['pyramid_min'], al['pyramid_max'])
        supp_pyrs = [keyframe.keyframe_pyramid(f, al['pyramid_min'], al['pyramid_max']) for f, _, _ in self.supp_frames]
        cost_params = copy.deepcopy(al.get('cost_params', {}))
        cost_params['mode'] = 'colour'
        cost_params['collect_stats'] = self.collect_stats
        count = 0
        for level in range(len(src_pyr)):
            src_l = src_pyr[level]
            supp_l = [p[level] for p in supp_pyrs]
            for _ in range(self.num_iters):
                outs = []
                for fid, frame_l in enumerate(supp_l):
                    _, current_T, pose_to_mat = self.supp_frames[fid]
                    outs.append(dense_optim.photomeric_cost(src_l, frame_l, self.src_depth_keypoints_opt,
                                                            pose=pose_to_mat(current_T), cost_config=cost_params))
                if self.stats_callback is not None:
                    self.stats_callback([dict_cpu(o) for o in outs], level)
                loss = torch.sum(torch.stack([torch.mean(torch.abs(o['residual'])) for o in outs]))
                self.losses.append(loss.detach())
                if count > 0:
                    loss.backward()
                    self.optim.step()
                    self.optim.zero_grad()
                count += 1
        return self

    # -- results -------------------------------------------------------------------------------------
    def poses(self):
        with torch.no_grad():
            return [fn(T).detach().clone() for _, T, fn in self.supp_frames]

    def keypoint_logdepths(self):
        return self.src_depth_keypoints_opt.detach().clone()
