"""Pose-parameter helpers -- the reference's ``lie/lietorch_utils.py`` API over the in-repo SE(3) (lie/se3.py)."""
import copy

import torch

from . import lie_algebra
from .se3 import SE3, LieGroupParameter


def lietorch_detach(pose):
    if isinstance(pose, LieGroupParameter):
        pose = pose.retr()
    return SE3(pose.mat.detach().clone())


def lietorch_new_param(pose):
    return LieGroupParameter(lietorch_detach(pose))


def print_pose(pose):
    print(lietorch_detach(pose).matrix())


def zero_out_lietorch_tensor(tensor):
    with torch.no_grad():
        tensor.data = torch.zeros_like(tensor.data)
    return tensor


def mat_to_lie(mat_pose, device='cuda:0'):
    if torch.is_tensor(mat_pose):
        return SE3.InitFromVec(lie_algebra.torch_pose_to_tq(mat_pose[None]))
    tq = lie_algebra.pose_to_tq(mat_pose[None])
    return SE3.InitFromVec(torch.from_numpy(tq).to(device).float())
