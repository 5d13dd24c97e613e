"""Pose-parameter helpers with the call surface of the reference's ``lie/lietorch_utils.py``, over the in-repo SE(3)
classes (``lie/se3.py``) instead of the ``lietorch`` extension, which has no ROCm build.

A "pose" here is either an ``SE3`` group element or a ``LieGroupParameter`` (an element plus a zero tangent that an
optimiser moves); ``_group_element`` reduces both to a detached, private ``SE3``."""
import torch

from . import lie_algebra
from .se3 import SE3, LieGroupParameter


def _group_element(pose):
    element = pose.retr() if isinstance(pose, LieGroupParameter) else pose
    return SE3(element.mat.detach().clone())


def lietorch_detach(pose):
    """Independent copy of the pose as a plain group element (no graph, no shared storage)."""
    return _group_element(pose)


def lietorch_new_param(pose):
    """Fresh optimisable parameter anchored at the current value of ``pose``."""
    return LieGroupParameter(_group_element(pose))


def print_pose(pose):
    print(_group_element(pose).matrix())


def zero_out_lietorch_tensor(tensor):
    """Reset a tangent in place (the tracking / mapping loops do this after folding the step into the pose)."""
    with torch.no_grad():
        tensor.data = torch.zeros_like(tensor.data)
    return tensor


def mat_to_lie(mat_pose, device='cuda:0'):
    """4x4 matrix (tensor, or numpy array uploaded to ``device``) -> SE3 via its translation+quaternion vector."""
    if not torch.is_tensor(mat_pose):
        tq = torch.from_numpy(lie_algebra.pose_to_tq(mat_pose[None])).to(device).float()
    else:
        tq = lie_algebra.torch_pose_to_tq(mat_pose[None])
    return SE3.InitFromVec(tq)
