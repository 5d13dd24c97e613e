"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's keyframe criteria (odometery/kf_criteria.py:7-34) and of the validity ratio the
driver computes next to them (odometery/odometery.py:1003-1004).  Pinned by tests/golden/g11_kf_criteria.npz, produced
by the REAL reference module (oracle/gen_goldens.py)."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def translation_difference(pose_src, pose_target, depth):
    """kf_criteria.py:7-21.  torch.median = lower middle element for an even count."""
    valid = depth > 1e-6
    scale = torch.median(depth[valid])
    diff = torch.linalg.norm(pose_src[:3, 3] - pose_target[:3, 3]) / (scale + 1e-6)
    return diff, scale


def rotation_difference(pose_src, pose_target):
    """kf_criteria.py:23-34 (to_np keeps float32: the inverse and the product are float32, scipy then works in float64)."""
    delta = np.linalg.inv(pose_src.numpy()) @ pose_target.numpy()
    return np.linalg.norm(Rotation.from_matrix(delta[:3, :3]).as_rotvec()) * 180.0 / np.pi


def validity_ratio(depth):
    """odometery/odometery.py:1003-1004."""
    valid = depth > 1e-6
    return valid.sum() / valid.nelement()
