"""SE(3) utilities -- the reference's ``lie/lie_algebra.py`` API.

Quaternion <-> matrix conversions follow the same closed forms (real-part-first internally, ``[t, q_xyzw]`` at
the tq boundary); ``renormalise_se3`` is the per-mapping-iteration re-orthonormalisation and runs as one HIP
launch for cuda inputs (``sp_renormalise_se3``) instead of ~40 tiny ATen ops; ``se3_exp`` / ``batch_se3`` use the
in-repo SE(3) exponential (``lie/se3.py``) in place of lietorch."""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from .. import _lib
from . import se3 as _se3


def quaternion_to_matrix(quaternions):
    """(...,4) real-part-first -> (...,3,3)  (lie_algebra.py:11-38)."""
    w, x, y, z = torch.unbind(quaternions, -1)
    s = 2.0 / (quaternions * quaternions).sum(-1)
    rows = (1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
            s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
            s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y))
    return torch.stack(rows, -1).reshape(quaternions.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    return torch.sqrt(torch.clamp(x, min=0))


def _matrix_to_quaternion_t(matrix):
    """(...,3,3) -> (...,4) real-part-first, choosing the best-conditioned candidate (lie_algebra.py:60-119)."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    lead = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(lead + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack((1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22), dim=-1))
    cands = torch.stack((torch.stack((q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), dim=-1),
                         torch.stack((m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20), dim=-1),
                         torch.stack((m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21), dim=-1),
                         torch.stack((m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2), dim=-1)), dim=-2)
    cands = cands / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(dim=-1)
    return torch.gather(cands, -2, best[..., None, None].expand(lead + (1, 4)))[..., 0, :]


def renormalise_se3(matricies):
    """Re-orthonormalise the rotation block through a quaternion round trip, IN PLACE (lie_algebra.py:41-47)."""
    if matricies.is_cuda and matricies.dtype == torch.float32 and matricies.is_contiguous():
        lib = _lib.load()
        n = matricies.numel() // 16
        _lib.check(lib.sp_renormalise_se3(_lib.ptr(matricies), n, _lib.stream_ptr()), "sp_renormalise_se3")
        return matricies
    matricies[..., :3, :3] = quaternion_to_matrix(_matrix_to_quaternion_t(matricies[..., :3, :3]))
    return matricies


def matrix_to_q_torch(matrix):
    q = _matrix_to_quaternion_t(matrix)
    return torch.cat((q[..., 1:], q[..., :1]), dim=-1)      # xyzw


def torch_pose_to_tq(pose):
    if pose.dim() == 2:
        return torch.cat((pose[:3, 3], matrix_to_q_torch(pose[:3, :3])), dim=0)
    return torch.cat((pose[:, :3, 3], matrix_to_q_torch(pose[:, :3, :3])), dim=1)


def pose_to_tq(pose):
    """numpy (4,4)|(B,4,4) -> (7,)|(B,7) [t, q_xyzw]."""
    if pose.ndim == 2:
        return np.concatenate((pose[:3, 3], Rotation.from_matrix(pose[:3, :3]).as_quat()), axis=0)
    return np.concatenate((pose[:, :3, 3], Rotation.from_matrix(pose[:, :3, :3]).as_quat()), axis=1)


def tq_to_pose(tq):
    single = tq.ndim == 1
    tq2 = tq[None] if single else tq
    T = np.zeros((tq2.shape[0], 4, 4))
    T[:, :3, :3] = Rotation.from_quat(tq2[:, 3:]).as_matrix()
    T[:, :3, 3] = tq2[:, :3]
    T[:, 3, 3] = 1.0
    return T[0] if single else T


def se3_exp(delta):
    """delta (B,6) ordered [omega, v] as in the reference wrapper (lie_algebra.py:177-181) -> (B,4,4)."""
    return _se3.se3_exp_matrix(torch.cat((delta[:, 3:], delta[:, :3]), dim=1))


def batch_se3(poses, delta_T):
    return torch.matmul(poses, se3_exp(delta_T))


def invertSE3(T):
    out = torch.empty_like(T)
    Rt = torch.transpose(T[..., :3, :3], -2, -1)
    out[..., :3, :3] = Rt
    out[..., :3, 3:4] = -torch.matmul(Rt, T[..., :3, 3:4])
    out[..., 3, :3] = 0.0
    out[..., 3, 3] = 1.0
    return out


def normalizeSE3_inplace(T):
    U, _, Vh = torch.linalg.svd(T[..., :3, :3])
    T[..., :3, :3] = torch.matmul(U, Vh)


def skew_symmetric(P):
    out = torch.zeros(tuple(P.shape) + (3,), device=P.device, dtype=P.dtype)
    out[..., 0, 1], out[..., 0, 2] = -P[..., 2], P[..., 1]
    out[..., 1, 0], out[..., 1, 2] = P[..., 2], -P[..., 0]
    out[..., 2, 0], out[..., 2, 1] = -P[..., 1], P[..., 0]
    return out


def SO3_expmap(w):
    """Rodrigues formula.  (The reference's version builds a malformed tensor literal and raises,
    lie_algebra.py:205-221; it has no caller.  This one works.)"""
    theta = torch.linalg.norm(w)
    if float(theta) < 1e-12:
        return torch.eye(3, device=w.device, dtype=w.dtype) + skew_symmetric(w)
    Kx = skew_symmetric(w) / theta
    return torch.eye(3, device=w.device, dtype=w.dtype) + torch.sin(theta) * Kx + (1 - torch.cos(theta)) * (Kx @ Kx)


def SO3_logmap(R, eps=1e-6):
    trace = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    d = trace - 3.0
    theta = torch.acos(0.5 * (trace - 1))
    mag = torch.where(d < -eps, theta / (2.0 * torch.sin(theta)), 0.5 - d / 12.0 + d * d / 60.0)
    v = torch.stack((R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]), dim=1)
    # NOTE: like upstream this broadcasts (B,) against (B,3) and is only meaningful for B == 1
    return mag * v


def SE3_logmap(T, eps=1e-6):
    """Bit-for-bit the reference's formula (lie_algebra.py:247-258), including its element-wise
    ``(0.5 * t) * (w x t)`` term -- kept so results match the reference on the same inputs."""
    w = SO3_logmap(T[:, :3, :3])
    theta = torch.clamp(torch.linalg.norm(w, dim=1), min=eps)
    wn = w / theta
    t = T[:, :3, 3]
    c = torch.linalg.cross(wn, t)
    V_inv_t = t - (0.5 * t) * c + (1.0 - theta / (2.0 * torch.tan(0.5 * theta))) * torch.linalg.cross(wn, c)
    return torch.cat((w, V_inv_t), dim=-1)
