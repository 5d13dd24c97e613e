"""A small SE(3) group object + tangent parameter with the slice of the lietorch API the reference's drivers use
(``SE3.InitFromVec / Random / Identity / mul / inv / matrix / vec``, ``LieGroupParameter(...).retr()``;
call sites: ``odometery/two_frame_sfm.py:77-84``, ``odometery/odometery.py:224-228,301,551-555,589-592``,
``lie/lietorch_utils.py:6-33``).

lietorch ships CUDA-only kernels, has no ROCm build and is not vendored in the reference tree, so its exact
source could not be read: semantics follow its public documentation -- group data ``[t, q_xyzw]``, tangent
ordered ``[tau (translation), phi (rotation)]``, **left** retraction ``retr(a) = Exp(a) * X``, full SE(3)
exponential ``t = V(phi) tau`` -- and the closed form is checked against ``scipy.linalg.expm`` in the tests.
Parity at this boundary is therefore UNPINNED; the tracking/mapping loops reset the tangent to
zero after every step, which makes their results independent of these conventions.

Everything here is ordinary differentiable torch code on tiny tensors (host-side glue, not the hot path).
"""
from __future__ import annotations

import torch


def _hat(phi):
    z = torch.zeros_like(phi[..., 0])
    return torch.stack((z, -phi[..., 2], phi[..., 1], phi[..., 2], z, -phi[..., 0], -phi[..., 1], phi[..., 0], z),
                       -1).reshape(phi.shape[:-1] + (3, 3))


def se3_exp_matrix(xi):
    """Twist (...,6) = [tau, phi] -> (...,4,4).  Safe to differentiate at xi = 0 (double-where)."""
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    small = th2 < 1e-8
    th = torch.sqrt(torch.where(small, torch.ones_like(th2), th2))
    A = torch.where(small, 1 - th2 / 6, torch.sin(th) / th)[..., None]
    B = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(th)) / (th * th))[..., None]
    C = torch.where(small, 1.0 / 6 - th2 / 120, (th - torch.sin(th)) / (th * th * th))[..., None]
    W = _hat(phi)
    W2 = W @ W
    I = torch.eye(3, dtype=xi.dtype, device=xi.device).expand_as(W)
    R = I + A * W + B * W2
    V = I + B * W + C * W2
    t = (V @ tau[..., None])
    top = torch.cat((R, t), dim=-1)
    bottom = torch.zeros(xi.shape[:-1] + (1, 4), dtype=xi.dtype, device=xi.device)
    bottom[..., 0, 3] = 1
    return torch.cat((top, bottom), dim=-2)


def _quat_xyzw_to_R(q):
    x, y, z, w = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    return torch.stack((1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                        s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                        s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)), -1).reshape(q.shape[:-1] + (3, 3))


class _FusedRetract(torch.autograd.Function):
    """T = Exp(a) * X as one HIP launch forward and one backward (``sp_se3_retract``); cuda fp32 only."""

    @staticmethod
    def forward(ctx, a, X):
        from .. import _lib
        lib = _lib.load()
        a_c, X_c = a.detach().contiguous().float(), X.detach().contiguous().float()
        n = a_c.shape[0]
        T = torch.empty(n, 4, 4, dtype=torch.float32, device=a.device)
        _lib.check(lib.sp_se3_retract(_lib.ptr(a_c), _lib.ptr(X_c), n, _lib.ptr(T), None, None, _lib.stream_ptr()),
                   "sp_se3_retract")
        ctx.save_for_backward(a_c, X_c)
        return T

    @staticmethod
    def backward(ctx, grad_T):
        from .. import _lib
        lib = _lib.load()
        a_c, X_c = ctx.saved_tensors
        g = grad_T.contiguous().float()
        ga = torch.empty_like(a_c)
        _lib.check(lib.sp_se3_retract(_lib.ptr(a_c), _lib.ptr(X_c), a_c.shape[0], None, _lib.ptr(g), _lib.ptr(ga),
                                      _lib.stream_ptr()), "sp_se3_retract(backward)")
        return ga, None


class SE3:
    """Batch of rigid transforms stored as (B,4,4) matrices."""
    tangent_dim = 6

    def __init__(self, mat):
        self.mat = mat

    # -- constructors ------------------------------------------------------------------------------
    @staticmethod
    def InitFromVec(tq):
        tq = tq.reshape(-1, 7)
        M = torch.zeros(tq.shape[0], 4, 4, dtype=tq.dtype, device=tq.device)
        M[:, :3, :3] = _quat_xyzw_to_R(tq[:, 3:])
        M[:, :3, 3] = tq[:, :3]
        M[:, 3, 3] = 1
        return SE3(M)

    @staticmethod
    def Identity(n=1, device=None, dtype=torch.float32):
        return SE3(torch.eye(4, dtype=dtype, device=device).repeat(n, 1, 1))

    @staticmethod
    def Random(n=1, sigma=1.0, device=None, dtype=torch.float32, generator=None):
        xi = sigma * torch.randn(n, 6, dtype=dtype, device=device, generator=generator)
        return SE3.exp(xi)

    @staticmethod
    def exp(xi):
        return SE3(se3_exp_matrix(xi.reshape(-1, 6)))

    # -- algebra -----------------------------------------------------------------------------------
    def mul(self, other):
        return SE3(self.mat @ other.mat)

    __mul__ = mul

    def inv(self):
        R = self.mat[:, :3, :3].transpose(1, 2)
        M = torch.zeros_like(self.mat)
        M[:, :3, :3] = R
        M[:, :3, 3:] = -(R @ self.mat[:, :3, 3:])
        M[:, 3, 3] = 1
        return SE3(M)

    def retr(self, a):
        """Left retraction Exp(a) * X.  On a cuda device this is the fused HIP forward/backward pair; elsewhere
        (CPU tests of the host glue) the differentiable torch expression."""
        a = a.reshape(-1, 6)
        if a.is_cuda and self.mat.dtype == torch.float32:
            return SE3(_FusedRetract.apply(a, self.mat))
        return SE3(se3_exp_matrix(a) @ self.mat)

    def matrix(self):
        return self.mat

    def vec(self):
        from .lie_algebra import torch_pose_to_tq
        return torch_pose_to_tq(self.mat)

    def to(self, *args, **kwargs):
        return SE3(self.mat.to(*args, **kwargs))

    def detach(self):
        return SE3(self.mat.detach())

    @property
    def shape(self):
        return self.mat.shape[:1]

    @property
    def device(self):
        return self.mat.device


class LieGroupParameter(torch.Tensor):
    """Leaf tensor holding a zero-initialised (B,6) tangent around a fixed group element; optimisers treat it
    like any parameter, ``retr()`` gives the current group element Exp(a) * X with autograd to ``a``."""

    @staticmethod
    def __new__(cls, group, requires_grad=True):
        data = torch.zeros(group.mat.shape[0], 6, dtype=group.mat.dtype, device=group.mat.device)
        return torch.Tensor._make_subclass(cls, data, requires_grad)

    def __init__(self, group, requires_grad=True):
        self.group = group.detach()

    def retr(self):
        return self.group.retr(self.as_subclass(torch.Tensor))

    def __deepcopy__(self, memo):
        out = LieGroupParameter(SE3(self.group.mat.clone()))
        out.data = self.data.clone()
        return out
