"""Small geometry operators on explicit point lists (mirror of ``core/ops.py``).

These are API-compatibility helpers for visualisers and drivers (``tool/viz.py:71,121`` calls
``dense_optim.project_points``).  The photometric hot path does NOT call them: it runs the fused HIP kernels in
``super_primitive_amd/csrc`` on the compact segment table.  ``estimate_depth_diff`` with table-backed inputs is
served by ``sp_depth_splat`` through ``core.depth_render``."""
import torch


def transform_points_batch(points_3d, poses):
    """(P,3)|(B,P,3), (B,4,4) -> (B,P,3)   (core/ops.py:5-17)."""
    R, t = poses[:, :3, :3], poses[:, :3, 3]
    eq = "bij,nj->bni" if points_3d.dim() == 2 else "bij,bnj->bni"
    return torch.einsum(eq, R, points_3d) + t[:, None, :]


def project_points_batch(points_3d, K):
    """(B,P,3), (B,3,3) -> (B,P,2) = (u,v); 1/z is replaced by 1e-6 where |z| <= 1e-6 (core/ops.py:19-40)."""
    eps = 1e-6
    x, y, z = points_3d[..., 0], points_3d[..., 1], points_3d[..., 2]
    safe = z.abs() > eps
    z_inv = torch.where(safe, 1.0 / torch.where(safe, z, torch.ones_like(z)), torch.full_like(z, eps))
    u = x * K[..., 0, 0][:, None] * z_inv + K[..., 0, 2][:, None]
    v = y * K[..., 1, 1][:, None] * z_inv + K[..., 1, 2][:, None]
    return torch.stack((u, v), dim=-1)


def project_points(points_3d, K):
    return project_points_batch(points_3d[None], K[None])[0]


def transform_points(points_3d, pose):
    return transform_points_batch(points_3d[None], pose[None])[0]


def unproject_points_mat(points_2d, depth_2d, K):
    homog = torch.cat((points_2d.float(), torch.ones_like(points_2d[:, :1], dtype=torch.float32)), dim=1)
    return (homog * depth_2d.reshape(-1, 1)) @ torch.inverse(K).T


def estimate_depth_diff(points_3d, K, spatial_dim, mean=False):
    """Explicit-point-list form of the z splat (core/ops.py:59-96), kept for callers that hold raw points.
    Keyframe renders go through ``core.depth_render.estimate_depth_kf_native`` (HIP)."""
    H, W = int(spatial_dim[0]), int(spatial_dim[1])
    z = points_3d[..., 2]
    with torch.no_grad():
        rc = project_points(points_3d, K).flip(-1).long()
    r, c = rc[..., 0], rc[..., 1]
    ok = (z.detach() > 1e-6) & (r >= 0) & (r < H) & (c >= 0) & (c < W)
    flat = torch.zeros(H * W, device=points_3d.device, dtype=torch.float32)
    idx = r[ok] * W + c[ok]
    if mean:
        flat.scatter_reduce_(0, idx, z[ok], reduce="mean")
    else:
        flat.scatter_(0, idx, z[ok])
    return flat.reshape(1, H, W), ok
