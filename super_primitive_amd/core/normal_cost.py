"""Normal-channel handling (mirror of ``core/normal_cost.py:5-31``).

For ``mode='colour'`` -- the only mode any reference caller or config uses -- both functions are
the identity, which is what the fused HIP cost assumes.  The rotating branch is kept for API parity; it touches
only the diagnostic ``src_pixels`` tensor, never the residual."""
import torch

from .cost_utils import split_by_mode


def transform_normals(src_pixels, pose, mode="colour"):
    if mode == "colour":
        return src_pixels
    return transform_normals_batch(src_pixels, pose[None], mode)


def transform_normals_batch(src_pixels, poses, mode="colour"):
    assert src_pixels.shape[0] == 1
    if mode == "colour":
        return src_pixels
    B = poses.shape[0]
    expanded = src_pixels.expand(B, -1, -1)
    _, normals, _ = split_by_mode(src_pixels, mode=mode)
    colour, _, kappa = split_by_mode(expanded, mode=mode)
    rotated = torch.einsum("bij,bjn->bin", poses[:, :3, :3].detach(), normals)
    parts = [colour, rotated] + ([kappa] if kappa is not None else [])
    return torch.cat(parts, dim=1)
