"""Normal-channel handling (API of the reference's ``core/normal_cost.py:5-31``).

With ``mode='colour'`` -- the only mode any reference caller or config selects -- there are no normal
channels and both functions return their input, which is what the fused HIP cost assumes.  For the other modes the
source normals (channels 3..5) are rotated into each target frame; that only ever touched the diagnostic
``src_pixels`` tensor, never the residual."""
import torch

from .cost_utils import split_by_mode


def _rotate_normal_channels(src_pixels, rotations, mode):
    """src_pixels (1,C,P), rotations (B,3,3) -> (B,C,P) with the normal channels rotated per batch element."""
    count = rotations.shape[0]
    colour, normals, kappa = split_by_mode(src_pixels.expand(count, -1, -1), mode=mode)
    turned = torch.matmul(rotations, normals[:1].expand(count, -1, -1))
    pieces = (colour, turned) if kappa is None else (colour, turned, kappa)
    return torch.cat(pieces, dim=1)


def transform_normals_batch(src_pixels, poses, mode="colour"):
    assert src_pixels.shape[0] == 1
    if mode == "colour":
        return src_pixels
    return _rotate_normal_channels(src_pixels, poses[:, :3, :3].detach(), mode)


def transform_normals(src_pixels, pose, mode="colour"):
    if mode == "colour":
        return src_pixels
    return transform_normals_batch(src_pixels, pose.unsqueeze(0), mode)
