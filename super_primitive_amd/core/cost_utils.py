"""Channel layout of the sampled pixel tensors (mirror of ``core/cost_utils.py:4-20``).

Only the first three (colour) channels ever enter the residual in the reference: the cosine / kappa terms are
never assigned (``core/dense_optim.py:241-261``), every caller passes ``mode='colour'`` and every shipped config
has ``include_normals: False``."""
import torch


def split_by_mode(src_pixels, mode="colour"):
    if mode == "colour":
        return src_pixels[:, :3], None, None
    if mode == "colour_norm":
        colour, normals = torch.split(src_pixels, [3, 3], dim=1)
        return colour, normals, None
    if mode == "colour_norm_kappa":
        colour, normals, kappa = torch.split(src_pixels, [3, 3, 1], dim=1)
        return colour, normals, kappa
    if mode == "norm_kappa":
        normals, kappa = torch.split(src_pixels, [3, 1], dim=1)
        return None, normals, kappa
    raise ValueError(f"unknown residual mode {mode!r}")
