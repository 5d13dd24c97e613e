"""Coordinate conventions shared by the whole path (mirror of the reference's ``tool/point_utils.py``).

Pixels <-> [-1, 1] use the align-corners convention ``2x/(d-1) - 1`` (tool/point_utils.py:31-35); the inverse
rounds half-to-even to an integer pixel (:37-40).  The "_og" pixel-centre variants are kept for API parity."""
import numpy as np
import torch


def _dims(dims, like):
    return torch.as_tensor(dims, dtype=torch.float32, device=like.device)


def normalise_coordinates(x_pixel, dims):
    return 2 * x_pixel * (1.0 / (_dims(dims, x_pixel) - 1)) - 1


def denormalise_coordinates(x_norm, dims):
    return (0.5 * (_dims(dims, x_norm) - 1) * (x_norm + 1)).round().long()


def normalise_coordinates_og(x_pixel, dims):
    inv = 1.0 / _dims(dims, x_pixel)
    return 2 * x_pixel * inv + inv - 1


def denormalise_coordinates_og(x_norm, dims):
    d = _dims(dims, x_norm)
    return (x_norm * d / 2.0 + d / 2.0 - 0.5).round().long()


def to_np(tensor):
    return tensor if isinstance(tensor, np.ndarray) else tensor.detach().cpu().numpy()


def img_to_np(img):
    if img.shape[0] == 1:
        img = img.squeeze(0)
    if img.shape[0] == 3:
        img = img.permute(1, 2, 0)
    return (img.detach().cpu().numpy() * 255).astype(np.uint8)


def normalise_coordinates_np(x_pixel, dims):
    return to_np(normalise_coordinates(torch.from_numpy(x_pixel.copy()), dims))


def denormalise_coordinates_np(x_norm, dims):
    return to_np(denormalise_coordinates(torch.from_numpy(x_norm.copy()), dims))
