"""Host <-> device conveniences the drivers and visualisers call (API of the reference's ``tool/etc.py``).

Everything funnels through two private helpers: ``_host`` (tensor -> detached CPU tensor, anything else untouched) and
``_is_handle`` (names starting with ``_sp`` are device-side handles of this package -- segment tables, packed levels --
which must not travel to a GUI process)."""
import numpy as np
import torch


def _host(value):
    if isinstance(value, torch.Tensor):
        return value.detach().cpu()
    return value


def _is_handle(name):
    return isinstance(name, str) and name[:3] == "_sp"


def _without_handles(mapping):
    return {name: _host(value) for name, value in mapping.items() if not _is_handle(name)}


def dict_cpu(d):
    """Plain dict of CPU copies (materialises a lazy statistics dict, see core.dense_optim.LazyStats)."""
    return _without_handles(d)


def attrs_on_cpu(obj):
    """``vars(obj)`` with every tensor attribute copied to the host."""
    return _without_handles(vars(obj))


def list_cpu(items):
    return list(map(_host, items))


def to_np(tensor):
    """numpy view of a tensor's host copy; arrays pass through."""
    if isinstance(tensor, np.ndarray):
        return tensor
    return _host(tensor).numpy()


def to_img(tensor):
    """(C,H,W) tensor -> (H,W,C) array; arrays pass through."""
    if isinstance(tensor, np.ndarray):
        return tensor
    return np.moveaxis(to_np(tensor), 0, -1)


def to_img_np(tensor):
    """8-bit (H,W,C) image of a [0,1] tensor (truncating cast, like the reference)."""
    return np.asarray(to_img(tensor) * 255).astype(np.uint8)


def from_np(array):
    """Tensor owning a copy of ``array``; tensors pass through."""
    if torch.is_tensor(array):
        return array
    return torch.from_numpy(np.array(array, copy=True))


def image_tt(image, device="cuda"):
    """(H,W,C) uint8 image -> (C,H,W) float tensor in [0,1] on ``device``."""
    chw = torch.from_numpy(image).permute(2, 0, 1)
    return (chw / 255.0).float().to(device)
