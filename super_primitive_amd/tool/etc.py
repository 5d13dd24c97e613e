"""Host <-> device conveniences used by the drivers and visualisers (mirror of ``tool/etc.py``)."""
import numpy as np
import torch


def dict_cpu(d):
    # keys starting with "_sp" are device-side handles of this package (segment table etc.): not for the GUI process
    return {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in d.items() if not str(k).startswith("_sp")}


def list_cpu(items):
    return [v.detach().cpu() for v in items]


def attrs_on_cpu(obj):
    return {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in obj.__dict__.items()
            if not k.startswith("_sp")}


def to_np(tensor):
    return tensor if isinstance(tensor, np.ndarray) else tensor.detach().cpu().numpy()


def to_img(tensor):
    return tensor if isinstance(tensor, np.ndarray) else tensor.detach().cpu().numpy().transpose(1, 2, 0)


def to_img_np(tensor):
    return (to_img(tensor) * 255).astype(np.uint8)


def from_np(array):
    return array if isinstance(array, torch.Tensor) else torch.from_numpy(array.copy())


def image_tt(image, device="cuda"):
    return (torch.from_numpy(image) / 255.0).float().to(device).permute(2, 0, 1)
