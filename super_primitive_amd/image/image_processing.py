"""Scharr image gradients -- the reference's ``image/image_processing.py`` API (used by the segmentation frontend;
the keyframe post-processing path calls the fused HIP kernel ``sp_depth_discontinuity`` instead)."""
import torch
import torch.nn as nn


class ImageGradientModule(nn.Module):
    """3x3 Scharr / 32 per channel, reflect (default) or zero padding (image_processing.py:4-30)."""

    def __init__(self, channels, device, dtype, reflect_padding=True):
        super().__init__()
        a = torch.tensor([3.0, 10.0, 3.0], device=device, dtype=dtype) / 32.0
        d = torch.tensor([-1.0, 0.0, 1.0], device=device, dtype=dtype)
        self.kernel_x = torch.outer(a, d).view(1, 1, 3, 3).repeat(channels, 1, 1, 1)
        self.kernel_y = torch.outer(d, a).view(1, 1, 3, 3).repeat(channels, 1, 1, 1)
        self.reflect_padding = reflect_padding

    def forward(self, x):
        p = nn.functional.pad(x, (1, 1, 1, 1), mode='reflect' if self.reflect_padding else 'constant')
        return (nn.functional.conv2d(p, self.kernel_x, groups=x.shape[1]),
                nn.functional.conv2d(p, self.kernel_y, groups=x.shape[1]))


def get_image_grad(image):
    image = image.unsqueeze(0)
    gx, gy = ImageGradientModule(image.shape[1], image.device, image.dtype, reflect_padding=False)(image)
    return torch.stack((gx, gy), dim=2).squeeze(0)
