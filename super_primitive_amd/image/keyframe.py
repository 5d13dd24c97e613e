"""KeyFrame container and keyframe pyramid -- the reference's ``image/keyframe.py`` API.

Attribute names are the reference's (GUIs and pickles read ``__dict__``): ``image`` (C,H,W) in [0,1], ``K``,
``K_img``, ``logdepth_perseg`` (N,H,W, zero outside the masks), ``keypoints`` (N,2) normalised (row,col),
``keypoint_regions`` (N,H,W) bool, ``id``.  Supporting frames carry only ``image`` and ``K``.

The compact segment table the HIP kernels consume is not stored here: it hangs off the ``keypoint_regions``
tensor (see ``segment_table.table_of``), so every pyramid level -- which shares that tensor when
``geo_down=False`` -- reuses one table.
"""
from __future__ import annotations

import torch
from torch import nn

from . import gaussian_pyramid
from ..tool import point_utils


def infer_spatial_size(logdepth_perseg):
    if logdepth_perseg.dim() == 3:
        return logdepth_perseg.shape[1:]
    assert logdepth_perseg.dim() == 2
    return logdepth_perseg.shape


_PREP_FIELDS = frozenset(("image", "K", "logdepth_perseg", "keypoints", "keypoint_regions", "segment_boxes"))


class KeyFrame(nn.Module):
    def __setattr__(self, name, value):
        # the batched set-up keeps a record of the five tensors it reads on the keyframe (optim/batch_prepare.py frame_records):
        # assigning any of them drops it
        if name in _PREP_FIELDS:
            self.__dict__.pop("_sp_prep", None)
        super().__setattr__(name, value)

    def __getstate__(self):
        # (copies and pickles must not carry the record: it holds the device addresses of THIS keyframe's tensors)
        state = self.__dict__.copy()
        state.pop("_sp_prep", None)
        return state

    def __init__(self, image, K, logdepth_perseg=None, keypoints=None, keypoint_regions=None, K_img=None, id=None):
        super().__init__()
        self.image = image
        self.K = K
        self.K_img = K if K_img is None else K_img
        self.id = id
        self.supporting = logdepth_perseg is None or keypoints is None or keypoint_regions is None
        # optional HINT for the batched set-up (not part of the reference's container): (N,4) int32 {row0, col0, row1, col1}, half open --
        # segment n has no mask pixel outside its box (SAM's frontend has exactly these: frontend/segment/mask_generation.py:93,155-180).
        # The count pass of optim/batch_prepare.py then reads the masks inside the boxes only; assign after construction.
        self.segment_boxes = None
        self.logdepth_perseg = None
        self.keypoints = None
        self.keypoint_regions = None
        if not self.supporting:
            assert keypoints.shape[0] == keypoint_regions.shape[0]
            self.logdepth_perseg = logdepth_perseg
            self.keypoints = keypoints
            self.keypoint_regions = keypoint_regions

    def geo_spatial_dim(self):
        return infer_spatial_size(self.get_logdepth())

    def get_logdepth(self):
        return self.logdepth_perseg

    def is_supporting(self):
        return self.supporting

    def normalised_keypoints(self):
        return point_utils.normalise_coordinates(self.keypoints.flip(-1), self.image.shape[1:])

    def num_segments(self):
        return self.keypoint_regions.shape[0]

    def __repr__(self):
        kp = f'with {self.keypoints.shape[0]} keypoints' if not self.supporting else 'with no keypoints'
        return f'{super().__repr__()} of shape {self.image.shape}\n{kp}'


def _to_gray(img):
    # ITU-R 601 luma, what torchvision's Grayscale(1) computes
    w = torch.tensor([0.2989, 0.587, 0.114], device=img.device, dtype=img.dtype)
    return (img * w[None, :, None, None]).sum(1, keepdim=True)


def keyframe_pyramid(keyframe, start_level, end_level, geo_down=False, drop_normals=False, grayscale=False):
    """List of KeyFrames, coarse -> fine, for levels start_level..end_level-1 (image/keyframe.py:77-148).

    With ``geo_down=False`` (what every caller uses) only the IMAGE is down-sampled; K, masks, log-depths and
    keypoints stay at full resolution and ``K_img`` carries the level intrinsics."""
    with torch.no_grad():
        extra_channels = keyframe.image.shape[0] > 3
        rgb = keyframe.image[:3][None]
        if grayscale:
            rgb = _to_gray(rgb)
        image_levels = gaussian_pyramid.ImagePyramidModule(rgb.shape[1], start_level, end_level, device=rgb.device,
                                                           dtype=rgb.dtype)(rgb)
        K_levels = gaussian_pyramid.IntrinsicsPyramidModule(start_level, end_level, device=rgb.device)(keyframe.K, [1.0, 1.0])
        n = len(image_levels)
        depth_levels = mask_levels = normal_levels = [None] * n
        nearest = gaussian_pyramid.DepthPyramidModule(start_level, end_level, mode='nearest_neighbor', device=rgb.device)
        if not keyframe.is_supporting() and geo_down:
            depth_levels = nearest(keyframe.logdepth_perseg.unsqueeze(1))
            mask_levels = nearest(keyframe.keypoint_regions.int().unsqueeze(1))
        if extra_channels:
            normal_levels = nearest(keyframe.image[3:][None])
        out = []
        for img, d, m, Kl, nrm in zip(image_levels, depth_levels, mask_levels, K_levels, normal_levels):
            img = img.squeeze(0)
            if nrm is not None and not drop_normals:
                img = torch.cat((img, nrm.squeeze(0)), dim=0)
            out.append(KeyFrame(img,
                                K=Kl if geo_down else keyframe.K.clone(),
                                logdepth_perseg=(d.squeeze(1) if d is not None else None) if geo_down else keyframe.logdepth_perseg,
                                keypoints=keyframe.keypoints,
                                keypoint_regions=(m.squeeze(1).bool() if m is not None else None) if geo_down else keyframe.keypoint_regions,
                                K_img=Kl, id=keyframe.id))
    return out


def put_keypoints_back(keypoints, masks, logdepth_perseg=None):
    """Snap each keypoint to the nearest pixel of its own mask; drop empty masks (image/keyframe.py:151-172)."""
    _, H, W = masks.shape
    kp = point_utils.denormalise_coordinates(keypoints, (H, W))
    keep = masks.sum(dim=(1, 2)) > 0
    kp, masks = kp[keep], masks[keep]
    if logdepth_perseg is not None:
        logdepth_perseg = logdepth_perseg[keep]
    for i in range(kp.shape[0]):
        rows, cols = torch.where(masks[i])
        j = torch.argmin(torch.sqrt((rows - kp[i, 0]) ** 2 + (cols - kp[i, 1]) ** 2))
        kp[i, 0], kp[i, 1] = rows[j], cols[j]
    new_kp = point_utils.normalise_coordinates(kp, (H, W))
    return (new_kp, masks, logdepth_perseg) if logdepth_perseg is not None else (new_kp, masks)


def put_keypoints_back_kf(keyframe):
    keyframe.keypoints, keyframe.keypoint_regions, keyframe.logdepth_perseg = put_keypoints_back(
        keyframe.keypoints, keyframe.keypoint_regions, keyframe.logdepth_perseg)
    return keyframe
