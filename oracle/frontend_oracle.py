"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's keyframe post-processing (frontend/segment/post_processer.py:13-181): depth
discontinuity mask, per-slice 4-connected components, area filter, re-seeded keypoints.  The reference calls cupy's
``ndimage.label``; cupy mirrors scipy's function, which is what is used here.  Pinned by golden vectors produced by
the REAL reference module with ``cupy`` / ``cupyx.scipy.ndimage`` stubbed by numpy / scipy (oracle/gen_goldens.py).
"""
import numpy as np
import scipy.ndimage
import torch
import torch.nn.functional as F

FOUR = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)


def discontinuity(logdepth, valid, filter_size=3, threshold=0.1):
    depth = torch.exp(logdepth)
    depth[~valid] = -1
    pooled = F.max_pool2d(depth[:, None], filter_size, stride=1, padding=filter_size // 2)
    p = F.pad(pooled, (1, 1, 1, 1), mode='reflect')
    k = torch.tensor([[-3.0, 0, 3], [-10, 0, 10], [-3, 0, 3]]) / 32.0
    gx = F.conv2d(p, k[None, None])
    gy = F.conv2d(p, k.T.contiguous()[None, None])
    disc = (torch.sqrt(gx ** 2 + gy ** 2).squeeze(1) > threshold) & valid
    return disc, valid & ~disc


def label_slices(split):
    """Consecutive labels in (slice, row, col) scan order, 4-connectivity inside each slice."""
    out = np.zeros(split.shape, dtype=np.int32)
    total = 0
    for n in range(split.shape[0]):
        lab, k = scipy.ndimage.label(split[n].numpy(), structure=FOUR)
        out[n] = np.where(lab > 0, lab + total, 0)
        total += k
    return out, total


def fix_disconnected(frame, keep_ratio=1e-3):
    """Returns (masks (K,H,W) bool, logdepth (K,H,W), keypoints (K,2)); consumes torch's global RNG like the
    reference (one ``torch.randint`` per part of every segment that is split into more than one kept part)."""
    L, masks, kps = frame.logdepth_perseg, frame.keypoint_regions, frame.keypoints
    N, H, W = masks.shape
    _, split = discontinuity(L, masks)
    labels, _ = label_slices(split)
    dims = torch.tensor([H, W], dtype=torch.float32)
    new_m, new_L, new_k = [], [], []
    for n in range(N):
        ids = np.unique(labels[n])                      # sorted; 0 (if present) is everything outside the split mask
        parts = [torch.from_numpy(labels[n] == i) & masks[n] for i in ids]
        keep = [m for m in parts if (m.sum().float() / (H * W)) > keep_ratio]
        if len(keep) == 0:
            continue
        if len(keep) == 1:
            new_m.append(masks[n][None]); new_L.append(L[n][None]); new_k.append(kps[n][None])
            continue
        pts = []
        for m in keep:
            r, c = torch.where(m)
            j = torch.randint(0, r.shape[0], (1,))[0]
            pts.append(torch.stack((r[j], c[j])))
        pts = torch.stack(pts)
        new_m.append(torch.stack(keep)); new_L.append(L[n].expand(len(keep), -1, -1))
        new_k.append(2 * pts * (1.0 / (dims - 1)) - 1)
    return torch.cat(new_m), torch.cat(new_L), torch.cat(new_k)
