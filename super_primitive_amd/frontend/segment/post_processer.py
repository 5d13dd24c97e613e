"""Split keyframe segments at depth discontinuities and into connected components -- the reference's
``frontend/segment/post_processer.py`` API without cupy.

``kf_fix_disconnected_regions`` runs on the GPU end to end: fused exp + masked max-pool + Scharr threshold
(``sp_depth_discontinuity``), union-find labelling (``sp_label_components``), a compact list of components and their
sizes (``sp_collect_parts``), the selection rule of ``post_process_kf`` on that small list (host), then the new masks
(``sp_build_part_masks``) and one random keypoint per new part (``torch.randint`` exactly like the reference, then
``sp_kth_mask_pixel``).  The reference copies every mask to the host for ``ndimage.label`` and loops over segments in
Python with dense (k,H,W) tensors.
"""
from __future__ import annotations

import copy

import numpy as np
import torch

from ... import _lib
from ...tool import point_utils
from ...tool.etc import to_np


def _u8(mask):
    return mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous().to(torch.uint8)


def _discontinuity(logdepth, depth_validty, filter_size, threshold):
    _lib.require_device(logdepth, depth_validty)
    lib = _lib.load()
    N, H, W = logdepth.shape
    dev = logdepth.device
    scratch = torch.empty(N, H, W, dtype=torch.float32, device=dev)
    split = torch.empty(N, H, W, dtype=torch.bool, device=dev)
    disc = torch.empty(N, H, W, dtype=torch.bool, device=dev)
    _lib.check(lib.sp_depth_discontinuity(_lib.ptr(logdepth.detach().contiguous().float()), _lib.ptr(_u8(depth_validty)), N, H, W,
                                          int(filter_size), float(threshold), _lib.ptr(scratch), _lib.ptr(split), _lib.ptr(disc),
                                          _lib.stream_ptr()), "sp_depth_discontinuity")
    return split, disc


def depth_discontinuity(logdepth, depth_validty, filter_size=3, threshold=0.1):
    """(N,H,W) bool: valid pixels whose max-pooled depth has a Scharr gradient magnitude above ``threshold``."""
    return _discontinuity(logdepth, depth_validty, filter_size, threshold)[1]


def mask_by_depth_discontinuity(logdepth, depth_validity):
    return _discontinuity(logdepth, depth_validity, 3, 0.1)[0]


def _label(fg, want_sizes=True):
    _lib.require_device(fg)
    lib = _lib.load()
    N, H, W = fg.shape
    dev = fg.device
    parent = torch.empty(N * H * W, dtype=torch.int32, device=dev)
    labels = torch.empty(N, H, W, dtype=torch.int32, device=dev)
    sizes = torch.empty(N * H * W, dtype=torch.int32, device=dev) if want_sizes else None
    _lib.check(lib.sp_label_components(_lib.ptr(_u8(fg)), N, H, W, _lib.ptr(parent), _lib.ptr(labels), _lib.ptr(sizes),
                                       _lib.stream_ptr()), "sp_label_components")
    return labels, sizes


def batch_label_connectivity():
    """The 3x3x3 structuring element the reference passes to ndimage.label: 4-connectivity inside a slice, none
    across slices (post_processer.py:39-54)."""
    s = np.zeros((3, 3, 3), dtype=bool)
    s[1] = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
    return s


def connected_components_batch(masks):
    """(labelled int32 array on the host, number of components), numbered consecutively in scan order like
    ``ndimage.label`` (post_processer.py:57-64)."""
    m = masks if torch.is_tensor(masks) else torch.from_numpy(np.asarray(masks)).cuda()
    labels, _ = _label(m.bool(), want_sizes=False)
    flat = labels.reshape(-1)
    is_root = flat == (torch.arange(flat.numel(), device=flat.device, dtype=torch.int32) + 1)
    rank = torch.cumsum(is_root.to(torch.int32), 0)                       # consecutive number of each root, scan order
    out = torch.where(flat > 0, rank[(flat - 1).clamp(min=0).long()], torch.zeros_like(flat))
    return to_np(out.reshape(labels.shape)), int(is_root.sum())


def sample_pts_in_mask(masks):
    """One uniformly random (row, col) inside every mask; draws ``torch.randint`` per mask in order, like the
    reference (post_processer.py:67-84), and resolves the draw with one kernel instead of K ``torch.where`` calls."""
    _lib.require_device(masks)
    lib = _lib.load()
    K, H, W = masks.shape
    dev = masks.device
    m8 = _u8(masks)
    row_counts = torch.empty(K * H, dtype=torch.int32, device=dev)
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    seg_off = torch.empty(K + 1, dtype=torch.int32, device=dev)
    _lib.check(lib.sp_mask_count(_lib.ptr(m8), K, H, W, _lib.ptr(row_counts), _lib.ptr(counts), _lib.ptr(seg_off), _lib.stream_ptr()),
               "sp_mask_count")
    counts_h = counts.cpu().tolist()
    kth = torch.tensor([int(torch.randint(0, c, (1,))[0]) for c in counts_h], dtype=torch.int32, device=dev)
    rc = torch.empty(K, 2, dtype=torch.int32, device=dev)
    _lib.check(lib.sp_kth_mask_pixel(_lib.ptr(m8), _lib.ptr(row_counts), K, H, W, _lib.ptr(kth), _lib.ptr(rc), _lib.stream_ptr()),
               "sp_kth_mask_pixel")
    return rc.long()


def remap_labels_to_arange(array):
    return np.unique(array, return_inverse=True)[1].reshape(array.shape)


def remap_cupylabel_outputs(connectivity):
    return np.stack([remap_labels_to_arange(connectivity[i]) for i in range(connectivity.shape[0])], axis=0)


def _select_parts(n_seg, comp, bg_sizes, HW, keep_ratio):
    """post_process_kf's rule on the compact component list (post_processer.py:120-152): per segment the parts are
    [mask & !split] (if non-empty, it is ndimage's label 0 there) followed by the components in scan order; a part
    is kept when its area ratio exceeds ``keep_ratio``; no kept part -> segment dropped; one -> original mask kept."""
    by_seg = [[] for _ in range(n_seg)]
    for n, root, size in comp:
        by_seg[n].append((root, size))
    out = []          # (slice, kind, root)
    origin = []       # source segment of every new part
    for n in range(n_seg):
        parts = [(1, -1, int(bg_sizes[n]))]
        for root, size in sorted(by_seg[n]):
            parts.append((0, int(root), int(size)))
        kept = [p for p in parts if (np.float32(p[2]) / np.float32(HW)) > np.float32(keep_ratio)]
        if len(kept) == 0:
            continue
        if len(kept) == 1:
            out.append((n, 2, -1))
            origin.append((n, True))
        else:
            for kind, root, _ in kept:
                out.append((n, kind, root))
                origin.append((n, False))
    return out, origin


def post_process_kf(kf, connectivity=None, keep_ratio=1e-3, _split=None):
    """New (masks, logdepth, keypoints) after splitting.  ``connectivity`` (the host label array of the reference
    API) is accepted for signature parity but not needed: labelling is (re)done on the device from ``_split`` or from
    the keyframe itself."""
    _lib.require_device(kf.keypoints)
    lib = _lib.load()
    dev = kf.keypoints.device
    masks = kf.keypoint_regions
    N, H, W = masks.shape
    split = _split if _split is not None else mask_by_depth_discontinuity(kf.logdepth_perseg, masks)
    labels, sizes = _label(split)
    cap = max(1024, N * 64)
    while True:
        parts = torch.empty(cap, 3, dtype=torch.int32, device=dev)
        n_parts = torch.zeros(1, dtype=torch.int32, device=dev)
        bg = torch.empty(N, dtype=torch.int32, device=dev)
        _lib.check(lib.sp_collect_parts(_lib.ptr(labels), _lib.ptr(sizes), _lib.ptr(_u8(masks)), _lib.ptr(_u8(split)), N, H, W, cap,
                                        _lib.ptr(parts), _lib.ptr(n_parts), _lib.ptr(bg), _lib.stream_ptr()), "sp_collect_parts")
        n = int(n_parts.item())
        if n <= cap:
            break
        cap = n
    comp = [(int(a), int(b), int(c)) for a, b, c in parts[:n].cpu().numpy()]       # (slice, root linear index, size)
    chosen, origin = _select_parts(N, comp, bg.cpu().numpy(), H * W, keep_ratio)
    K = len(chosen)
    if K == 0:
        raise ValueError("every segment was dropped by the area filter")
    desc = torch.tensor(chosen, dtype=torch.int32, device=dev)
    new_masks = torch.empty(K, H, W, dtype=torch.bool, device=dev)
    _lib.check(lib.sp_build_part_masks(_lib.ptr(_u8(masks)), _lib.ptr(_u8(split)), _lib.ptr(labels), H, W, _lib.ptr(desc), K,
                                       _lib.ptr(new_masks), _lib.stream_ptr()), "sp_build_part_masks")
    src_seg = torch.tensor([n for n, _ in origin], dtype=torch.long, device=dev)
    new_logdepth = kf.logdepth_perseg[src_seg]
    # keypoints: untouched for segments that stay whole; one random pixel of each part otherwise, drawn segment by
    # segment in the reference's order so the same torch seed gives the same keypoints
    new_kp = torch.empty(K, 2, dtype=kf.keypoints.dtype, device=dev)
    whole = torch.tensor([w for _, w in origin], dtype=torch.bool, device=dev)
    if whole.any():
        new_kp[whole] = kf.keypoints[src_seg[whole]]
    if (~whole).any():
        idx = torch.nonzero(~whole).reshape(-1)
        pts = sample_pts_in_mask(new_masks[idx])
        new_kp[idx] = point_utils.normalise_coordinates(pts, (H, W)).to(new_kp.dtype)
    return new_masks, new_logdepth, new_kp


def kf_fix_disconnected_regions(kf, filter_size=3, depth_threshold=0.1, area_keep_ratio=1e-3):
    """post_processer.py:160-181.  Like the reference, the *split* used for labelling comes from
    ``mask_by_depth_discontinuity`` with its default filter/threshold (the arguments only feed the unused count)."""
    split = mask_by_depth_discontinuity(kf.logdepth_perseg, kf.keypoint_regions)
    new_mask, new_logdepth, new_keypoints = post_process_kf(kf, None, keep_ratio=area_keep_ratio, _split=split)
    # post_processer.py:176 of the reference: a deep copy, nothing aliased.  Tensor.__deepcopy__ also copies the tensors'
    # __dict__, i.e. the caches hanging off them (segment table with up to 8 sampled levels, packed target): drop those first --
    # the masks / log-depths they describe are replaced right below anyway
    from ... import segment_table
    segment_table.invalidate(kf)
    kf_new = copy.deepcopy(kf)
    kf_new.logdepth_perseg = new_logdepth
    kf_new.keypoint_regions = new_mask
    kf_new.keypoints = new_keypoints
    return kf_new
