"""Compact per-segment point table of a source keyframe (device resident).

The reference re-derives the point list from dense (N,H,W) tensors on every iteration
(``core/dense_optim.py:38-114``: three dense passes, ``torch.where`` with a host sync, 3xP int64 indices).
Masks and base log-depths never change after a keyframe is built, so the build compacts them ONCE into

    pix[P] (row/col/validity), baseL[P], kp_L[N], seg_off[N+1]       -- level independent
    src4[P] = {source rgb at this pyramid level, baseL}              -- one per level, made on first use
    tiles[T] = {pair, segment, start, count}, seg_tile_off[N+1]      -- the work list of the cost kernel

(20 B per point per iteration instead of >= 5 dense N*H*W*4-byte passes).  The table is attached to the
``keypoint_regions`` tensor object, which ``keyframe_pyramid(geo_down=False)`` shares between all levels, and is
rebuilt whenever masks / log-depths / keypoints are replaced or modified in place.

Cache keys are OBJECT identities, not addresses: every cache entry keeps strong references to the tensors it was
built from and compares them with ``is`` (a freed address that the caching allocator hands to a new tensor can
therefore never alias an entry), plus the tensors' version counters for in-place edits.  Writes that bypass the
version counter (``t.data.copy_()``, ``t.data[...] = x``) are invisible to it: call ``invalidate(kf)`` after those.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

DEFAULT_TILE_POINTS = 1024


class _Ident:
    """Identity of a tensor for cache keys: the object itself (held, compared with ``is``) + its version counter."""
    __slots__ = ("t", "version")

    def __init__(self, t):
        self.t, self.version = t, t._version

    def matches(self, t):
        return t is self.t and t._version == self.version


def _same(idents, tensors):
    return idents is not None and len(idents) == len(tensors) and all(i.matches(t) for i, t in zip(idents, tensors))


def make_tiles(counts, tile_points, pair=0, first_point=0, granule=256):
    """Host-side work list: split every segment into ceil(count / tile_points) runs of (nearly) EQUAL length,
    each a multiple of ``granule`` (= the workgroup size, so every trip of the tile loop is full) except the
    last.  Equal runs matter: one workgroup per tile, and a 4096 + 1733 split costs as much as two 4096 tiles.

    Returns (tiles int32 (T,4) = {pair, segment, start, count}, seg_tile_off int32 (N+1,))."""
    counts = np.asarray(counts, dtype=np.int64)
    n_t = (counts + tile_points - 1) // tile_points
    seg_tile_off = np.zeros(len(counts) + 1, dtype=np.int32)
    np.cumsum(n_t, out=seg_tile_off[1:])
    T = int(seg_tile_off[-1])
    tiles = np.zeros((T, 4), dtype=np.int32)
    if T:
        seg = np.repeat(np.arange(len(counts)), n_t)
        within = np.arange(T) - np.repeat(seg_tile_off[:-1], n_t)
        seg_start = np.concatenate(([0], np.cumsum(counts)[:-1]))
        size = (counts + np.maximum(n_t, 1) - 1) // np.maximum(n_t, 1)          # ceil(count / n_t)
        size = (size + granule - 1) // granule * granule
        start = within * size[seg]
        tiles[:, 0] = pair
        tiles[:, 1] = seg
        tiles[:, 2] = first_point + seg_start[seg] + start
        tiles[:, 3] = np.clip(counts[seg] - start, 0, size[seg])
    return tiles, seg_tile_off


class SegmentTable:
    def __init__(self, masks, logdepth, keypoints, tile_points=DEFAULT_TILE_POINTS):
        _lib.require_device(masks, logdepth, keypoints)
        lib = _lib.load()
        assert masks.dtype == torch.bool and masks.dim() == 3
        N, H, W = masks.shape
        assert logdepth.shape == masks.shape and keypoints.shape == (N, 2)
        dev = masks.device
        self.N, self.H, self.W, self.device = N, H, W, dev
        masks_c = masks.contiguous()
        L_c = logdepth.detach().contiguous().float()
        kp_c = keypoints.detach().contiguous().float()
        row_counts = torch.empty(N * H, dtype=torch.int32, device=dev)
        counts = torch.empty(N, dtype=torch.int32, device=dev)
        self.seg_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
        s = _lib.stream_ptr()
        _lib.check(lib.sp_mask_count(_lib.ptr(masks_c), N, H, W, _lib.ptr(row_counts), _lib.ptr(counts),
                                     _lib.ptr(self.seg_off), s), "sp_mask_count")
        counts_h = counts.cpu().numpy()            # the one host sync of the table build (sizes the arrays)
        self.counts = counts_h
        self.P = int(counts_h.sum())
        if self.P == 0:
            raise ValueError("keyframe has no segment pixels")
        self.pix = torch.empty(self.P, dtype=torch.int32, device=dev)
        self.baseL = torch.empty(self.P, dtype=torch.float32, device=dev)
        self.kp_L = torch.empty(N, dtype=torch.float32, device=dev)
        _lib.check(lib.sp_table_fill(_lib.ptr(masks_c), _lib.ptr(L_c), _lib.ptr(kp_c), N, H, W, _lib.ptr(self.seg_off),
                                     _lib.ptr(row_counts), _lib.ptr(self.pix), _lib.ptr(self.baseL), _lib.ptr(self.kp_L),
                                     s), "sp_table_fill")
        self.set_tile_points(tile_points)
        self._levels = []          # [(idents of (image, K), src4)], most recent last
        self._key = None
        self._validity_set = False

    def set_tile_points(self, tile_points):
        tiles, sto = make_tiles(self.counts, tile_points)
        self.tile_points = tile_points
        self.n_tiles = tiles.shape[0]
        self.tiles = torch.from_numpy(tiles).to(self.device)
        self.seg_tile_off = torch.from_numpy(sto).to(self.device)

    # -- per-level source samples ---------------------------------------------------------------
    def source_level(self, image, K, kld, cache=True):
        """{rgb at this level, baseL} per point, cached per (image, K) object identity.  The first call after the table
        is built also sets the source-validity bit of ``pix`` (geometry only; shared by every level; it assumes depths
        above the reference's 1e-7 floor, i.e. finite log-depth seeds -- DESIGN.md section 5).  ``cache=False``: callers
        that pass freshly built pyramid tensors on every call (optim/window.py) would never hit the identity-keyed cache
        and only pin dead level images and their samples -- nothing is looked up or kept for them."""
        if cache:
            for idents, hit in self._levels:
                if _same(idents, (image, K)):
                    return hit
        _lib.require_device(image, K, kld)
        lib = _lib.load()
        img = image[:3].detach().contiguous().float()
        Hl, Wl = img.shape[-2:]
        src4 = torch.empty(self.P, 4, dtype=torch.float32, device=self.device)
        Kc = K.detach().contiguous().float()
        kc = kld.detach().contiguous().float()
        _lib.check(lib.sp_table_sample_source(_lib.ptr(self.pix), _lib.ptr(self.baseL), _lib.ptr(self.seg_off),
                                              _lib.ptr(self.kp_L), _lib.ptr(kc), self.N, self.P, self.H, self.W,
                                              _lib.ptr(img), Hl, Wl, _lib.ptr(Kc), _lib.ptr(src4),
                                              0 if self._validity_set else 1, _lib.stream_ptr()),
                   "sp_table_sample_source")
        self._validity_set = True
        if cache:
            if len(self._levels) >= 8:
                del self._levels[0]
            self._levels.append(((_Ident(image), _Ident(K)), src4))
        return src4


def table_of(kf, tile_points=None):
    """The (cached) SegmentTable of a keyframe-like object with the reference's attribute names."""
    masks, L, kp = kf.keypoint_regions, kf.get_logdepth() if hasattr(kf, "get_logdepth") else kf.logdepth_perseg, kf.keypoints
    tab = getattr(masks, "_sp_table", None)
    # (the table hangs off the masks object itself, so only the masks' version needs checking -- holding the masks
    #  in the key would make an uncollectable-by-refcount cycle that delays freeing the keyframe's device memory)
    if tab is None or tab._key[0] != masks._version or not _same(tab._key[1], (L, kp)):
        tab = SegmentTable(masks, L, kp, tile_points or DEFAULT_TILE_POINTS)
        tab._key = (masks._version, (_Ident(L), _Ident(kp)))
        try:
            masks._sp_table = tab
        except Exception:  # pragma: no cover
            pass
    elif tile_points is not None and tab.tile_points != tile_points:
        tab.set_tile_points(tile_points)
    return tab


def invalidate(kf_or_tensor):
    """Drop every cache hanging off a keyframe (or a single tensor): segment table, per-level source samples, packed
    target.  Needed only after writes that bypass torch's version counter (``.data``)."""
    objs = [kf_or_tensor] if torch.is_tensor(kf_or_tensor) else [getattr(kf_or_tensor, a, None) for a in
                                                               ("keypoint_regions", "logdepth_perseg", "keypoints", "image")]
    for t in objs:
        if torch.is_tensor(t):
            for attr in ("_sp_table", "_sp_rgba"):
                if hasattr(t, attr):
                    delattr(t, attr)


def packed_target(images):
    """(3,H,W) or (B,3,H,W) planar f32 -> (B,H,W,3) packed (HWC3), cached ON the tensor object (so the cache dies with
    it) and keyed on its version counter."""
    _lib.require_device(images)
    key = images._version
    hit = getattr(images, "_sp_rgba", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    lib = _lib.load()
    img = images.detach()
    if img.dim() == 3:
        img = img[None]
    img = img[:, :3].contiguous().float()
    B, _, H, W = img.shape
    out = torch.empty(B, H, W, 3, dtype=torch.float32, device=img.device)
    _lib.check(lib.sp_pack_rgb(_lib.ptr(img), B, H, W, _lib.ptr(out), _lib.stream_ptr()), "sp_pack_rgb")
    try:
        images._sp_rgba = (key, out)
    except Exception:  # pragma: no cover
        pass
    return out
