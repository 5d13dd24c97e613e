"""Batched preparation of many frame pairs: point tables, image pyramids, per-level source samples and packed targets of
a whole ``PairBatch`` with a dozen launches and ONE host synchronisation (include/sp_hip.h, "Batched preparation").

The per-keyframe path (``segment_table.SegmentTable`` + ``image.gaussian_pyramid``) costs ~15 small launches, a host
synchronisation and a few hundred microseconds of Python per pair -- fine for one keyframe, 20x the optimiser's own time
when hundreds of new pairs are set up per batch.  Here every pass is one launch whose grid rows are job records in device
memory, the per-segment counts of ALL tables come back in one copy, and the host-side layout (padded runs, chunks, spans,
descriptors) is numpy-vectorised over the batch.

What it replaces per pair (reference): core/dense_optim.py:38-114 ``unproject_segments`` (the torch.where compaction),
image/gaussian_pyramid.py:53-85 (pyramids), core/dense_optim.py:315-317 (source sampling) -- same arithmetic and point
order as the per-keyframe kernels (shared device functions, sp_table.hip)."""
from __future__ import annotations

import ctypes
import os
import operator
import time

import numpy as np
import torch

from .. import _lib

GRANULE = 256


def _dev(t, dev):
    """``t`` on ``dev`` as contiguous float32 without touching tensors that already are."""
    if t.dtype == torch.float32 and t.device == dev and not t.requires_grad and t.is_contiguous():
        return t
    t = t.detach()
    if t.device != dev:
        t = t.to(dev)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_data_ptr, _is_contiguous, _numel = torch.Tensor.data_ptr, torch.Tensor.is_contiguous, torch.Tensor.numel
_dtype_of, _device_of = operator.attrgetter('dtype'), operator.attrgetter('device')


def handles(tensors, dev, dtype=torch.float32):
    """Device pointers of a list of tensors as a uint64 array, plus the tensors that own them.  The checks run through C-level
    ``map`` calls (0.3 us per tensor); only a list that fails one goes through the per-tensor conversion (``_dev``: 1.2 us each --
    3 ms of a 384-pair build when every input took it)."""
    n = len(tensors)
    if n and set(map(_dtype_of, tensors)) == {dtype} and set(map(_device_of, tensors)) == {dev} and all(map(_is_contiguous, tensors)):
        return np.fromiter(map(_data_ptr, tensors), dtype=np.uint64, count=n), tensors
    own = [_dev(t, dev) if dtype == torch.float32 else t.to(dev).contiguous() for t in tensors]
    return np.fromiter(map(_data_ptr, own), dtype=np.uint64, count=n), own


_FRAME_DT = np.dtype([('masks', '<u8'), ('image', '<u8'), ('logdepth', '<u8'), ('keypoints', '<u8'), ('K', '<u8'), ('N', '<i8'), ('H', '<i8'), ('W', '<i8'),
                      ('boxes', '<u8')])


def frame_records(frames, dev):
    """What the set-up needs of every source keyframe -- the device pointers of its masks, image, log-depths, keypoints and intrinsics
    and the mask shape -- as one structured array.  A keyframe's record is made once and kept ON THE KEYFRAME (``_sp_prep``; valid for
    as long as the five attributes are the same tensor objects): a keyframe is the source of many pairs -- every tracked frame, every
    window it is part of -- and validating five tensors per pair again was a third of the interpreter time of a build."""
    from ..image.keyframe import KeyFrame
    out = []
    for f in frames:
        hooked = type(f) is KeyFrame                 # (its __setattr__ drops the record when one of the five attributes is assigned)
        try:
            c = f.__dict__.get('_sp_prep')
        except AttributeError:
            c = None
        boxes = getattr(f, 'segment_boxes', None)
        if c is None or c[5] != dev or not (hooked or (c[0] is f.keypoint_regions and c[1] is f.image and c[2] is f.logdepth_perseg
                                                       and c[3] is f.keypoints and c[4] is f.K and c[7][0].data_ptr() == c[8] and c[9] is boxes)):
            m = f.keypoint_regions
            assert m.dtype == torch.bool and m.dim() == 3
            own = (m.contiguous(), _dev(f.image, dev), _dev(f.logdepth_perseg, dev), _dev(f.keypoints, dev), _dev(f.K, dev))
            if boxes is not None:                    # the optional segment-box hint of the count pass (SpPrepTable.boxes)
                if tuple(boxes.shape) != (m.shape[0], 4):
                    raise ValueError("segment_boxes: (N, 4) {row0, col0, row1, col1} per segment")
                own = own + (boxes.to(device=dev, dtype=torch.int32).contiguous(),)
            _lib.require_device(*own)
            rec = np.zeros(1, dtype=_FRAME_DT)
            rec[0] = tuple(t.data_ptr() for t in own[:5]) + tuple(m.shape) + ((own[5].data_ptr(),) if boxes is not None else (0,))
            c = (f.keypoint_regions, f.image, f.logdepth_perseg, f.keypoints, f.K, dev, rec.tobytes(), own, own[0].data_ptr(), boxes)
            # (the record is kept only when it points at the keyframe's OWN tensors -- a converted copy (other dtype, device or
            #  layout) would go stale at the caller's next in-place edit of the original, which no hook sees; such keyframes are converted
            #  again on every build)
            originals = (f.keypoint_regions, f.image, f.logdepth_perseg, f.keypoints, f.K)
            if all(o is t for o, t in zip(own[:5], originals)) and (boxes is None or own[5] is boxes):
                try:
                    f.__dict__['_sp_prep'] = c
                except AttributeError:
                    pass
        out.append(c)
    recs = np.frombuffer(b''.join([c[6] for c in out]), dtype=_FRAME_DT)
    return recs, [c[7] for c in out]


def segment_boxes_of(masks):
    """(N,4) int32 {row0, col0, row1, col1} (half open) of a bool (N,H,W) mask stack -- what ``KeyFrame.segment_boxes`` holds; an empty
    mask gets an empty box.  A frontend that made the masks has these already (frontend/segment/mask_generation.py:93,155-180); this is
    for callers that do not (two reductions over the stack: as much traffic as the count pass the hint saves -- worth it for a keyframe
    that is the source of many pairs)."""
    rows, cols = masks.any(dim=2), masks.any(dim=1)
    H, W = masks.shape[1:]
    first = lambda b: torch.where(b.any(1), b.float().argmax(1), torch.zeros_like(b[:, 0], dtype=torch.long))
    last = lambda b, n: torch.where(b.any(1), n - b.flip(1).float().argmax(1), torch.zeros_like(b[:, 0], dtype=torch.long))
    return torch.stack((first(rows), first(cols), last(rows, H), last(cols, W)), dim=1).to(torch.int32)


def flat_layout(counts, n_off, granule=GRANULE, table=None):
    """Padded layout of many tables at once.  counts: real points of every segment, all tables concatenated; n_off[t]:
    first segment of table t.  Returns (pc, seg_pos, p_off): padded run length of every segment, its position relative to
    its own table, and the tables' offsets into one flat array.  ``granule``: 256, or 64 for wave-span work lists."""
    counts = np.asarray(counts, dtype=np.int64)
    pc = (counts + granule - 1) // granule * granule
    cum = np.concatenate(([0], np.cumsum(pc)))
    p_off = cum[n_off]
    if table is None:
        table = np.repeat(np.arange(len(n_off) - 1), np.diff(n_off))
    return pc, cum[:-1] - p_off[table], p_off


def host_layout(counts, n_off, granule=GRANULE):
    """``flat_layout`` of several lattices at once through ``sp_host_layout``.  counts: (lattices, segments) int32.  Returns dict(counts
    int64, pc, seg_pos (lattices, segments), p_off (lattices, pairs + 1), points (lattices, pairs), seg_off_pinned: (lattices, 2 segments)
    int32 in pinned memory -- positions inside the flat array, then relative to the pair)."""
    lib = _lib.load()
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n_off = np.ascontiguousarray(n_off, dtype=np.int64)
    nL, S = counts.shape
    M = len(n_off) - 1
    pc, seg_pos = np.empty((nL, S), dtype=np.int64), np.empty((nL, S), dtype=np.int64)
    p_off, points = np.empty((nL, M + 1), dtype=np.int64), np.empty((nL, M), dtype=np.int64)
    seg_off = torch.empty((nL, 2 * S), dtype=torch.int32, pin_memory=True)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(lib.sp_host_layout(vp(counts), nL, S, vp(n_off), M, int(granule), vp(pc), vp(seg_pos), vp(p_off), ctypes.c_void_p(seg_off.data_ptr()), vp(points)),
               "sp_host_layout")
    return dict(counts=counts.astype(np.int64), pc=pc, seg_pos=seg_pos, p_off=p_off, points=points, seg_off_pinned=seg_off)


def flat_work_list(pc, seg_pos, n_off, span_points, tile_points, granule=GRANULE):
    """Chunks, spans and record offsets (include/sp_hip.h, "Work list") of all pairs at once (same chunks, same greedy spans,
    same order as ``pair_batch.build_work_list``), built by the library's host helper ``sp_host_work_list`` -- tens of
    microseconds where the numpy form below takes 1-2 ms per lattice.  pc / seg_pos: padded run length and pair-relative position
    of every segment; n_off: first segment of every pair.  Returns dict(chunks (C,4), spans (S,4), seg_tile_off (sum(N)+M,),
    sto_off (M+1,), c_off (M+1,), s_off (M+1,))."""
    lib = _lib.load()
    pc = np.ascontiguousarray(pc, dtype=np.int64)
    seg_pos = np.ascontiguousarray(seg_pos, dtype=np.int64)
    n_off = np.ascontiguousarray(n_off, dtype=np.int64)
    M, S = len(n_off) - 1, len(pc)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rec = 4 if granule == GRANULE else 1                  # segment records per chunk: one per wave, or one (wave spans)
    C = lib.sp_host_work_list_chunks(vp(pc), S, int(tile_points), int(granule))
    if C < 0:
        _lib.check(C, "sp_host_work_list_chunks")
    chunks = np.empty((max(C, 1), 4), dtype=np.int32)
    spans = np.empty((max(C, 1), 4), dtype=np.int32)
    seg_tile_off = np.empty(S + M, dtype=np.int32)
    sto_off, c_off, s_off = (np.empty(M + 1, dtype=np.int64) for _ in range(3))
    ns = lib.sp_host_work_list(vp(pc), vp(seg_pos), vp(n_off), M, int(span_points), int(tile_points), int(granule), rec, vp(chunks), vp(spans),
                               vp(seg_tile_off), vp(sto_off), vp(c_off), vp(s_off))
    if ns < 0:
        _lib.check(ns, "sp_host_work_list")
    return dict(chunks=chunks[:C], spans=spans[:ns], seg_tile_off=seg_tile_off, sto_off=sto_off, c_off=c_off, s_off=s_off)


def work_lists_staged(specs, tile_points, granule, dev):
    """The work lists of several lattices of one batch -- ``specs`` = [(pc, seg_pos, n_off, span_points)] -- written by
    ``sp_host_work_list`` STRAIGHT INTO one pinned staging buffer and sent to the device with one asynchronous copy (fresh numpy arrays
    of a megabyte each cost more in page faults than the helper's loops, and were then copied into the staging buffer anyway).
    Returns one dict per spec: device ``chunks`` (C,4), ``spans`` (S,4), ``seg_tile_off``; host ``sto_off``, ``c_off``, ``s_off``."""
    lib = _lib.load()
    rec = 4 if granule == GRANULE else 1
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    prepared, total = [], 0
    for pc, seg_pos, n_off, span_points in specs:
        pc = np.ascontiguousarray(pc, dtype=np.int64)
        seg_pos = np.ascontiguousarray(seg_pos, dtype=np.int64)
        n_off = np.ascontiguousarray(n_off, dtype=np.int64)
        M, S = len(n_off) - 1, len(pc)
        # (an upper bound of the number of chunks sizes the buffer: one more pass over the segments saved; the helper returns the count)
        chunk_max = max(int(granule), int(tile_points) // int(granule) * int(granule))
        C = S + int(pc.sum()) // chunk_max
        sizes = (16 * max(C, 1), 16 * max(C, 1), (4 * (S + M) + 15) // 16 * 16)
        prepared.append((pc, seg_pos, n_off, int(span_points), M, S, C, total, sizes))
        total += sum(sizes)
    host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
    base = host.data_ptr()
    def fill(item):
        pc, seg_pos, n_off, span_points, M, S, C, off, sizes = item
        sto_off, c_off, s_off = (np.empty(M + 1, dtype=np.int64) for _ in range(3))
        at = lambda k: ctypes.c_void_p(base + off + sum(sizes[:k]))
        ns = lib.sp_host_work_list(vp(pc), vp(seg_pos), vp(n_off), M, span_points, int(tile_points), int(granule), rec, at(0), at(1), at(2),
                                   vp(sto_off), vp(c_off), vp(s_off))
        return sto_off, c_off, s_off, ns, int(c_off[-1])

    # (the lattices' lists on pool threads side by side -- the helper runs without the interpreter lock -- measured slower than one after
    #  the other: 0.32 against 0.26 ms for three lists of 24 576 segments)
    heads = [fill(item) for item in prepared]
    for h in heads:
        if h[3] < 0:
            _lib.check(h[3], "sp_host_work_list")
    d = host.to(dev, non_blocking=True)
    out = []
    for (pc, seg_pos, n_off, span_points, M, S, _, off, sizes), (sto_off, c_off, s_off, ns, C) in zip(prepared, heads):
        i32 = lambda k, n: d[off + sum(sizes[:k]): off + sum(sizes[:k]) + 4 * n].view(torch.int32)
        out.append(dict(chunks=i32(0, 4 * C).reshape(C, 4), spans=i32(1, 4 * ns).reshape(ns, 4), seg_tile_off=i32(2, S + M),
                        sto_off=sto_off, c_off=c_off, s_off=s_off, n_chunks=C, n_spans=ns))
    return out


def flat_work_list_numpy(pc, seg_pos, n_off, span_points, tile_points):
    """The same work list in vectorised numpy (kept as the independent statement the host helper is tested against)."""
    pc = np.asarray(pc, dtype=np.int64)
    n_off = np.asarray(n_off, dtype=np.int64)
    M, S = len(n_off) - 1, len(pc)
    Ns = np.diff(n_off)
    pair_of_seg = np.repeat(np.arange(M), Ns)
    seg_in_pair = np.arange(S) - n_off[pair_of_seg]
    chunk_max = max(GRANULE, tile_points // GRANULE * GRANULE)
    k = -(-pc // chunk_max)                                                     # pieces per segment (0 for an empty one)
    per = -(-(pc // GRANULE) // np.maximum(k, 1)) * GRANULE                    # (nearly) equal, granule-aligned lengths
    cumk = np.concatenate(([0], np.cumsum(k)))
    C = int(cumk[-1])
    seg_idx = np.repeat(np.arange(S), k)
    j = np.arange(C) - cumk[:-1][seg_idx]
    start = seg_pos[seg_idx] + j * per[seg_idx]
    cnt = np.minimum(per[seg_idx], pc[seg_idx] - j * per[seg_idx])
    chunks = np.stack((pair_of_seg[seg_idx], seg_in_pair[seg_idx], start, cnt), axis=1).astype(np.int32)
    c_off = cumk[n_off]
    # per-pair CSR of the segment records: 4 records per chunk (one per wave)
    ext_pair = np.repeat(np.arange(M), Ns + 1)
    sto_off = n_off + np.arange(M + 1)
    ext_local = np.arange(S + M) - sto_off[:-1][ext_pair]
    seg_tile_off = (4 * (cumk[n_off[ext_pair] + ext_local] - cumk[n_off[ext_pair]])).astype(np.int32)
    # spans: greedy runs of consecutive chunks of one pair, all pairs advanced together
    cum = np.concatenate(([0], np.cumsum(cnt)))
    cur, end = c_off[:-1].copy(), c_off[1:]
    parts = []
    act = np.nonzero(cur < end)[0]
    while act.size:
        q0 = cur[act]
        q1 = np.searchsorted(cum, cum[q0] + span_points, side='right') - 1
        q1 = np.minimum(np.maximum(q1, q0 + 1), end[act])
        parts.append(np.stack((q0, q1 - q0, cum[q1] - cum[q0], act), axis=1))
        cur[act] = q1
        act = act[q1 < end[act]]
    spans = np.concatenate(parts) if parts else np.zeros((0, 4), dtype=np.int64)
    spans = spans[np.lexsort((spans[:, 0], spans[:, 3]))].astype(np.int32)
    s_off = np.concatenate(([0], np.cumsum(np.bincount(spans[:, 3], minlength=M))))
    return dict(chunks=chunks, spans=spans, seg_tile_off=seg_tile_off, sto_off=sto_off, c_off=c_off, s_off=s_off)


class PreparedTables:
    """Device arrays of one lattice stride for the base pairs: pix (sum Ppad,), src4 {level: (sum Ppad, 4)}, plus the host
    layout: counts / pc / seg_pos per segment and p_off per pair."""


_TABLE_DT, _SAMPLE_DT, _IMAGE_DT = np.dtype(_lib.SpPrepTable), np.dtype(_lib.SpPrepSample), np.dtype(_lib.SpPrepImage)
_IMAGE_PACK_DT = np.dtype(_lib.SpPrepImagePack)


_TORCH_OF = {np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}


def stage(arrays, dev):
    """Host arrays -> device tensors through ONE pinned staging buffer and one asynchronous copy (a copy from pageable memory
    would make the host wait for everything already enqueued on the stream).  int32 / float32 arrays come back typed and
    shaped, anything else (job records) as bytes; all are views of one device buffer."""
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, o in zip(arrays, offs):
        hv[o: o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    d = host.to(dev, non_blocking=True)
    out = []
    for a, o in zip(arrays, offs):
        v = d[o: o + a.nbytes]
        tt = _TORCH_OF.get(a.dtype)
        out.append(v.view(tt).reshape(a.shape) if tt is not None and tt != torch.uint8 else v)
    return out


_NO_PAIRED_SAMPLE = os.environ.get('SP_NO_PAIRED_SAMPLE', '0') == '1'       # (A/B switch of tools/setup_bench.py)


class _Timer:
    """Optional per-pass HIP-event timing of ``prepare_pairs`` (bench.py's ``roofline_setup``): ``with timer('count'): launch``."""

    def __init__(self):
        self.events = []
        self.marks = []                     # (name, host seconds since the timer was made): where the HOST is, for tools/setup_bench.py
        self.t0 = time.perf_counter()
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def mark(self, name):
        self.marks.append((name, time.perf_counter() - self.t0))

    def timeline(self):
        """[(pass, GPU start ms, GPU end ms)] since the timer was made, and the host marks in ms (call after a synchronisation)."""
        return [(n, self.e0.elapsed_time(a), self.e0.elapsed_time(b)) for n, a, b in self.events], [(n, 1e3 * t) for n, t in self.marks]

    class _Span:
        def __init__(self, owner, name):
            self.owner, self.name = owner, name

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

        def __exit__(self, *exc):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.owner.events.append((self.name, self.e0, e1))

    def __call__(self, name):
        return _Timer._Span(self, name)

    def milliseconds(self):
        """{pass: ms} (call after a synchronisation)."""
        out = {}
        for name, e0, e1 in self.events:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


class _NoTimer:
    class _Null:
        def __enter__(self):
            pass

        def __exit__(self, *exc):
            pass

    def __call__(self, name):
        return _NoTimer._Null()

    def mark(self, name):
        pass


def prepare_pairs(src_frames, trg_images, trg_Ks, klds, level_ids, coarse, dev, full_levels=None, timer=None, granule=GRANULE, depth_table=False):
    """Everything PairBatch needs from the raw frames, for the M0 given pairs.

    coarse: [(level, stride)] -- ADDITIONAL decimated tables (stride > 1) sampled at that level; the stride-1 tables are
    always built.  ``full_levels``: the pyramid levels at which the stride-1 (all points) tables are sampled NOW (default:
    every level); the others can be sampled later with the returned ``sample_full(levels)`` -- a schedule that iterates its
    coarse levels on decimated tables never reads them (2 x 16 B per point of writes and 24 of the 36 tap loads per point saved at
    3 levels).  Returns dict(tabs {stride: PreparedTables}, kp_L (sum N,), trg {level: (flat HWC3, trg_off, [(Hl, Wl)])}, n_off,
    shapes (M0, 3) = N, H, W, Ks (2 M0, 3, 3) = the source and target intrinsics on the HOST for the descriptors,
    sample_full, bytes = algorithmic bytes of every pass)."""
    timer = timer or _NoTimer()
    lib = _lib.load()
    M0 = len(src_frames)
    s_ptr = _lib.stream_ptr()
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError("super_primitive_amd: the frames live on a device that is not the current one")
    frec, frame_keep = frame_records(src_frames, dev)
    timer.mark('frame records')
    shp = np.stack((frec['N'], frec['H'], frec['W']), axis=1)                    # (M0, 3): N, H, W
    Ns, Hs, Ws = shp[:, 0], shp[:, 1], shp[:, 2]
    if (Hs > 32767).any() or (Ws > 65535).any():
        raise ValueError("image too large for the packed pixel word")
    n_off = np.concatenate(([0], np.cumsum(Ns)))
    S = int(n_off[-1])
    all_strides = sorted({1} | {int(s) for _, s in coarse if int(s) > 1})
    nS = len(all_strides)
    if nS > _lib.SP_PREP_MAX_STRIDES:
        raise ValueError(f"{nS} lattice strides exceed SP_PREP_MAX_STRIDES = {_lib.SP_PREP_MAX_STRIDES}")
    if len(level_ids) > _lib.SP_PREP_MAX_LEVELS:
        raise ValueError(f"{len(level_ids)} pyramid levels exceed SP_PREP_MAX_LEVELS = {_lib.SP_PREP_MAX_LEVELS}")

    # ---- pass 1: counts of every lattice of every keyframe (masks read once); the copy back is asynchronous ----
    rows = Ns * Hs
    rc_off = np.concatenate(([0], np.cumsum(rows)))
    row_counts = torch.empty(nS * int(rc_off[-1]), dtype=torch.int32, device=dev)
    counts_d = torch.empty(nS * S, dtype=torch.int32, device=dev)
    recs = np.zeros(M0, dtype=_TABLE_DT)
    recs['masks'] = frec['masks']
    recs['N'], recs['H'], recs['W'], recs['n_strides'] = Ns, Hs, Ws, nS
    for si, s in enumerate(all_strides):
        recs['stride'][:, si] = s
        recs['row_counts'][:, si] = row_counts.data_ptr() + 4 * (si * int(rc_off[-1]) + rc_off[:-1])
        recs['counts'][:, si] = counts_d.data_ptr() + 4 * (si * S + n_off[:-1])
    max_rows, max_N = int(rows.max()), int(Ns.max())
    # the masks as packed bit words (one per 16 pixels), written by the count pass and read by the fill pass instead of the masks;
    # only for keyframes on the count pass's fast path (include/sp_hip.h SpPrepTable.bits)
    fast = ((Ws % 16 == 0) & (Ws <= 1024) & (recs['masks'] % 16 == 0)) if all(s in (1, 2, 4, 8, 16) for s in all_strides) else np.zeros(M0, dtype=bool)
    words = np.where(fast, rows * (Ws // 16), 0)
    w_off = np.concatenate(([0], np.cumsum(words)))
    bits = torch.empty(max(int(w_off[-1]), 1), dtype=torch.int32, device=dev)
    recs['bits'] = np.where(fast, bits.data_ptr() + 4 * w_off[:-1], 0).astype(np.uint64)
    recs['boxes'] = np.where(fast, frec['boxes'], 0).astype(np.uint64)         # (the segment-box hint: fast path only)
    recs['logdepth'], recs['keypoints'] = frec['logdepth'], frec['keypoints']
    timer.mark('count records')
    staged = stage([recs], dev)
    timer.mark('count launch')
    with timer('count'):
        # (every keyframe of the batch on the fast path WITH the segment-box hint: the count pass that tests 64 blocks of rows against the boxes at once)
        count = lib.sp_prepare_count_boxed if bool((recs['boxes'] != 0).all()) else lib.sp_prepare_count
        _lib.check(count(_lib.ptr(staged[0]), M0, max_rows, max_N, s_ptr), "sp_prepare_count")
    counts_pinned = torch.empty(nS * S, dtype=torch.int32, pin_memory=True)
    counts_pinned.copy_(counts_d, non_blocking=True)
    counts_ready = torch.cuda.Event()
    counts_ready.record()
    # (Round 6, measured and NOT kept: the image passes -- pyramids, packed targets -- on a side stream next to the count and fill passes,
    #  which they do not depend on.  The passes then overlap and take as long together as one after the other: count 1.84 -> 3.18 ms,
    #  pyramid 0.82 -> 1.25, pack 0.68 -> 1.57, the whole set-up 5.94 -> 6.07 ms per 384 pairs (profiles/r06_setup_side_stream.txt).  The
    #  set-up as a whole runs at the rate the memory system sustains for its mix of reads and writes, 0.56-0.58 of the HBM peak; what is
    #  left to gain is in its BYTES, not in its scheduling.)

    # image pyramids of both frames and packed targets: independent of the counts, enqueued right behind the count pass
    # (the first three channels of a contiguous (C, H, W) image start where the image starts: no slicing of images with extra channels)
    timg_ptr, timg = handles(trg_images, dev)
    timer.mark('image handles')
    simg_ptr = frec['image']
    max_level = max(level_ids)
    # (Round 6: ONE pass per pyramid step makes the next level of both frames AND the packed forms of the target's levels
    #  (sp_prepare_blur_pack): as a separate pass the packing re-read every planar target level the step had just had in registers --
    #  12 B per target pixel and level and a launch; bit-identical.  A batch without a pyramid (one level) still packs by sp_prepare_pack.)
    hw = {0: np.stack((Hs, Ws), axis=1)}
    for l in range(1, max_level + 1):
        hw[l] = (hw[l - 1] + 1) // 2
    trg, packed_ptr = {}, {}
    for l in level_ids:
        sizes = 3 * hw[l][:, 0] * hw[l][:, 1]
        off = np.concatenate(([0], np.cumsum(sizes)))
        buf = torch.empty(int(off[-1]), dtype=torch.float32, device=dev)
        packed_ptr[l] = buf.data_ptr() + 4 * off[:-1]
        trg[l] = (buf, off, hw[l])                   # (flat packed targets, their offsets, (M0, 2) level sizes)
    pyramid, blur_jobs = [], []
    ptr_lv = {0: (simg_ptr, timg_ptr)}                                          # level -> (source, target) image pointers
    for l in range(1, max_level + 1):
        sizes = 3 * hw[l][:, 0] * hw[l][:, 1]
        off = np.concatenate(([0], np.cumsum(np.tile(sizes, 2))))
        buf = torch.empty(int(off[-1]), dtype=torch.float32, device=dev)
        jobs = np.zeros(2 * M0, dtype=_IMAGE_PACK_DT)
        jobs['inp'] = np.concatenate(ptr_lv[l - 1])
        jobs['out'] = buf.data_ptr() + 4 * off[:-1]
        jobs['H'], jobs['W'] = np.tile(hw[l - 1][:, 0], 2), np.tile(hw[l - 1][:, 1], 2)
        ptr_lv[l] = (jobs['out'][:M0].copy(), jobs['out'][M0:].copy())
        if l in packed_ptr:
            jobs['packed_out'][M0:] = packed_ptr[l]
        if l == 1 and 0 in packed_ptr:
            jobs['packed_in'][M0:] = packed_ptr[0]
        if l == max_level:
            jobs['out'][M0:] = 0                     # (nothing reads the targets' last planar level)
        blur_jobs.append(jobs)
        pyramid.append(buf)                  # read by later launches: must not return to the allocator before they are enqueued
    pack_jobs = np.zeros(M0 if max_level == 0 else 0, dtype=_IMAGE_DT)
    if max_level == 0:
        pack_jobs['inp'], pack_jobs['out'] = timg_ptr, packed_ptr[0]
        pack_jobs['H'], pack_jobs['W'] = hw[0][:, 0], hw[0][:, 1]
    staged = stage([pack_jobs] + blur_jobs, dev)
    timer.mark('pyramid jobs staged')
    if max_level > 0:
        with timer('pyramid'):
            for l in range(1, max_level + 1):
                _lib.check(lib.sp_prepare_blur_pack(_lib.ptr(staged[l]), 2 * M0, int((hw[l][:, 0] * hw[l][:, 1]).max()), s_ptr), "sp_prepare_blur_pack")
    else:
        with timer('pack'):
            _lib.check(lib.sp_prepare_pack(_lib.ptr(staged[0]), M0, int((hw[0][:, 0] * hw[0][:, 1]).max()), s_ptr), "sp_prepare_pack")

    # the per-pair inputs: initial log-depths and target intrinsics -- behind the count AND the pyramid passes, which are launched first
    # (round 6: the GPU waited 0.25 ms for the handles below before it got anything to do; with the segment-box hint the count pass is over
    # after 0.25 ms and the GPU waited again, for the pyramid jobs).  The intrinsics come back to the host for the descriptors: the host
    # waits for them at the end of this function, by when every pass has been enqueued.  The intrinsics of both frames and the log-depths
    # (one flat array, the optimisation variable) are collected by ONE gather launch from the pointer lists -- torch.stack / torch.cat over
    # hundreds of small tensors cost 0.7 us of interpreter time per tensor
    kld_ptr, kld = handles(klds, dev)
    Ktrg_ptr, Ktrg = handles(trg_Ks, dev)
    timer.mark('kld / K handles')
    if sum(map(_numel, klds)) != S or sum(map(_numel, trg_Ks)) != 9 * M0:
        raise ValueError("one (N_m,) log-depth vector and one (3,3) target intrinsics matrix per pair")
    # (one output buffer -- the 2 M0 intrinsics matrices, then the S log-depths -- and ONE launch over the 3 M0 sources)
    gathered = torch.empty(18 * M0 + S, dtype=torch.float32, device=dev)
    Ks_d, kld_flat = gathered[:18 * M0].view(2 * M0, 3, 3), gathered[18 * M0:]
    g = stage([np.concatenate((frec['K'], Ktrg_ptr, kld_ptr)), np.concatenate((9 * np.arange(2 * M0, dtype=np.int64), 18 * M0 + n_off.astype(np.int64)))], dev)
    _lib.check(lib.sp_prepare_gather(_lib.ptr(g[0]), _lib.ptr(g[1]), 3 * M0, _lib.ptr(gathered), s_ptr), "sp_prepare_gather")
    Ks_pinned = torch.empty(2 * M0, 3, 3, dtype=torch.float32, pin_memory=True)      # complete once the counts have been waited for
    Ks_pinned.copy_(Ks_d, non_blocking=True)
    Ks_ready = torch.cuda.Event()
    Ks_ready.record()

    timer.mark('gathers enqueued')

    # ---- host: padded layouts; device: fill straight into them ----
    timer.mark('wait for counts')
    t_wait = time.perf_counter()
    counts_ready.synchronize()                    # the one host synchronisation of the set-up (the pyramids keep the GPU busy)
    waited = time.perf_counter() - t_wait
    timer.mark('counts here')
    counts_h = counts_pinned.numpy().reshape(nS, S)
    tabs = {}
    kp_L = torch.empty(S, dtype=torch.float32, device=dev)
    recs['kp_L'] = kp_L.data_ptr() + 4 * n_off[:-1]
    # padded layouts of all lattices by the library's host helper (one pass; the numpy form -- flat_layout -- took a dozen array
    # operations per lattice), the segment positions straight into a pinned buffer
    lay = host_layout(counts_h, n_off, granule)
    if (lay['points'][all_strides.index(1)] == 0).any():
        raise ValueError("keyframe has no segment pixels")
    seg_off_d = lay['seg_off_pinned'].to(dev, non_blocking=True)
    dense_L = bool(((Ns.astype(np.int64) * Hs.astype(np.int64) * Ws.astype(np.int64)) < (1 << 32)).all())
    for si, s in enumerate(all_strides):
        t = PreparedTables()
        t.stride = s
        t.counts, t.pc, t.seg_pos, t.p_off, t.points = lay['counts'][si], lay['pc'][si], lay['seg_pos'][si], lay['p_off'][si], lay['points'][si]
        total = max(int(t.p_off[-1]), 1)
        t.pix = torch.empty(total, dtype=torch.int32, device=dev)               # (the sampler writes the padding: zero = invalid point)
        # (round 6: NO baseL copy of the points' log-depths -- SpPrepTable.baseL stays NULL, the sampler reads the keyframe's dense array at the
        #  point's (segment, row, column), SP_PREP_DENSE_L: 8 bytes per lattice point of writes and re-reads less, and the fill pass loses the
        #  log-depth loads that were its longest dependent chain.  Needs N H W < 2^32 per keyframe, else the copy is made as before)
        t.baseL = None if dense_L else torch.empty(total, dtype=torch.float32, device=dev)
        # segment positions: inside the flat array (fill) and relative to the pair's own table (sampler, cost kernels)
        t.counts_d = counts_d[si * S: (si + 1) * S]
        t.seg_off = seg_off_d[si]
        recs['pix'][:, si], recs['baseL'][:, si] = t.pix.data_ptr(), (0 if dense_L else t.baseL.data_ptr())
        recs['seg_off'][:, si] = t.seg_off.data_ptr() + 4 * n_off[:-1]
        t.src4 = {}
        tabs[s] = t
    timer.mark('layouts')

    # ---- source samples: the stride-1 tables at every level, a decimated table at its own level(s), all levels of a table
    #      in one pass (which also sets the table's source-validity bits) ----
    levels_of = {1: [int(l) for l in (level_ids if full_levels is None else full_levels)]}
    for l, s in coarse:
        if int(s) > 1 and int(l) not in levels_of.setdefault(int(s), []):
            levels_of[int(s)].append(int(l))

    def sample_jobs(levels_of):
        """SpPrepSample records sampling table ``stride`` at the levels ``levels_of[stride]`` (allocates the src4 arrays)."""
        jobs = np.zeros(len(levels_of) * M0, dtype=_SAMPLE_DT)
        max_P = 1
        for ji, (s, lv) in enumerate(levels_of.items()):
            t = tabs[s]
            jb = jobs[ji * M0: (ji + 1) * M0]
            P = np.diff(t.p_off)
            max_P = max(max_P, int(P.max()))
            jb['pix'] = t.pix.data_ptr() + 4 * t.p_off[:-1]
            jb['baseL'] = frec['logdepth'] if dense_L else t.baseL.data_ptr() + 4 * t.p_off[:-1]
            jb['seg_off'] = t.seg_off.data_ptr() + 4 * (S + n_off[:-1])              # the pair-relative half
            jb['counts'] = t.counts_d.data_ptr() + 4 * n_off[:-1]
            jb['kp_L'] = kp_L.data_ptr() + 4 * n_off[:-1]
            jb['kld'], jb['K'] = kld_ptr, frec['K']
            jb['N'], jb['P'], jb['H'], jb['W'], jb['n_levels'] = Ns, P, Hs, Ws, len(lv)
            jb['granule'] = granule | (_lib.SP_PREP_DEPTH_TABLE if depth_table else 0) | (_lib.SP_PREP_DENSE_L if dense_L else 0)      # (depth tables: src4.w = exp(L), include/sp_hip.h)
            for k, l in enumerate(lv):
                t.src4[l] = torch.empty(max(int(t.p_off[-1]), 1), 4, dtype=torch.float32, device=dev)
                jb['image'][:, k] = ptr_lv[l][0]
                jb['src4'][:, k] = t.src4[l].data_ptr() + 16 * t.p_off[:-1]
                jb['Hl'][:, k], jb['Wl'][:, k] = hw[l][:, 0], hw[l][:, 1]
        return jobs, max_P

    jobs, max_P = sample_jobs(levels_of)
    timer.mark('sample jobs')
    staged = stage([recs, jobs], dev)
    timer.mark('fill launch')
    with timer('fill'):
        _lib.check(lib.sp_prepare_fill(_lib.ptr(staged[0]), M0, max_rows, max_N, s_ptr), "sp_prepare_fill")
    with timer('sample'):
        # one launch per lattice: the grid is (blocks of the LARGEST table, jobs), and a stride-4 table has 1/16 of the points of
        # a stride-1 table -- in one launch over all lattices two thirds of the workgroups would start only to find nothing to do
        rec_bytes = _SAMPLE_DT.itemsize
        order = list(levels_of)
        grp_P = {s_: max(int(np.diff(tabs[s_].p_off).max()), 1) for s_ in order}
        job_ptr = lambda s_: ctypes.c_void_p(staged[1].data_ptr() + order.index(s_) * M0 * rec_bytes)
        # (round 6, late: two tables that gather from the SAME source level -- the all-points table and the stride-2 lattice, both at level 0 --
        #  in ONE launch, a keyframe's two records on consecutive workgroups: a launch per table fetches that level from memory twice)
        paired = None
        if not _NO_PAIRED_SAMPLE:
            for a in order:
                b = next((b for b in order if b != a and set(levels_of[a]) & set(levels_of[b])), None)
                if b is not None:
                    paired = (a, b)
                    break
        if paired:
            a, b = paired
            _lib.check(lib.sp_prepare_sample_pairs(job_ptr(a), grp_P[a], job_ptr(b), grp_P[b], M0, s_ptr), "sp_prepare_sample_pairs")
        for s_ in order:
            if paired and s_ in paired:
                continue
            _lib.check(lib.sp_prepare_sample(job_ptr(s_), M0, grp_P[s_], s_ptr), "sp_prepare_sample")

    def sample_full(levels):
        """Sample the stride-1 tables at further pyramid levels (those left out of ``full_levels``): {level: (sum Ppad, 4)}.
        Holds the source pyramid and the per-pair inputs alive for as long as the caller keeps this function."""
        levels = [int(l) for l in levels if int(l) not in tabs[1].src4]
        if levels:
            jb, mp = sample_jobs({1: levels})
            st = stage([jb], dev)
            _lib.check(lib.sp_prepare_sample(_lib.ptr(st[0]), len(jb), mp, _lib.stream_ptr()), "sp_prepare_sample")
        return tabs[1].src4

    sample_full._keep = (pyramid, frame_keep, kld)
    def pass_bytes():
        """algorithmic bytes of every pass: what it must read and write once (made on demand: bench.py asks)"""
        n_pts = {s: int(t.counts.sum()) for s, t in tabs.items()}
        img_px = {l: int((hw[l][:, 0] * hw[l][:, 1]).sum()) for l in hw}
        sampled_levels = sorted({l for lv in levels_of.values() for l in lv})
        # count pass: masks in (1 B / pixel and segment), bit words out -- with the segment-box hint only the 16-pixel pieces that meet a
        # segment's box, and the bit words of its box rows
        count_bytes = 0
        for i in range(M0):
            own = frame_keep[i]
            if fast[i] and len(own) > 5:
                b = own[5].cpu().numpy().astype(np.int64)
                r0, r1 = np.clip(b[:, 0], 0, Hs[i]), np.clip(b[:, 2], 0, Hs[i])
                c0, c1 = np.clip(b[:, 1], 0, Ws[i]), np.clip(b[:, 3], 0, Ws[i])
                box_rows = np.where((r1 > r0) & (c1 > c0), r1 - r0, 0)
                pieces = np.where(c1 > c0, -(-c1 // 16) - c0 // 16, 0)
                count_bytes += int((box_rows * pieces * 16).sum()) + 4 * int(box_rows.sum()) * int(Ws[i] // 16)
            elif fast[i] and all_strides[0] == 1:
                # (round 6, late: without boxes the bit words of EMPTY rows are not written -- the fill pass never reads them: a row's words
                #  are kept in registers until its count is known.  Rows that hold a pixel: counted here from the masks, on demand)
                full_rows = int(own[0].view(torch.uint8).reshape(int(Ns[i]), int(Hs[i]), int(Ws[i])).amax(dim=2).ne(0).sum())
                count_bytes += int(Ns[i] * Hs[i] * Ws[i]) + 4 * full_rows * int(Ws[i] // 16)
            else:
                count_bytes += int(Ns[i] * Hs[i] * Ws[i]) + 4 * int(words[i])
        return {
            'count': count_bytes,
            # fill: the set bits in, pix out per lattice point (+ with a baseL copy: L in once per mask pixel, baseL out per lattice point)
            'fill': (4 * sum(n_pts.values()) + n_pts[1] // 4) if dense_L else (4 * n_pts[1] + 8 * sum(n_pts.values()) + n_pts[1] // 4),
            'sample': sum((12 + 16 * len(lv)) * n_pts[s] for s, lv in levels_of.items())     # pix + baseL in, pix out, one src4 per sampled level out
                      + sum(12 * img_px[l] for l in sampled_levels),                         # ... and every sampled source level read once
            # both frames: level l-1 in, level l out (the targets' last planar level is not written); + the targets' packed levels out
            'pyramid': (sum(2 * 12 * (img_px[l - 1] + img_px[l]) for l in range(1, max_level + 1)) - (12 * img_px[max_level] if max_level > 0 else 0)
                        + (sum(12 * img_px[l] for l in level_ids) if max_level > 0 else 0)),
            'pack': sum(2 * 12 * img_px[l] for l in level_ids) if max_level == 0 else 0,     # (a batch without a pyramid: planar in, HWC3 out)
        }

    del pyramid
    # (temporaries -- job records, row counts -- are released here; the caching allocator orders their reuse after the launches
    #  above on this stream)
    t_wait = time.perf_counter()
    Ks_ready.synchronize()
    waited += time.perf_counter() - t_wait
    timer.mark('prepare_pairs returns')
    return dict(tabs=tabs, kp_L=kp_L, trg=trg, n_off=n_off, shapes=shp, Ks=Ks_pinned.numpy(), kld=kld_flat, sample_full=sample_full, bytes=pass_bytes,
                host_wait_s=waited)
