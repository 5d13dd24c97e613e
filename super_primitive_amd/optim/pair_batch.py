"""Many independent frame pairs optimised side by side on one GPU (BASELINE.json configs 2 and 5).

Each pair is one two-frame problem of the reference (``odometery/two_frame_sfm.py``): a source keyframe with N
segments, a target frame, unknowns = relative pose (SE(3)) + one log-depth per segment.  Pairs never interact,
so M of them are packed into flat device arrays, described by one ``SpPair`` record each, and every optimiser
iteration is exactly two launches for the WHOLE batch, with no host synchronisation:

    sp_pairs_cost   (grid = all spans of all pairs; fused cost + gradient | Gauss-Newton normal equations)
    sp_pairs_*_step (grid = M; fixed-order reduction of the partial records, Adam+SE(3) retraction | Schur-complement LM solve)

This is the configuration the HBM-roofline number is measured on: one 640x480x64 pair is ~10 MB of algorithmic
traffic and lives in the 256 MB Infinity Cache, M >= 64 pairs stream from HBM.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib
from ..image import gaussian_pyramid
from . import batch_prepare

_SP_PAIR_DTYPE = np.dtype(_lib.SpPair)

DEFAULT_BATCH_TILE_POINTS = 8192   # longest chunk (run of one segment's points between two segment-level flushes)
DEFAULT_SPAN_POINTS = 16384        # most points per workgroup: consecutive chunks of a pair are grouped up to this many
MIN_SPANS = 2304                   # ... but a launch should still have about this many workgroups (256 CUs x 4 x 2.25)
GRANULE = 256                      # SP_BLOCK: every segment is padded to a multiple of it in the batch's tables

# The coarse-to-fine schedule "frame pairs per second" is quoted on (bench.py) and that tests/test_gpu_fullsize.py
# requires to land within the north-star bar (1e-4 rad / 1e-4 t / 1e-3 relative depth) of the minimiser of the
# reference cost at 640x480x64 (golden g15): LM iterations per pyramid level with the default IRLS epsilon and PER-PAIR
# termination on the device (PairBatch.run_converging: a pair leaves a level once an accepted step buys less than
# ``conv_tol`` of its cost), then a polish at the finest level with the epsilon at ``polish_eps`` under ``polish_tol``.
# FIXED_FRAME_PAIR_SCHEDULE is the same without early termination (every pair runs the maximum), kept for comparison.
# FRAME_PAIR_POINT_STRIDE: PairBatch(point_stride=...) of that schedule -- at pyramid level l (image decimated 2^l times) the
# schedule's Gauss-Newton iterations run on the source points whose pixel coordinates are multiples of the stride (1 / stride^2
# of them: about one source sample per TARGET pixel of that level instead of 4^l); the finest level and the polish use every
# point, so the minimiser reached is that of the full reference cost.  (The reference evaluates every full-resolution source
# point at every level, odometery/two_frame_sfm.py:128-207; every per-level method of PairBatch does too.)
# Round 4: level 0 ALSO iterates on its stride-2 lattice (a quarter of the points); only the polish -- the phase that fixes the end state,
# with its own convergence test -- runs on all points.  Once the resident set is kept full (run_scheduled(slots=...)) the frame-pair rate
# is bound by the cost kernel's throughput, i.e. by the WORK per pair, and the all-points level-0 phase was half of it: 29.6 k -> 38.1 k
# pairs/s near start, 25.1 k -> 30.6 k from the reference's start, same converged fraction, same end states (tools/phase_sweep.py,
# profiles/r04_phase_sweep.txt).
FRAME_PAIR_POINT_STRIDE = (2, 2, 4)
# Round 6: ``predicted_exit`` (SP_PHASE_PREDICTED_EXIT) -- a pair leaves a phase right after the step that is PREDICTED to buy less than
# the phase's tolerance (the first-order change of sum |r| along the step, which the solver has at hand) instead of after one more
# evaluation has shown that it did.  That confirming evaluation was a SIXTH of a frame pair's work (one per phase; the all-points polish's
# alone a tenth): 45.4 -> 38.9 iterations and 50 -> 39 cost evaluations per pair, 33.8 k -> 52.1 k pairs/s from the reference's start on
# the grid, the same convergence record (9216 grid + 9216 ragged starts: nothing missed, nothing more flagged; the worst end state a
# little BETTER, 6.6e-6 against 7.8e-6 rad: the last step is taken instead of thrown away), and on the ragged starts the two
# Gauss-Newton attempts lost, five of six now come home without the third (profiles/r06_reference_start_predicted_exit_*.txt).
FRAME_PAIR_SCHEDULE = dict(max_iters_per_level=25, conv_tol=2e-3, polish_max=15, polish_eps=1e-5, polish_tol=1e-4, check_every=3, predicted_exit=True)
FIXED_FRAME_PAIR_SCHEDULE = dict(iters_per_level=15, polish_iters=10, polish_eps=1e-5)
# The schedule for the REFERENCE'S OWN starting distribution (odometery/two_frame_sfm.py:77-81,103-105: pose = T_gt Exp(0.05
# randn(6)), depth seeds log(2 + 2 rand)): tests/test_gpu_sigma05.py requires it to converge wherever the real reference loop does
# (golden g19), inside the north-star bar of the reference's end state; bench.py quotes ``frame_pairs_per_sec`` on it next to the
# near-start figure (tools/sigma05_sweep.py holds the sweep it was chosen from, profiles/r03_sigma05_sweep.txt its results).
# Round 5: THE SECOND ATTEMPT.  About one of the reference's own starts in 650 ends in the wrong basin at the first attempt whichever way
# the coarse phases are arranged, and which one is a matter of round-off (caps of 15 / 30, a looser IRLS epsilon,
# a fourth level -- each loses DIFFERENT pairs; the basin is always the same one: every depth runs off to infinity and the pose
# explains the image by a rotation, 0.017 rad / 0.048 t from the truth at 14 x the cost).  So the schedule ends with a per-pair VERDICT
# on the device (include/sp_hip.h SpVerdict: finiteness, how the polish ended, the largest log-depth excursion from the seeds, the valid
# fraction) and a pair that fails it is put back to its start and run once more, in the slot it already has, with ``retry_phases`` in
# front -- a pose-only phase at the coarsest level with the OTHER IRLS epsilon (1e-2 against 1e-3: a smoother or a sharper cost.  Either
# order rescued EVERY first-attempt failure among 9216 starts; pose-only phases on a fourth pyramid level rescued 8 of 14,
# tools/verdict_sweep.py, profiles/r05_reference_start.txt) -- and then the same joint phases and polish.  What fails twice is reported (``PairBatch.status``,
# ``failed()``), never returned silently.
REFERENCE_START_LEVELS = (0, 3)
REFERENCE_START_POINT_STRIDE = (2, 2, 4)
# pose_first_iters is a CAP (the phase ends by its convergence test): 15 cut the 2-4 sigma tail of the start distribution short (rotation
# errors of 0.09-0.22 rad need ~20 pose-only iterations; 7 of bench.py's 1536 starts, among them the one of its first 384, diverged in the
# joint phase -- while the real reference converges from that start, golden g20x); 30 loses 2 of 1536 (tools/hard_starts.py,
# profiles/r04_reference_start.txt).  Longer is not better without the convergence test: a pose fitted for 25 iterations to depth seeds
# that are 50 % off (IRLS epsilon 1e-2, where the test triggers late) loses 12 %.
# Which pose-only phase goes first is a matter of speed: epsilon 1e-2 with a cap of 15 in the first attempt and epsilon 1e-3 with a cap of
# 30 in the second (below) loses 9 of 9216 starts at the first attempt and none after the second, at 33.6 k pairs/s and 32.9 iterations per
# pair; the other way round (round 4's first attempt) 14 / none at 31.2 k and 36.7 (profiles/r05_reference_start.txt).
# ``coarse_damped`` = "auto" -> (16, 12) or (12, 12) by the batch's segment statistics (PairBatch.auto_coarse_damping; 16 for the headline
# workloads): between the pose-only phase and the joint phases, 12 iterations at the coarsest level with the log-depth
# block DAMPED by 16 (include/sp_hip.h SP_PHASE_DEPTH_DAMP: the depths follow at 1/17 of their Gauss-Newton step -- the reference's Adam
# moves them at a tenth of the pose's rate).  On the grid tiling the undamped schedule is enough (9 second attempts per 9216 starts, all
# rescued); on SAM-like ragged masks over near-planar scenes (bench.py --shape blobs: depth range e^0.2) it walks into the second solution
# of the plane's homography from 6 % of the reference's starts and the second attempt brings home only half of those -- 375 of 12288
# flagged, where the real reference converges (goldens g20y).  With the damped phase: 13 second attempts and THREE flagged pairs of 12288,
# at a HIGHER rate there (24.4 k against 21.4 k pairs/s: no time lost in failing attempts) and 30 k against 32.7 k on the grid, no second
# attempt among 9216 starts.  The strength is a compromise over workloads (flagged after the second attempt; tools/verdict_sweep.py,
# profiles/r05_reference_start_sweep_5_*.txt):
#     damping     64 ragged segments (of 12288)   300 ragged (of 3072)   1200 ragged (of 1536)   128 grid segments (of 3072)
#        8                 ~24                            9                     31                       0
#       12                   6                            7                     23                       0
#       16                   3                            7                     35                       0
#       20                   4                           17                     71                       0
#       24                   4                           34                     97                       0
#       31                   2                           82                    129                       0
# (many small segments have few points each on the coarse lattice: depths held still for 12 iterations there let the pose settle on a
# biased optimum.)  Outcomes of starts near the basin boundary turn with round-off: golden g20y's pair 90 converges with 16 in bench's
# layout (replicas share replica 0's source samples: colours one ulp apart) and fails -- flagged -- as a batch of one built from its own
# depth seeds; tests/test_gpu_sigma05.py runs both.
REFERENCE_START_RETRY = (dict(level=2, stride=4, max_iters=30, irls_eps=1e-3, conv_tol=2e-3, pose_only=True),)
# Round 6: THE THIRD ATTEMPT IS THE REFERENCE'S OWN OPTIMISER.  Thirteen of 49152 of the reference's starts on ragged masks end both
# Gauss-Newton attempts in the second solution of the near-plane's homography (all flagged; profiles/r05_reference_start_sweep_6_*) --
# and the real reference loop converges from them (goldens g20y: 2437 and, this round, three more).  Adam at the reference's rates and
# budget (lr 1e-2 on the pose tangent, 1e-3 on the log-depths, 500 iterations per pyramid level coarse to fine, one optimiser over the
# three levels: odometery/two_frame_sfm.py:116-123,128-155) brings home 12 of the 13 on the device (tools/third_attempt_probe.py,
# profiles/r06_third_attempt_probe.txt: 500 at the coarsest level alone 10, 500 + 300 at the two coarsest 11, 300 + 200: 11): so a pair
# that fails its verdict twice restarts from its initial values once more through SP_PHASE_ADAM phases (include/sp_hip.h) -- the same
# cost pass, an Adam step instead of the Schur solve -- on the level's lattice, then joins the schedule at its finest joint phase and is
# polished like everybody else.  It keeps its slot for those 1500 rounds; at 3 pairs in 10000 the throughput does not notice.
REFERENCE_START_ADAM = (dict(level=2, stride=4, max_iters=500, irls_eps=1e-5, conv_tol=0.0, adam=True),
                        dict(level=1, stride=2, max_iters=500, irls_eps=1e-5, conv_tol=0.0, adam=True),
                        dict(level=0, stride=2, max_iters=500, irls_eps=1e-5, conv_tol=0.0, adam=True))
ADAM_LR_POSE, ADAM_LR_KLD = 1e-2, 1e-3          # odometery/two_frame_sfm.py:116-123
REFERENCE_START_SCHEDULE = dict(FRAME_PAIR_SCHEDULE, pose_first_iters=15, pose_first_eps=1e-2, coarse_damped="auto", retry_phases=REFERENCE_START_RETRY,
                                retry2_phases=REFERENCE_START_ADAM)
# The verdict's thresholds (SpVerdict): a log-depth more than ``kld_bound`` from its seed (a factor e^kld_bound in depth: the reference's
# seeds log(2 + 2 rand) are at most a factor 2 off) has run away; fewer than ``valid_min`` of the points projecting into the target
# frame at the end of an alignment that started with both frames overlapping is a lost pair.  ``retry_on``: the status bits that send
# a pair into its second attempt.  tools/verdict_sweep.py / profiles/r05_reference_start.txt: what each bit catches over 8192 starts.
# ``cost_outlier``: the one test that looks at the BATCH (on the host side of the run, a handful of tensor operations, no synchronisation):
# a final cost above this multiple of the batch's median final cost.  On ragged SAM-like masks a few starts in a thousand end in a
# genuine local minimum of the cost -- the polish converges at once, no depth runs away, nothing a pair can see by itself is wrong --
# at 8-50 x the cost of the converged pairs, whose final costs lie within 1.15 x of their median (profiles/r05_reference_start.txt).  Such
# pairs get the second attempt too (a short second scheduled run over them alone) and SP_STATUS_COST if they are still outliers.  Needs
# at least COST_OUTLIER_MIN_PAIRS pairs; a batch of one has no median to speak of.
# ``seg_max_ratio`` / ``seg_mean_ratio`` (round 6, SP_STATUS_SEGMENTS): THE WITHIN-PAIR TEST -- the worst segment's mean |r| against the
# pair's own median segment, and the pair's cost against that median.  What a pair in the plane's second solution looks like from the
# inside: some segments explained, others not.  Over the 13 known misses of 49152 ragged starts: worst / median 3.1-44, cost / median
# 1.12-2.7 (the well-behaved local minimum of start 9847, which no other per-pair test sees: 4.4 / 1.49); over every CONVERGED pair of
# the sweeps of round 6 (grid 64 segments, ragged 64 / 300 / 1200: 32 k pairs, profiles/r06_reference_start_*): worst / median at most
# 2.75 / 2.96 / 3.34 / 3.78, cost / median at most 1.09 / 1.14 / 1.10 / 1.11.  Shipped: 8 and 1.3 -- the cost ratio does the work (a
# third above the largest converged value, a seventh below 9847's), the worst-segment ratio only catches the grossly uneven.  It needs
# no second pair: a batch of ONE is judged like a batch of thousands.  (The three misses below 1.3 -- 2437, 35432 and one more -- carry
# run-away depths, SP_STATUS_DEPTH_RANGE.)  ``seg_product`` = 0.9 on (cost / median - 1) x (worst / median): HELD-OUT SAM-realistic scenes
# (seeds 7000..., profiles/r06_reference_start_heldout_*) produced a wrong-basin end state at 1.27 / 4.75 -- under both single thresholds, which
# cannot come down: converged pairs reach 1.19 (sam) and 3.78 (1200 small segments).  Its product is 1.28 (9847: 2.2) where the PRODUCT OF THE
# MAXIMA over every converged pair of every sweep is 0.51: a moderately raised cost together with a clear outlier segment.
COST_OUTLIER_MIN_PAIRS = 8
VERDICT_DEFAULTS = dict(kld_bound=2.0, cost_bound=0.0, cost_ratio=0.0, valid_min=0.5, cost_outlier=4.0, seg_max_ratio=8.0, seg_mean_ratio=1.3, seg_product=0.9,
                        retry_on=_lib.SP_STATUS_NONFINITE | _lib.SP_STATUS_LAST_CAP | _lib.SP_STATUS_DEPTH_RANGE | _lib.SP_STATUS_VALID | _lib.SP_STATUS_COST
                        | _lib.SP_STATUS_SEGMENTS)


def _level_images(img, max_level):
    """[(3,H,W) at level 0, level 1, ...] through the HIP blur+decimate kernel."""
    out, cur = [img], img[None]
    for _ in range(max_level):
        cur = gaussian_pyramid.blur_decimate(cur)
        out.append(cur[0])
    return out


def pad_layout(counts, device):
    """Padded layout of one segment table: segment n occupies [pseg_off[n], pseg_off[n] + counts[n]) + padding up to a
    multiple of GRANULE.  ``dst`` maps table point i to its padded position."""
    counts = np.asarray(counts, dtype=np.int64)
    pc = (counts + GRANULE - 1) // GRANULE * GRANULE
    pseg_off = np.concatenate(([0], np.cumsum(pc)))
    seg_off = np.concatenate(([0], np.cumsum(counts)))
    dst = torch.from_numpy(np.repeat(pseg_off[:-1] - seg_off[:-1], counts) + np.arange(int(counts.sum()))).to(device)
    return dict(pc=pc, pseg_off=pseg_off, Ppad=int(pseg_off[-1]), dst=dst)


def pad_points(x, pd):
    """(P, ...) table array -> (Ppad, ...) with zeros (= invalid points) in the padding."""
    out = torch.zeros((pd['Ppad'],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out[pd['dst']] = x
    return out


def build_work_list(pads, span_points, tile_points):
    """Chunks {pair, seg, start, count}, spans {first chunk, n chunks, points, pair} and the per-pair record offsets of
    the many-pairs cost kernels (include/sp_hip.h, "Work list") for pairs whose padded layouts are ``pads``.
    Returns dict(chunks (C,4) int32, spans (S,4) int32, seg_rec_offs [per pair (N+1,) int32], c_off, s_off)."""
    chunk_max = max(GRANULE, tile_points // GRANULE * GRANULE)
    chunks, spans, seg_rec_offs, c_off, spans_per_pair = [], [], [], [0], []
    for m, pd in enumerate(pads):
        first = len(chunks)
        sto = [0]
        for n, (pc, off) in enumerate(zip(pd['pc'], pd['pseg_off'][:-1])):
            k = int(-(-pc // chunk_max))                       # pieces of (nearly) equal, granule-aligned length
            if k:
                per = int(-(-(pc // GRANULE) // k)) * GRANULE
                done = 0
                while done < pc:
                    cnt = int(min(per, pc - done))
                    chunks.append((m, n, int(off + done), cnt))
                    done += cnt
            sto.append(4 * (len(chunks) - first))              # records: 4 per chunk (one per wave)
        seg_rec_offs.append(np.asarray(sto, dtype=np.int32))
        c_off.append(len(chunks))
        # spans: greedy runs of consecutive chunks
        ns, q = 0, first
        while q < len(chunks):
            q1, pts = q, 0
            while q1 < len(chunks) and (q1 == q or pts + chunks[q1][3] <= span_points):
                pts += chunks[q1][3]
                q1 += 1
            spans.append((q, q1 - q, pts, m))
            ns += 1
            q = q1
        spans_per_pair.append(ns)
    return dict(chunks=np.asarray(chunks, dtype=np.int32).reshape(-1, 4), spans=np.asarray(spans, dtype=np.int32).reshape(-1, 4),
                seg_rec_offs=seg_rec_offs, c_off=c_off, s_off=np.concatenate(([0], np.cumsum(spans_per_pair))))


_SIDE_STREAMS = {}           # device -> [torch.cuda.Stream]: the extra streams of run_scheduled(streams=K)


class _Layout:
    """A point set with its work list: pix / src4 tables, chunks / spans, per-level descriptors and partial buffers."""


class _Lazy(dict):
    """{level: value} whose missing entries are made on first access (full-resolution source samples and descriptors of
    pyramid levels a decimated schedule never touches)."""

    def __init__(self, make, *a, **kw):
        super().__init__(*a, **kw)
        self._make = make

    def __missing__(self, key):
        self[key] = v = self._make(key)
        return v


class PairBatch:
    def __init__(self, src_frames, trg_images, trg_Ks, poses, klds, levels=(0, 3), use_affine=False,
                 tile_points=DEFAULT_BATCH_TILE_POINTS, zmin=1e-7, replicate=1, span_points=None, point_stride=None,
                 extra_tables=(), lazy_levels=True, timer=None, granule=GRANULE, depth_table=True):
        """src_frames: keyframe-like objects (image, K, logdepth_perseg, keypoints, keypoint_regions) on one cuda
        device; trg_images: list of (3,H,W); trg_Ks: list of (3,3); poses: (M,4,4) initial target<-source;
        klds: list of (N_m,) initial keypoint log-depths; levels = (pyramid_min, pyramid_max) like
        ``config['aligment']`` (max exclusive).  ``replicate`` = R > 1 lays the M0 given pairs out R times in
        device memory (distinct copies of every array, poses/klds = ``poses[r*M0+m]`` when (R*M0,4,4) poses are
        given, log-depths = ``klds[r*M0+m]`` when R*M0 vectors are given): bench.py uses it to build a large streaming batch
        without uploading R*M0 dense keyframes.

        Layout (include/sp_hip.h, "Work list"): the batch keeps its own PADDED copy of every table -- each segment's
        run of points extended to a multiple of 256 with invalid points -- so that a workgroup can stream through a SPAN
        of several consecutive chunks (segments, or pieces of at most ``tile_points`` points of a long segment) of up
        to ``span_points`` points without any trip mixing two segments.  ``span_points=None``: 16384, reduced for small
        batches so that a launch keeps about 2300 workgroups (a single pair then runs one chunk per workgroup).

        ``point_stride`` (one integer per pyramid level, finest first; default all 1): levels with a stride s > 1 get, IN
        ADDITION, a decimated copy of the point tables -- the mask pixels whose column and row are multiples of s -- with its
        own work list, descriptors and partial buffers (``self.coarse[(level, s)]``); ``extra_tables`` = further (level,
        stride) combinations for explicit ``schedule(phases=...)`` lists.  Only ``run_scheduled`` uses them; every per-level
        method below (cost_pass, gn_step, adam_step, evaluate, run, run_converging) works on all points.
        ``lazy_levels``: the all-points source samples (and descriptors) of a level that has a decimated table are made on first
        use instead of at construction -- the scheduled run never reads them (``self.src4`` / ``self.desc`` fill in on access).
        ``timer``: a ``batch_prepare._Timer`` that collects per-pass HIP-event times of the set-up.
        ``granule`` = 64: WAVE SPANS (include/sp_hip.h SP_COST_WAVE_SPANS) -- segments are padded to multiples of 64 instead of
        256 points, a span belongs to one wave, one segment record per chunk.  For batches of many small ragged segments: 1200
        SAM-like masks of ~280 pixels pad 40 % at 256 and 11 % at 64.  The single-launch forms (``fused=True``) are not available.
        ``depth_table`` (default): the batch's tables hold exp(L) in ``src4[..., 3]`` instead of L and the cost passes run in their
        depth-table form (include/sp_hip.h SP_COST_DEPTH_TABLE: one multiply per point instead of add + multiply + v_exp_f32, the
        exponential of the segment's shift taken once per chunk; -1.8 % kernel time).  ``False``: log-depth tables, as the per-keyframe
        path (``segment_table``) keeps them; the single-launch forms (``fused=True``) need that."""
        assert granule in (GRANULE, 64)
        self.granule = int(granule)
        self.rec_per_chunk = 4 if granule == GRANULE else 1
        self.wave_flag = 0 if granule == GRANULE else _lib.SP_COST_WAVE_SPANS
        self.depth_table = bool(depth_table)
        self.table_flag = _lib.SP_COST_DEPTH_TABLE if self.depth_table else 0
        lib = _lib.load()
        self.lib = lib
        M0 = len(src_frames)
        R = int(replicate)
        M = M0 * R
        assert M0 == len(trg_images) == len(trg_Ks) and len(klds) in (M0, M) and poses.shape[0] in (M0, M)
        if poses.shape[0] == M0 and R > 1:
            poses = poses.repeat(R, 1, 1)
        dev = src_frames[0].image.device
        _lib.require_device(src_frames[0].image)
        self.M, self.device = M, dev
        self.level_ids = list(range(levels[0], levels[1]))
        self.tile_points = tile_points
        self.point_stride = {l: 1 for l in self.level_ids}
        if point_stride is not None:
            assert len(point_stride) == len(self.level_ids), "one stride per pyramid level, finest first"
            self.point_stride = {l: int(s) for l, s in zip(self.level_ids, point_stride)}
        coarse_keys = sorted({(l, s) for l, s in self.point_stride.items() if s > 1} | {(int(l), int(s)) for l, s in extra_tables if int(s) > 1})

        # tables, pyramids, source samples and packed targets of the base pairs: a dozen launches, one host synchronisation
        # (optim/batch_prepare.py)
        klds_all = klds if len(klds) == M and R > 1 else None
        klds = klds[:M0]
        decimated = {l for l, s in coarse_keys if l in self.point_stride and self.point_stride[l] == s}
        full_levels = [l for l in self.level_ids if not (lazy_levels and l in decimated)] or [min(self.level_ids)]
        prep = batch_prepare.prepare_pairs(src_frames, trg_images, trg_Ks, klds, self.level_ids, coarse_keys, dev, full_levels=full_levels,
                                           timer=timer, granule=self.granule, depth_table=self.depth_table)
        self._setup_bytes = prep['bytes']                      # (callable: the algorithmic bytes of every set-up pass, made on demand)
        self.setup_host_wait_s = prep['host_wait_s']          # of the constructor's wall time, what the host spent WAITING for the GPU (the counts)
        mark = timer.mark if timer is not None else (lambda name: None)
        tabs, kp_L, trg, n_off0 = prep['tabs'], prep['kp_L'], prep['trg'], prep['n_off']
        rep = (lambda x: x) if R == 1 else (lambda x: x.repeat(*([R] + [1] * (x.dim() - 1))))
        tile = (lambda a: a) if R == 1 else (lambda a: np.tile(a, R))
        Ns0 = np.diff(n_off0)
        Ns = tile(Ns0)
        n_off = np.concatenate(([0], np.cumsum(Ns)))
        full = tabs[1]
        counts, pc, seg_pos = tile(full.counts), tile(full.pc), tile(full.seg_pos)
        Ppad = tile(np.diff(full.p_off))
        p_off = np.concatenate(([0], np.cumsum(Ppad)))
        self.Ns = Ns.tolist()
        self.Ps = tile(full.points).tolist()
        self.Ppads = Ppad.tolist()
        self.max_N = int(Ns.max())
        self.n_off, self.p_off = n_off, p_off

        # flat, pair-major device arrays
        self.kp_L = rep(kp_L)
        if klds_all is None:
            self.kld = rep(prep['kld']) if R > 1 else prep['kld']          # (gathered into one flat array by the set-up: owned, updated in place)
        else:
            self.kld = torch.cat([batch_prepare._dev(k, dev).reshape(-1) for k in klds_all]).contiguous()
            assert self.kld.numel() == int(n_off[-1]), "one (N_m,) log-depth vector per replicated pair"
        self.pose = poses.detach().to(device=dev, dtype=torch.float32).reshape(M, 16).clone()       # owned: updated in place
        self.aff = torch.zeros(M, 4, dtype=torch.float32, device=dev) if use_affine else None
        self.pix = rep(full.pix)
        sample_full = prep['sample_full']
        self.src4 = _Lazy(lambda l: rep(sample_full([l])[l]).reshape(-1), {l: rep(full.src4[l]).reshape(-1) for l in full_levels})
        self.trg4 = {l: rep(trg[l][0]) for l in self.level_ids}
        trg_off = {l: np.concatenate(([0], np.cumsum(tile(np.diff(trg[l][1]))))) for l in self.level_ids}
        hl_of = {l: np.tile(trg[l][2].astype(np.int32), (R, 1)) for l in self.level_ids}                    # (M, 2) level sizes
        self.level_hw = _Lazy(lambda l: list(map(tuple, hl_of[l].tolist())))                                  # {level: [(Hl, Wl)] per pair}

        # (wave spans: a span is a quarter of a workgroup's worth of work, so four times as many of them)
        span_div = 1 if self.granule == GRANULE else 4
        if span_points is None:
            span_points = min(DEFAULT_SPAN_POINTS, int(p_off[-1]) // MIN_SPANS) // span_div
        self.span_points = max(int(span_points), self.granule)
        # work lists: chunks {pair, seg, start, count}, spans {first chunk, n chunks, points, pair} and the per-pair CSR of the segment
        # records of the all-points tables and of every decimated lattice (the levels that share a lattice share its list), made by
        # the library's host helper straight into one staging buffer and sent asynchronously (the preparation kernels are still running)
        same = lambda a: len(set(self.Ns)) == 1 and bool((np.asarray(a).reshape(M, -1) == np.asarray(a).reshape(M, -1)[0]).all())
        self._uniform_layout = same(pc)
        self.coarse = {}
        specs, spec_of, c_host = [(pc, seg_pos, n_off, self.span_points)], {}, {}
        for l, stride in coarse_keys:
            if stride not in spec_of:
                t = tabs[stride]
                c_pc, c_seg_pos = tile(t.pc), tile(t.seg_pos)
                c_p_off = np.concatenate(([0], np.cumsum(tile(np.diff(t.p_off)))))
                c_span = max(self.granule, min(DEFAULT_SPAN_POINTS, int(c_p_off[-1]) // MIN_SPANS) // span_div)
                spec_of[stride] = len(specs)
                specs.append((c_pc, c_seg_pos, n_off, c_span))
                c_host[stride] = (c_p_off, tile(t.points))
                self._uniform_layout = self._uniform_layout and same(c_pc)
        mark('device arrays')
        lists = batch_prepare.work_lists_staged(specs, tile_points, self.granule, dev)
        mark('work lists')
        wl = lists[0]
        self.chunks, self.spans, self.seg_tile_off = wl['chunks'], wl['spans'], wl['seg_tile_off']
        self.n_chunks, self.n_spans = wl['n_chunks'], wl['n_spans']
        self._s_off = wl['s_off']
        self.span_pair = self.spans[:, 3].long()
        self.n_seg_records = self.rec_per_chunk * self.n_chunks
        # decimated point sets of the coarse levels (run_scheduled): own tables, work list, descriptors, partial buffers
        coarse_host = {}
        for l, stride in coarse_keys:
            t = tabs[stride]
            lay = _Layout()
            lay.stride = stride
            c_wl = lists[spec_of[stride]]
            c_p_off, c_points = c_host[stride]
            lay.points = c_points.tolist()
            shared = next((o for (l2, s2), o in self.coarse.items() if s2 == stride), None)
            lay.pix = shared.pix if shared is not None else rep(t.pix)
            lay.src4 = rep(t.src4[l]).reshape(-1)
            lay.n_chunks, lay.n_spans = c_wl['n_chunks'], c_wl['n_spans']
            lay.s_off = c_wl['s_off']
            lay.chunks, lay.spans, lay.seg_tile_off = c_wl['chunks'], c_wl['spans'], c_wl['seg_tile_off']
            coarse_host[(l, stride)] = (c_wl, c_p_off)
            self.coarse[(l, stride)] = lay

        # descriptors, one array per level (numpy view of struct SpPair, filled column-wise)
        Ks = prep['Ks']
        Ks_src, Ks_trg = Ks[:M0], Ks[M0:]
        k4 = lambda K: np.tile(np.stack((K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]), axis=1).astype(np.float32), (R, 1))
        K4_src, K4_trg = k4(Ks_src), k4(Ks_trg)
        HW = np.tile(prep['shapes'][:, 1:].astype(np.int32), (R, 1))                    # full-resolution source size
        pair_idx = np.arange(M, dtype=np.int64)

        # (everything the descriptor maker needs is bound to LOCALS: the lazily evaluated entries of self.src4 / self.desc keep this
        #  closure alive, and a closure over ``self`` would make every PairBatch a reference cycle -- its ~25 MB per pair of tables
        #  would then wait for the cyclic garbage collector instead of being freed when the last reference goes)
        kp_L_t, trg4_t, kld_t, pose_t, aff_t = self.kp_L, self.trg4, self.kld, self.pose, self.aff
        pix_t, seg_tile_off_t, Ps_a, rec_per_chunk = self.pix, None, np.asarray(self.Ps), self.rec_per_chunk

        def descriptors(level, pix, src4, seg_tile_off, lay_p_off, lay_wl, real_points):
            d = np.zeros(M, dtype=_SP_PAIR_DTYPE)
            d['pix'] = pix.data_ptr() + 4 * lay_p_off[:-1]
            d['src4'] = src4.data_ptr() + 16 * lay_p_off[:-1]
            d['kp_L'] = kp_L_t.data_ptr() + 4 * n_off[:-1]
            d['trg3'] = trg4_t[level].data_ptr() + 4 * trg_off[level][:-1]
            d['kld'] = kld_t.data_ptr() + 4 * n_off[:-1]
            d['pose'] = pose_t.data_ptr() + 64 * pair_idx
            d['aff'] = (aff_t.data_ptr() + 16 * pair_idx) if use_affine else 0
            d['seg_tile_off'] = seg_tile_off.data_ptr() + 4 * lay_wl['sto_off'][:-1]
            d['K_src'], d['K_trg'] = K4_src, K4_trg
            d['N'], d['P'] = Ns, real_points
            d['H'], d['W'] = HW[:, 0], HW[:, 1]
            d['Hl'], d['Wl'] = hl_of[level][:, 0], hl_of[level][:, 1]
            d['tile0'], d['n_tiles'] = lay_wl['s_off'][:-1], np.diff(lay_wl['s_off'])
            d['zmin'] = zmin
            d['rec0'] = rec_per_chunk * lay_wl['c_off'][:-1]
            return d

        mark('work lists staged')
        # (what the FINE-GRAINED work lists of the third attempt's Adam phases are built from, on first use: PairBatch.fine_layout)
        self._fine_src = dict(host={s_: (specs[i][0], specs[i][1], c_host[s_][0]) for s_, i in spec_of.items()}, n_off=n_off, descriptors=descriptors,
                              tile_points=tile_points)
        self._fine = {}
        host = [descriptors(l, self.pix, self.src4[l], self.seg_tile_off, p_off, wl, np.asarray(self.Ps)) for l in full_levels]
        for (l, stride), lay in self.coarse.items():
            c_wl, c_p_off = coarse_host[(l, stride)]
            host.append(descriptors(l, lay.pix, lay.src4, lay.seg_tile_off, c_p_off, c_wl, np.maximum(np.asarray(lay.points), 1)))
        mark('descriptors')
        staged = batch_prepare.stage(host, dev)
        src4_d, seg_tile_off_t = self.src4, self.seg_tile_off
        self.desc = _Lazy(lambda l: batch_prepare.stage([descriptors(l, pix_t, src4_d[l], seg_tile_off_t, p_off, wl, Ps_a)], dev)[0],
                          dict(zip(full_levels, staged)))
        by_stride = {}
        for i, lay in enumerate(self.coarse.values()):
            lay.desc = staged[len(full_levels) + i]
            # (the levels of one lattice share its work list AND its partial buffers: a pair is at one level at a time, and the
            #  scheduled run makes one launch of the phases that share a work list)
            first = by_stride.setdefault(lay.stride, lay)
            if first is lay:
                lay.partials = torch.empty(max(lay.n_spans, 1) * _lib.SP_GN_PARTIAL_FLOATS, dtype=torch.float32, device=dev)
                lay.seg_partials = torch.empty(max(self.rec_per_chunk * lay.n_chunks, 1) * _lib.SP_GN_SEG_FLOATS, dtype=torch.float32, device=dev)
            else:
                lay.partials, lay.seg_partials = first.partials, first.seg_partials

        # optimiser state / workspaces
        self.partials = torch.empty(self.n_spans * _lib.SP_GN_PARTIAL_FLOATS, dtype=torch.float32, device=dev)
        self.seg_partials = torch.empty(self.n_seg_records * _lib.SP_GN_SEG_FLOATS, dtype=torch.float32, device=dev)
        # (one zeroed allocation per type, cut into the per-pair arrays: eight fill launches less per build)
        widths = (1, 2 + 2 * (self.max_N + 8), _lib.SP_LM_STATE_FLOATS, 16 + self.max_N)
        sizes = [(M * w + 3) // 4 * 4 for w in widths]                      # (every array starts on a 16-byte boundary)
        cuts = np.concatenate(([0], np.cumsum(sizes)))
        state_f = torch.zeros(int(cuts[-1]), dtype=torch.float32, device=dev)
        self._costs = state_f[cuts[0]: cuts[0] + M]
        self.adam_state = state_f[cuts[1]: cuts[1] + M * widths[1]].view(M, widths[1])
        self.lm_state = state_f[cuts[2]: cuts[2] + M * widths[2]].view(M, widths[2])
        self.backup = state_f[cuts[3]: cuts[3] + M * widths[3]].view(M, widths[3])
        state_i = torch.zeros(4, M, dtype=torch.int32, device=dev)
        self.arrivals = state_i[0]        # per-pair tile-arrival counters (fused launch)
        self.done = state_i[1]            # per-pair convergence flags (gn_step(conv_tol=...))
        self.phase = state_i[2]           # per-pair position in a device-side schedule (run_scheduled)
        self.phase_iters = state_i[3]
        mark('workspaces')
        self.reset_lm()
        self._graphs = {}
        self._flags_more = []
        self._flag = None
        self._verdict_arrays = None
        self.status = None
        self._initial = (self.pose.clone(), self.kld.clone())
        mark('constructor returns')

    @property
    def setup_bytes(self):
        """{pass: algorithmic bytes} of the set-up that built this batch (bench.py's ``roofline_setup``)."""
        if callable(self._setup_bytes):
            self._setup_bytes = self._setup_bytes()
        return self._setup_bytes

    @property
    def seg_records(self):
        """(pair, segment) owner of every segment record, for host-side consumers (tests, ``evaluate``)."""
        return torch.repeat_interleave(self.chunks[:, :2], self.rec_per_chunk, dim=0)

    def restore_initial(self):
        """Poses, log-depths, affine pairs and optimiser state back to what the batch was built with."""
        self.pose.copy_(self._initial[0])
        self.kld.copy_(self._initial[1])
        if self.aff is not None:
            self.aff.zero_()
        self.adam_state.zero_()
        self.done.zero_()
        self.phase.zero_()
        self.phase_iters.zero_()
        self.reset_lm()

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_synth(cls, pairs, levels=(0, 3), device="cuda:0", **kw):
        from ..image.keyframe import KeyFrame
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in pairs]
        return cls(src, [t(p.trg_image) for p in pairs], [t(p.K) for p in pairs],
                   torch.stack([t(p.pose_init) for p in pairs]), [t(p.kld_init) for p in pairs], levels=levels, **kw)

    def reset_lm(self, lam=1e-4):
        self._lam0 = float(lam)                 # (what a slot of a queue run / a second attempt starts from)
        self.lm_state.zero_()
        self.lm_state[:, 0] = lam
        self.lm_state[:, 1] = -1.0

    # ------------------------------------------------------------------------------------------------
    def cost_pass(self, level, mode, irls_eps=1e-3):
        if mode not in (0, 1) and self.table_flag:
            raise ValueError("cost mode 2 and the developer modes read log-depth tables: build the batch with depth_table=False")
        _lib.check(self.lib.sp_pairs_cost(_lib.ptr(self.desc[level]), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, mode | self.wave_flag | (self.table_flag if mode in (0, 1) else 0),
                                          float(irls_eps), _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.stream_ptr()), "sp_pairs_cost")

    def gn_step(self, level=0, irls_eps=1e-3, lm_up=8.0, lm_down=0.5, lm_min=1e-7, fused=False, conv_tol=0.0):
        """One Gauss-Newton/LM iteration of every pair at pyramid ``level``.  Returns the (M,) device tensor of costs
        (= the reference's residual) evaluated at the parameters BEFORE this step.  Default: two launches (cost
        pass, then one workgroup per pair).  ``fused=True``: a single launch in which the workgroup finishing a
        pair's last tile also solves that pair -- bitwise the same results; measured 0-4 % slower on MI355X (the
        solver's register/LDS footprint costs the cost kernel one wave per SIMD), kept for launch-bound hosts."""
        if fused:
            assert not self.wave_flag and not self.table_flag, "the single-launch form has no wave-span / depth-table variant (PairBatch(depth_table=False))"
            _lib.check(self.lib.sp_pairs_gn_iterate(_lib.ptr(self.desc[level]), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, self.M,
                                                    self.max_N, float(irls_eps), _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.ptr(self.arrivals),
                                                    float(lm_up), float(lm_down), float(lm_min), _lib.ptr(self.lm_state),
                                                    _lib.ptr(self.backup), _lib.ptr(self._costs), _lib.stream_ptr()),
                       "sp_pairs_gn_iterate")
            return self._costs
        if conv_tol > 0.0:
            # per-pair convergence on the device: pairs in self.done are skipped by both launches (``run(conv_tol=...)``)
            _lib.check(self.lib.sp_pairs_cost_active(_lib.ptr(self.desc[level]), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, 1 | self.wave_flag | self.table_flag,
                                                     float(irls_eps), _lib.ptr(self.partials), _lib.ptr(self.seg_partials),
                                                     _lib.ptr(self.done), _lib.stream_ptr()), "sp_pairs_cost_active")
            _lib.check(self.lib.sp_pairs_gn_step_conv(_lib.ptr(self.desc[level]), self.M, self.max_N, _lib.ptr(self.partials),
                                                      _lib.ptr(self.seg_partials), float(lm_up), float(lm_down), float(lm_min),
                                                      _lib.ptr(self.lm_state), _lib.ptr(self.backup), _lib.ptr(self._costs),
                                                      float(conv_tol), _lib.ptr(self.done), _lib.stream_ptr()), "sp_pairs_gn_step_conv")
            return self._costs
        self.cost_pass(level, 1, irls_eps)
        return self.solve_gn(level, lm_up, lm_down, lm_min)

    def solve_gn(self, level=0, lm_up=8.0, lm_down=0.5, lm_min=1e-7):
        """The second launch of a Gauss-Newton iteration: per-pair reduction of the mode-1 partials + Schur-complement LM step."""
        _lib.check(self.lib.sp_pairs_gn_step(_lib.ptr(self.desc[level]), self.M, self.max_N, _lib.ptr(self.partials), _lib.ptr(self.seg_partials),
                                             float(lm_up), float(lm_down), float(lm_min), _lib.ptr(self.lm_state),
                                             _lib.ptr(self.backup), _lib.ptr(self._costs), _lib.stream_ptr()),
                   "sp_pairs_gn_step")
        return self._costs

    def adam_step(self, level=0, lr_kld=1e-3, lr_pose=1e-2, lr_aff=5e-3, fused=False):
        """One Adam iteration (reset-tangent flavour of the reference's tracking/mapping loops) of every pair."""
        if fused:
            assert not self.wave_flag and not self.table_flag, "the single-launch form has no wave-span / depth-table variant (PairBatch(depth_table=False))"
            _lib.check(self.lib.sp_pairs_adam_iterate(_lib.ptr(self.desc[level]), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, self.M,
                                                      self.max_N, _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.ptr(self.arrivals),
                                                      float(lr_kld), float(lr_pose), float(lr_aff), _lib.ptr(self.adam_state),
                                                      _lib.ptr(self._costs), _lib.stream_ptr()), "sp_pairs_adam_iterate")
            return self._costs
        self.cost_pass(level, 0)
        return self.solve_adam(level, lr_kld, lr_pose, lr_aff)

    def solve_adam(self, level=0, lr_kld=1e-3, lr_pose=1e-2, lr_aff=5e-3):
        """The second launch of an Adam iteration: per-pair reduction of the mode-0 partials + Adam / SE(3) retraction."""
        _lib.check(self.lib.sp_pairs_adam_step(_lib.ptr(self.desc[level]), self.M, self.max_N, _lib.ptr(self.partials), _lib.ptr(self.seg_partials),
                                               float(lr_kld), float(lr_pose), float(lr_aff), _lib.ptr(self.adam_state),
                                               _lib.ptr(self._costs), _lib.stream_ptr()), "sp_pairs_adam_step")
        return self._costs

    def evaluate(self, level=0):
        """Residual of every pair at the current parameters (no update): (M,) device tensor."""
        self.cost_pass(level, 0)
        p = self.partials[: self.n_spans * _lib.SP_GRAD_PARTIAL_FLOATS].reshape(self.n_spans, _lib.SP_GRAD_PARTIAL_FLOATS)
        sums = torch.zeros(self.M, dtype=torch.float64, device=self.device).index_add_(0, self.span_pair, p[:, 0].double())
        P = torch.tensor(self.Ps, dtype=torch.float64, device=self.device)
        return (sums / (3.0 * P)).float()

    def graph(self, level, mode="gn", iters=1, **kw):
        """Capture ``iters`` optimiser iterations at ``level`` (2 launches each, no host interaction) into a hipGraph
        and return it; ``g.replay()`` re-runs them.  The graph bakes in device pointers only -- poses, log-depths,
        LM / Adam state all live in device memory -- so it stays valid until the batch is rebuilt."""
        step = self.gn_step if mode == "gn" else self.adam_step
        step(level, **kw)                      # warm-up outside capture (module load, lazy init)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                step(level, **kw)
        return g

    def run_converging(self, max_iters_per_level=25, conv_tol=1e-3, polish_max=15, polish_eps=1e-5, polish_tol=1e-5, check_every=3, predicted_exit=False, **kw):
        """Coarse-to-fine Gauss-Newton with PER-PAIR termination (the batch counterpart of the reference's relative-loss
        early stop, odometery.py:907-915): at every level a pair leaves the iteration as soon as an accepted step lowers its
        cost by less than ``conv_tol`` -- its spans and its solve are skipped from then on -- and the level ends when every
        pair has left (polled every ``check_every`` iterations) or after ``max_iters_per_level``.  The finest level ends with
        up to ``polish_max`` iterations at IRLS epsilon ``polish_eps`` under ``polish_tol``.  Returns the iterations launched
        per phase.  A batch costs the SUM of the iterations its pairs need instead of pairs x the maximum.  (``predicted_exit`` is accepted
        so that a schedule dict can be passed as it is, and ignored: this level-synchronised form ends a pair's level by the evaluated
        test only; the predicted test belongs to the device-side schedule, ``run_scheduled``.)"""
        launched = []
        phases = [(level, max_iters_per_level, kw.get("irls_eps", 1e-3), conv_tol) for level in reversed(self.level_ids)]
        if polish_max > 0:
            phases.append((min(self.level_ids), polish_max, polish_eps, polish_tol))
        for level, n_max, eps, tol in phases:
            self.lm_state[:, 1] = -1.0
            self.lm_state[:, 4] = 0.0
            self.done.zero_()
            it = 0
            while it < n_max:
                for _ in range(min(check_every, n_max - it)):
                    self.gn_step(level, **{**kw, "irls_eps": eps, "conv_tol": tol})
                    it += 1
                if bool(self.done.all()):
                    break
            launched.append(it)
        self.done.zero_()
        return launched

    def schedule(self, max_iters_per_level=25, conv_tol=1e-3, polish_max=15, polish_eps=1e-5, polish_tol=1e-5, irls_eps=1e-3, phases=None,
                 use_coarse=True, pose_first_iters=0, pose_first_eps=None, joint_levels=None, use_levels=None, retry_pose_first=None,
                 retry_phases=None, retry_join=None, depth_damp=None, coarse_damped=None, retry2_phases=None, retry2_join=None,
                 adam_lr_pose=ADAM_LR_POSE, adam_lr_kld=ADAM_LR_KLD, predicted_exit=False):
        """The coarse-to-fine phases of ``run_converging`` as the ``SpSchedule`` of sp_pairs_schedule_* (host memory); levels
        built with a ``point_stride`` run on their decimated point set unless ``use_coarse=False``.  ``phases``: an explicit
        list of dict(level, stride, max_iters, irls_eps, conv_tol) instead (every (level, stride > 1) needs its table:
        ``point_stride`` / ``extra_tables`` of the constructor; optional ``pose_only=True``).  ``pose_first_iters`` > 0 puts a
        POSE-ONLY phase of at most that many iterations in front, at the coarsest level (SP_PHASE_POSE_ONLY, include/sp_hip.h): the
        depths keep their seeds while the pose is aligned -- what makes the schedule converge from the reference's own starting
        distribution (REFERENCE_START_SCHEDULE).  A TUPLE of caps puts one pose-only phase per entry at the coarsest levels in turn
        (coarsest first); ``joint_levels`` = k restricts the joint (pose + depth) phases to the k finest levels: a level above those only
        aligns the pose.  ``depth_damp`` = (d_coarsest, ...): extra LM damping of the log-depth block in the joint phases, coarsest level
        first (include/sp_hip.h SP_PHASE_DEPTH_DAMP; levels not named: none); ``coarse_damped`` = (damping, cap): one more joint phase IN FRONT
        of the joint phases, at the coarsest of their levels, with that depth damping and iteration cap (REFERENCE_START_SCHEDULE).
        ``use_levels`` = k: only the k finest levels of the batch take part in all of that (the batch may carry coarser
        ones for the second attempt).

        THE SECOND ATTEMPT (SpSchedule.retry_entry, SpVerdict): ``retry_phases`` -- a list of phase dicts like ``phases`` -- or its
        shorthand ``retry_pose_first`` = ((level, cap), ...), pose-only phases -- are what a pair that fails its verdict runs FIRST when it
        is put back to its start, before it joins the list above at phase ``retry_join`` (default: the first phase that is not pose-only).
        They sit in front of the list in the SpSchedule; a first attempt enters behind them.

        THE THIRD ATTEMPT (SpSchedule.retry2_entry): ``retry2_phases`` -- phase dicts with ``adam=True`` (SP_PHASE_ADAM: one Adam step of
        the reference's optimiser per iteration at ``adam_lr_pose`` / ``adam_lr_kld`` instead of a Gauss-Newton step; ``max_iters`` is the
        budget, there is no convergence test) -- are what a pair runs when its second attempt fails too, from its initial values again,
        before it joins the list at phase ``retry2_join`` (default: the last phase in front of the polish, i.e. the finest joint phase).
        ``adam=True`` phases may also stand in ``phases`` / ``retry_phases``.  ``predicted_exit`` (SP_PHASE_PREDICTED_EXIT; per phase dict
        or for all phases): a pair leaves a phase right after a step PREDICTED to buy less than the phase's tolerance, without the evaluation
        that confirms it."""
        if phases is None:
            phases = []
            caps = tuple(pose_first_iters) if isinstance(pose_first_iters, (tuple, list)) else ((int(pose_first_iters),) if pose_first_iters > 0 else ())
            coarse_first = sorted(self.level_ids, reverse=True)
            if use_levels is not None:
                coarse_first = coarse_first[len(coarse_first) - int(use_levels):]
            assert len(caps) <= len(coarse_first)
            for level, cap in zip(coarse_first, caps):
                phases.append(dict(level=level, stride=self.point_stride[level] if use_coarse else 1, max_iters=int(cap),
                                   irls_eps=irls_eps if pose_first_eps is None else pose_first_eps, conv_tol=conv_tol, pose_only=True))
            joint = coarse_first if joint_levels is None else coarse_first[len(coarse_first) - int(joint_levels):]
            damps = dict(zip(joint, depth_damp)) if depth_damp else {}              # (coarsest joint level first)
            if coarse_damped == "auto":
                coarse_damped = self.auto_coarse_damping(joint[0], self.point_stride[joint[0]] if use_coarse else 1)
            if coarse_damped:
                phases.append(dict(level=joint[0], stride=self.point_stride[joint[0]] if use_coarse else 1, max_iters=int(coarse_damped[1]), irls_eps=irls_eps,
                                   conv_tol=conv_tol, depth_damp=float(coarse_damped[0])))
            phases += [dict(level=level, stride=self.point_stride[level] if use_coarse else 1, max_iters=max_iters_per_level, irls_eps=irls_eps,
                           conv_tol=conv_tol, depth_damp=damps.get(level, 0.0)) for level in joint]
            finest = min(self.level_ids)
            if polish_max > 0:
                phases.append(dict(level=finest, stride=1, max_iters=polish_max, irls_eps=polish_eps, conv_tol=polish_tol))
            elif use_coarse and self.point_stride[finest] > 1:
                # (with the finest level on a decimated lattice only the polish sees every point -- without one the end state
                #  is the minimiser of a quarter-point lattice, silently)
                raise ValueError("the finest level iterates on a decimated lattice (point_stride > 1): the schedule needs its all-points polish "
                                 "(polish_max > 0), or use_coarse=False")
        if retry_phases is None and retry_pose_first:
            retry_phases = [dict(level=int(l), stride=self.point_stride[int(l)] if use_coarse else 1, max_iters=int(cap),
                                 irls_eps=irls_eps if pose_first_eps is None else pose_first_eps, conv_tol=conv_tol, pose_only=True)
                            for l, cap in retry_pose_first]
        retry_phases = list(retry_phases or [])
        if retry_phases:
            if retry_join is None:
                retry_join = next((i for i, ph in enumerate(phases) if not ph.get("pose_only", False)), 0)
            assert 0 <= int(retry_join) < len(phases)
        retry2_phases = list(retry2_phases or [])
        if retry2_phases:
            if retry2_join is None:
                retry2_join = max(0, len(phases) - 2)
            assert 0 <= int(retry2_join) < len(phases)
        all_phases = retry2_phases + retry_phases + list(phases)
        if len(all_phases) > _lib.SP_MAX_PHASES:
            raise ValueError(f"{len(all_phases)} phases exceed SP_MAX_PHASES = {_lib.SP_MAX_PHASES}")
        sched = _lib.SpSchedule()
        for p, spec in enumerate(all_phases):
            level, stride = int(spec["level"]), int(spec.get("stride", 1))
            ph = sched.phase[p]
            if stride == 1:
                ph.pairs, ph.chunks, ph.spans, ph.n_spans = _lib.ptr(self.desc[level]), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans
                ph.span_partials, ph.seg_partials = _lib.ptr(self.partials), _lib.ptr(self.seg_partials)
            else:
                # (Adam phases -- the third attempt -- run on the lattice's FINE-GRAINED work list: fine_layout)
                lay = self.fine_layout(level, stride) if spec.get("adam", False) and spec.get("fine_spans", True) and hasattr(self, "_fine_src") else self.coarse[(level, stride)]
                ph.pairs, ph.chunks, ph.spans, ph.n_spans = _lib.ptr(lay.desc), _lib.ptr(lay.chunks), _lib.ptr(lay.spans), lay.n_spans
                ph.span_partials, ph.seg_partials = _lib.ptr(lay.partials), _lib.ptr(lay.seg_partials)
            ph.irls_eps, ph.conv_tol, ph.max_iters = float(spec.get("irls_eps", irls_eps)), float(spec["conv_tol"]), int(spec["max_iters"])
            damp = int(round(8.0 * float(spec.get("depth_damp", 0.0))))            # (SP_PHASE_DEPTH_DAMP: eighths, at most 31.875)
            if not 0 <= damp <= 255:
                raise ValueError("depth_damp: 0 .. 31.875 in steps of 1/8")
            ph.flags = ((_lib.SP_PHASE_POSE_ONLY if spec.get("pose_only", False) else 0) | (_lib.SP_PHASE_WAVE_SPANS if self.wave_flag else 0)
                        | (_lib.SP_PHASE_DEPTH_TABLE if self.table_flag else 0) | (damp << _lib.SP_PHASE_DEPTH_DAMP_SHIFT)
                        | (_lib.SP_PHASE_ADAM if spec.get("adam", False) else 0)
                        | (_lib.SP_PHASE_PREDICTED_EXIT if spec.get("predicted_exit", predicted_exit) else 0))
            ph.next = 0
        n2, n1 = len(retry2_phases), len(retry_phases)
        sched.n_phases = len(all_phases)
        sched.entry = n2 + n1
        sched.retry_entry = -1
        sched.retry2_entry = -1
        if retry_phases:
            sched.retry_entry = n2
            join = n2 + n1 + int(retry_join)
            sched.phase[n2 + n1 - 1].next = join if join != n2 + n1 else 0
        if retry2_phases:
            sched.retry2_entry = 0
            sched.phase[n2 - 1].next = n2 + n1 + int(retry2_join)          # (never the following phase unless there is no second attempt and join 0)
            if sched.phase[n2 - 1].next == n2:
                sched.phase[n2 - 1].next = 0
        sched.adam_lr_pose, sched.adam_lr_kld = float(adam_lr_pose), float(adam_lr_kld)
        sched.adam_state = self.adam_state.data_ptr() if any(spec.get("adam", False) for spec in all_phases) else None
        return sched

    FINE_SPAN_POINTS = 512

    def fine_layout(self, level, stride):
        """The FINE-GRAINED work list of lattice ``stride`` at ``level`` (round 6): the same chunks cut into spans of at most FINE_SPAN_POINTS
        points, with its own descriptors (a descriptor carries its pair's span range) and partial records -- for the Adam phases of the
        THIRD ATTEMPT.  A pair in its third attempt iterates 1500 rounds, mostly as the last busy slot of its run; a round then costs the
        LATENCY of its two launches, and the coarse lists' spans -- sized for throughput: 3840 points = 60 trips one after the other in ONE
        wave -- make that ~85 us, spans of 8 trips ~25 us (tools/lone_pair_latency.py).  Built on first use (a schedule with
        ``adam=True`` phases); their launches are skipped while no pair is in such a phase (the library's idle mask)."""
        key = (int(level), int(stride))
        if key in self._fine:
            return self._fine[key]
        lay = self.coarse[key]
        first = next((f for (l2, s2), f in self._fine.items() if s2 == key[1]), None)
        src = self._fine_src
        if first is None:
            pc, seg_pos, c_p_off = src['host'][key[1]]
            # (a span is made of whole chunks, so the chunks are cut to the span size as well: a segment's run -- 1450 points on the
            #  stride-2 lattice of the headline workload, 12 k for the largest SAM mask -- is otherwise the shortest span there is)
            fine = max(self.granule, self.FINE_SPAN_POINTS)
            wl = batch_prepare.work_lists_staged([(pc, seg_pos, src['n_off'], fine)], fine, self.granule, self.device)[0]
        else:
            wl, c_p_off = first.wl, first.c_p_off
        f = _Layout()
        f.stride, f.wl, f.c_p_off = key[1], wl, c_p_off
        f.chunks, f.spans, f.seg_tile_off, f.n_chunks, f.n_spans, f.s_off = wl['chunks'], wl['spans'], wl['seg_tile_off'], wl['n_chunks'], wl['n_spans'], wl['s_off']
        f.desc = batch_prepare.stage([src['descriptors'](key[0], lay.pix, lay.src4, f.seg_tile_off, c_p_off, wl, np.maximum(np.asarray(lay.points), 1))], self.device)[0]
        if first is None:
            f.partials = torch.empty(max(f.n_spans, 1) * _lib.SP_GN_PARTIAL_FLOATS, dtype=torch.float32, device=self.device)
            f.seg_partials = torch.empty(max(self.rec_per_chunk * f.n_chunks, 1) * _lib.SP_GN_SEG_FLOATS, dtype=torch.float32, device=self.device)
        else:
            f.partials, f.seg_partials = first.partials, first.seg_partials
        self._fine[key] = f
        return f

    def auto_coarse_damping(self, level, stride):
        """(damping, iterations) of the damped coarse phase FROM THE SEGMENT STATISTICS of the batch instead of one
        constant: what the damping trades is a depth block that runs ahead of the pose (large segments: hundreds of lattice points each
        pin their depth long before the pose is right -- damp hard) against depths that, held still for a dozen iterations, let the pose
        settle on a biased optimum (many small segments with a handful of lattice points each: damp less).  Measured (DESIGN.md section 6,
        profiles/r05_reference_start_sweep_5_*): 64 segments of ~360 coarse points want 16-31, 1200 segments of ~19 want 12.  Rule: 16
        when the median segment has at least 48 points on the phase's lattice, else 12."""
        pts = np.asarray(self.Ps if stride == 1 else self.coarse[(level, stride)].points, dtype=np.float64)
        per_seg = float(np.median(pts / np.maximum(np.asarray(self.Ns, dtype=np.float64), 1.0)))
        return (16.0, 12) if per_seg >= 48.0 else (12.0, 12)

    def _verdict(self, sched, verdict):
        """The SpVerdict of a scheduled run (host struct; its arrays live on the batch: ``status``, ``diag``, ``attempts``) with the run's
        starting point copied into ``pose0`` / ``kld0``.  ``verdict``: None / True = VERDICT_DEFAULTS, a dict = overrides, False = none."""
        if verdict is False:
            self.status = None
            return None
        opt = dict(VERDICT_DEFAULTS, **(verdict if isinstance(verdict, dict) else {}))
        if getattr(self, "_verdict_arrays", None) is None:
            ints = torch.zeros(2, self.M, dtype=torch.int32, device=self.device)
            self._verdict_arrays = (ints[0], ints[1], torch.zeros(self.M, _lib.SP_DIAG_FLOATS, dtype=torch.float32, device=self.device),
                                    torch.empty_like(self.pose), torch.empty_like(self.kld))
        status, attempts, diag, pose0, kld0 = self._verdict_arrays
        status.zero_(); attempts.zero_(); diag.zero_()
        pose0.copy_(self.pose); kld0.copy_(self.kld)
        v = _lib.SpVerdict()
        v.status, v.diag, v.attempts = status.data_ptr(), diag.data_ptr(), attempts.data_ptr()
        v.pose0, v.kld0, v.pose_base, v.kld_base = pose0.data_ptr(), kld0.data_ptr(), self.pose.data_ptr(), self.kld.data_ptr()
        v.kld_bound, v.cost_bound, v.cost_ratio, v.valid_min = float(opt["kld_bound"]), float(opt["cost_bound"]), float(opt["cost_ratio"]), float(opt["valid_min"])
        v.retry_mask = int(opt["retry_on"]) if (sched.retry_entry >= 0 or sched.retry2_entry >= 0) else 0
        v.lam0 = float(getattr(self, "_lam0", 1e-4))
        v.seg_max_ratio, v.seg_mean_ratio, v.seg_product = float(opt["seg_max_ratio"]), float(opt["seg_mean_ratio"]), float(opt["seg_product"])
        # (diagnostics, ``count_evaluations=True`` in the verdict options: cost evaluations per pair and phase -> ``self.evals`` (M, SP_MAX_PHASES))
        self.evals = None
        if opt.get("count_evaluations", False):
            self.evals = torch.zeros(self.M, _lib.SP_MAX_PHASES, dtype=torch.int32, device=self.device)
            v.evals = self.evals.data_ptr()
        self.status, self.attempts, self.diag = status, attempts, diag
        return v

    def failed(self):
        """(M,) bool device tensor: the pairs whose last scheduled run ended with a failing verdict (after their second attempt, when the
        schedule has one) -- their poses and depths are NOT to be trusted.  ``self.status`` holds the SP_STATUS_* bits, ``self.diag`` the
        numbers behind them (include/sp_hip.h SpVerdict), ``self.attempts`` who was run twice."""
        if getattr(self, "status", None) is None:
            raise RuntimeError("no verdict: run_scheduled(verdict=False), or no scheduled run yet")
        return (self.status & _lib.SP_STATUS_FAILED) != 0

    def run_scheduled(self, check_every=4, lm_up=8.0, lm_down=0.5, lm_min=1e-7, slots=None, verdict=None, return_status=False, streams=1, **schedule_kw):
        """``run_converging`` with the schedule itself on the device: every pair walks through ITS OWN coarse-to-fine phases
        (sp_pairs_schedule_cost / sp_pairs_schedule_gn_step), moving to the next level the moment it converges instead of
        waiting for the slowest pair of the batch, and the host only polls ``min(phase)`` every ``check_every`` iterations.
        Per pair the arithmetic is that of ``run_converging`` with check_every = 1 -- except at levels built with a
        ``point_stride`` > 1, which iterate on their decimated point set.  Returns the iterations launched
        (``return_status=True``: and the (M,) int32 device tensor of SP_STATUS_* bits).

        THE VERDICT (``verdict``: a dict of overrides of VERDICT_DEFAULTS, False = none).  The workgroup that takes a pair out of its last
        phase checks what the pair ended with and writes ``self.status[pair]`` (0 = converged; ``self.failed()``), ``self.diag``; with
        ``retry_pose_first`` / ``retry_phases`` in the schedule a failing pair is put back to its starting point and run ONCE more through
        those phases, in place, during the same run (``self.attempts``).  The reference only asserts finiteness
        (core/dense_optim.py:311,321,340-343); its Adam loop has no notion of "did not converge" either -- but it also does not end in
        the wrong basin where this schedule, one start in a thousand, does.

        ``slots`` < M: SLOT-LEVEL CONTINUOUS BATCHING (include/sp_hip.h SpQueue, sp_pairs_schedule_run_queue).  Only ``slots`` pairs are
        worked on at a time; the solver launch that finishes a pair hands its slot to the next waiting pair of the batch, so every launch
        but the very last ones works on a FULL resident set instead of a thinning one (a scheduled batch otherwise ends in a tail: its
        last 10 % of pairs iterate almost alone for a third of the launches).  The pairs may have DIFFERENT padded layouts (ragged segment
        sets): the cost pass runs over virtual spans, as many per slot as the largest pair has.  Every pair's result is bitwise the one it
        gets with all pairs resident.  ``streams`` = K > 1 (queue runs): the slots are driven as K groups on K HIP streams by K host
        loops that share the one queue -- same results per pair, the groups fill each other's launch gaps and tails."""
        sched = self.schedule(**schedule_kw)
        # (a second attempt can spend the phases in front of the entry as well)
        # (every later attempt can spend the phases in front of the entry and then the list proper once more)
        tail = sum(sched.phase[p].max_iters for p in range(sched.entry, sched.n_phases))
        bound = sum(sched.phase[p].max_iters for p in range(sched.n_phases)) + tail * ((sched.retry_entry >= 0) + (sched.retry2_entry >= 0))
        if self._flag is None:
            self._flag = (torch.zeros(8, dtype=torch.int32, device=self.device), torch.zeros(8, dtype=torch.int32).pin_memory())
        v = self._verdict(sched, verdict)
        v_addr = ctypes.addressof(v) if v is not None else None
        outlier = float(dict(VERDICT_DEFAULTS, **(verdict if isinstance(verdict, dict) else {}))["cost_outlier"]) if v is not None else 0.0
        if sched.adam_state:
            self.adam_state.zero_()
        if slots is not None and int(slots) < self.M:
            it = self._run_queue(sched, int(slots), bound, check_every, lm_up, lm_down, lm_min, v, v_addr, streams=streams, schedule_kw=schedule_kw)
            if outlier > 0.0 and self.M >= COST_OUTLIER_MIN_PAIRS:
                it += self._cost_outlier_pass(outlier, v, bound, check_every, lm_up, lm_down, lm_min, schedule_kw)
            return (it, self.status) if return_status else it
        self.phase.fill_(sched.entry)
        self.phase_iters.zero_()
        self.lm_state[:, 1] = -1.0
        self.lm_state[:, 4:] = 0.0
        # the whole host loop in ONE foreign call (sp_pairs_schedule_run): nothing is issued from Python per iteration, and the
        # interpreter lock is free for the other host threads of a PairStream while this batch runs
        it = self.lib.sp_pairs_schedule_run(ctypes.addressof(sched), self.M, self.max_N, float(lm_up), float(lm_down), float(lm_min),
                                            _lib.ptr(self.lm_state), _lib.ptr(self.backup), _lib.ptr(self._costs), _lib.ptr(self.phase),
                                            _lib.ptr(self.phase_iters), int(check_every), int(bound), _lib.ptr(self._flag[0]),
                                            self._flag[1].data_ptr(), v_addr, _lib.stream_ptr())
        if it < 0:
            _lib.check(it if it > -1000 else -(it + 1000), "sp_pairs_schedule_run")
        if outlier > 0.0 and self.M >= COST_OUTLIER_MIN_PAIRS:
            it += self._cost_outlier_pass(outlier, v, bound, check_every, lm_up, lm_down, lm_min, schedule_kw)
        return (it, self.status) if return_status else it

    def _cost_outlier_pass(self, factor, v, bound, check_every, lm_up, lm_down, lm_min, schedule_kw):
        """The batch-relative part of the verdict (VERDICT_DEFAULTS['cost_outlier']), after the scheduled run proper: pairs that passed
        their own verdict with a final cost above ``factor`` x the batch median get their next attempt too -- put back to their starting
        point and run through the retry phases (the second attempt's, or the third's for a pair that has had its second) in a short
        scheduled run of their own (everybody else is finished: their workgroups return at once) -- and whoever is an outlier after
        its last attempt carries SP_STATUS_COST.  The median is taken over the pairs that FINISHED CLEANLY with a finite cost (ADVICE
        r05: one NaN cost made the median NaN and switched the test off for the whole batch; unfinished pairs pulled it down); fewer
        than COST_OUTLIER_MIN_PAIRS of those: no test.  No second run is issued when nobody retries.  Returns the iterations launched."""
        F, M = _lib.SP_STATUS_FAILED, self.M
        cost = self.diag[:, 0]
        clean = ((self.status & F) == 0) & torch.isfinite(cost) & (cost > 0)
        n_clean = int(clean.sum())
        if n_clean < COST_OUTLIER_MIN_PAIRS:
            return 0
        bound_c = factor * cost[clean].median()
        it = 0
        sched = self.schedule(**schedule_kw)
        entries = [e for e in (sched.retry_entry, sched.retry2_entry) if e >= 0]
        for _ in range(len(entries)):
            out = (self.diag[:, 0] > bound_c) & ((self.status & F) == 0)
            # (an outlier restarts at the attempt after the last one it has made: attempts 0 -> retry_entry, 1 -> retry2_entry)
            nxt = torch.full_like(self.attempts, -1)
            if sched.retry_entry >= 0:
                nxt = torch.where(self.attempts == 0, 1, nxt)
            if sched.retry2_entry >= 0:
                nxt = torch.where((self.attempts == 1) | ((self.attempts == 0) & (nxt < 0)), 2, nxt)
            again = out & (nxt > 0)
            if not bool(again.any()):
                break
            if getattr(self, "_seg_pair", None) is None:
                self._seg_pair = torch.repeat_interleave(torch.arange(M, device=self.device), torch.as_tensor(self.Ns, device=self.device))
            pose0, kld0 = self._verdict_arrays[3], self._verdict_arrays[4]
            self.pose.copy_(torch.where(again[:, None], pose0, self.pose))
            self.kld.copy_(torch.where(again[self._seg_pair], kld0, self.kld))
            self.attempts.copy_(torch.where(again, nxt, self.attempts))
            entry = torch.where(nxt == 1, sched.retry_entry, sched.retry2_entry)
            self.phase.copy_(torch.where(again, entry, sched.n_phases).to(torch.int32))
            self.phase_iters.zero_()
            ls = self.lm_state
            ls[:, 0] = torch.where(again, float(getattr(self, "_lam0", 1e-4)), ls[:, 0])
            ls[:, 1] = torch.where(again, -1.0, ls[:, 1])
            ls[:, 4:] = torch.where(again[:, None], 0.0, ls[:, 4:])
            self.diag[:, 5] = torch.where(again, 0.0, self.diag[:, 5])
            if sched.adam_state:
                self.adam_state.zero_()
            n = self.lib.sp_pairs_schedule_run(ctypes.addressof(sched), M, self.max_N, float(lm_up), float(lm_down), float(lm_min),
                                               _lib.ptr(self.lm_state), _lib.ptr(self.backup), _lib.ptr(self._costs), _lib.ptr(self.phase),
                                               _lib.ptr(self.phase_iters), int(check_every), int(bound), _lib.ptr(self._flag[0]),
                                               self._flag[1].data_ptr(), ctypes.addressof(v), _lib.stream_ptr())
            if n < 0:
                _lib.check(n if n > -1000 else -(n + 1000), "sp_pairs_schedule_run")
            it += n
        out = (self.diag[:, 0] > bound_c) & ((self.status & F) == 0)
        self.status.bitwise_or_(out.to(torch.int32) * _lib.SP_STATUS_COST)
        return it

    def _run_queue(self, sched, slots, bound, check_every, lm_up, lm_down, lm_min, v, v_addr, streams=1, schedule_kw=None):
        assert slots >= 1
        M, dev = self.M, self.device
        rec = ctypes.sizeof(_lib.SpPair)
        lam0 = float(getattr(self, "_lam0", 1e-4))
        # slot descriptors: a private copy of the first `slots` records of every descriptor array the phases use (the solver overwrites a
        # slot's records with the next pair's, whole); the work lists are the batch's -- a descriptor carries its pair's span range, and
        # the cost pass is launched over `max_spans` virtual spans per slot (the largest pair's count: ragged batches share the slots)
        wave = bool(self.wave_flag)
        most = lambda s_off: int(np.diff(np.asarray(s_off)).max())
        spans_of = {_lib.ptr(self.spans).value: most(self._s_off)}
        for lay in list(self.coarse.values()) + list(getattr(self, "_fine", {}).values()):
            spans_of[_lib.ptr(lay.spans).value] = most(lay.s_off)
        # SEVERAL STREAMS (round 6): the slots are cut into `streams` groups, each driven by its own host loop (a thread in
        # sp_pairs_schedule_run_queue: the foreign call releases the interpreter lock) on its own HIP stream, all taking pairs off the ONE
        # queue (`head` is shared; a pair's partial records, unknowns and verdict are its own wherever it runs).  The launches of one group
        # fill the other's launch gaps, poll waits and the tails of its cost passes: a round of ONE group leaves a tenth of the GPU's
        # time unused (profiles/r06_schedule_kernel_stats.csv: 89 % busy)
        n_groups = max(1, min(int(streams), slots))
        cuts = [slots * g // n_groups for g in range(n_groups + 1)]
        head = torch.tensor([slots], dtype=torch.int32, device=dev)
        slot_pair = torch.arange(slots, dtype=torch.int32, device=dev)
        active = torch.zeros(slots, dtype=torch.int32, device=dev)
        q_costs = torch.zeros(M, dtype=torch.float32, device=dev)
        q_lm = torch.zeros(M, _lib.SP_LM_STATE_FLOATS, dtype=torch.float32, device=dev)
        self.phase.fill_(sched.n_phases)            # (the per-slot arrays are the first `slots` entries of the batch's per-pair ones)
        self.phase[:slots] = sched.entry
        self.phase_iters.zero_()
        self.lm_state.zero_()
        self.lm_state[:, 0] = lam0
        self.lm_state[:, 1] = -1.0
        rounds = bound * (-(-M // slots) + 1)
        while len(self._flags_more) < n_groups - 1:
            self._flags_more.append((torch.zeros(8, dtype=torch.int32, device=dev), torch.zeros(8, dtype=torch.int32).pin_memory()))
        flags = [self._flag] + self._flags_more[: n_groups - 1]
        keep, groups = [], []
        for g in range(n_groups):
            lo, n_g = cuts[g], cuts[g + 1] - cuts[g]
            sg = sched if g == 0 else self.schedule(**(schedule_kw or {}))
            q = _lib.SpQueue()
            slot_desc = {}
            for p in range(sg.n_phases):
                ph = sg.phase[p]
                full_ptr = ph.pairs
                if full_ptr not in slot_desc:
                    full = next(t for t in list(self.desc.values()) + [lay.desc for lay in list(self.coarse.values()) + list(getattr(self, "_fine", {}).values())]
                                if t.data_ptr() == full_ptr)
                    slot_desc[full_ptr] = full[lo * rec: (lo + n_g) * rec].clone()
                    keep.append(full)
                q.qpairs[p] = full_ptr
                q.slot_pairs[p] = slot_desc[full_ptr].data_ptr()
                ph.pairs = slot_desc[full_ptr].data_ptr()
                ms = spans_of[ph.spans] if ph.n_spans > 0 else 0
                q.max_spans[p] = (ms + 3) // 4 * 4 if wave else ms
            q.n_queue, q.head, q.slot_pair, q.q_costs, q.q_lm, q.lam0 = M, head.data_ptr(), slot_pair[lo:].data_ptr(), q_costs.data_ptr(), q_lm.data_ptr(), lam0
            q.active = active[lo:].data_ptr()              # (the tail of the run is launched over the busy slots only, include/sp_hip.h SpQueue.active)
            if sg.adam_state:
                sg.adam_state = self.adam_state[lo:].data_ptr()
            keep.append(slot_desc)
            keep.append(active)
            groups.append((sg, q, lo, n_g))

        def drive(g, stream):
            sg, q, lo, n_g = groups[g]
            return self.lib.sp_pairs_schedule_run_queue(ctypes.addressof(sg), ctypes.addressof(q), n_g, self.max_N, float(lm_up), float(lm_down),
                                                        float(lm_min), _lib.ptr(self.lm_state[lo:]), _lib.ptr(self.backup[lo:]), _lib.ptr(self._costs[lo:]),
                                                        _lib.ptr(self.phase[lo:]), _lib.ptr(self.phase_iters[lo:]), int(check_every), int(rounds),
                                                        _lib.ptr(flags[g][0]), flags[g][1].data_ptr(), v_addr, stream)

        if n_groups == 1:
            its = [drive(0, _lib.stream_ptr())]
        else:
            import threading
            main = torch.cuda.current_stream()
            # (ONE pool of side streams per device for the whole process: a stream per batch made hundreds of them over a long run, and
            #  HIP maps streams onto a handful of hardware queues)
            pool = _SIDE_STREAMS.setdefault(dev, [])
            while len(pool) < n_groups - 1:
                pool.append(torch.cuda.Stream(device=dev))
            side = pool[: n_groups - 1]
            its = [0] * n_groups
            for st in side:
                st.wait_stream(main)

            def work(g):
                torch.cuda.set_device(dev)             # (a new host thread starts on device 0)
                its[g] = drive(g, ctypes.c_void_p(side[g - 1].cuda_stream))

            threads = [threading.Thread(target=work, args=(g,)) for g in range(1, n_groups)]
            for t in threads:
                t.start()
            its[0] = drive(0, _lib.stream_ptr())
            for t in threads:
                t.join()
            for st in side:
                main.wait_stream(st)
        for it in its:
            if it < 0:
                _lib.check(it if it > -1000 else -(it + 1000), "sp_pairs_schedule_run_queue")
        it = max(its)
        # results per PAIR (what the per-slot arrays hold is whichever pairs came last)
        min_phase = min(int(f[1][0]) for f in flags)
        taken = min(max(int(f[1][1]) for f in flags), M)
        self._costs.copy_(q_costs)
        self.lm_state.copy_(q_lm)
        self._queue_stats = dict(slots=slots, head=taken, slot_pair=slot_pair, finished=min_phase >= sched.n_phases, rounds_by_stream=list(its))
        if min_phase < sched.n_phases:
            #  the run hit its round limit: pairs still in a slot carry SP_STATUS_UNFINISHED (set by the library), pairs that never
            # got a slot are marked here; without a verdict there is no way to tell the caller per pair, so that is an error
            if v is None:
                raise RuntimeError(f"run_scheduled(slots={slots}) ended on its round limit ({rounds}) with pairs unfinished")
            self.status[taken:] = _lib.SP_STATUS_UNFINISHED
        self.phase.fill_(sched.n_phases)
        return it

    def run(self, iters_per_level, mode="gn", use_graph=False, polish_iters=0, polish_eps=1e-5, **kw):
        """Coarse-to-fine schedule like ``two_frame_sfm.py:150-155``: ``iters_per_level`` iterations at each level (an int, or
        one count per level, coarse to fine).
        ``use_graph``: replay one captured iteration per level instead of issuing 2 launches per iteration from
        Python (the captured warm-up iteration counts towards ``iters_per_level``).
        ``polish_iters`` (Gauss-Newton only): that many more iterations at the finest level with the IRLS epsilon at
        ``polish_eps`` -- the default epsilon (1e-3) smooths |r| like a Huber kernel and leaves the fixed point about
        1e-4 rad / 2e-3 log-depth from the minimiser of the reference's L1 cost; 30-40 iterations at 1e-5 close that gap
        to a few 1e-6 (golden g12)."""
        self._run_levels(iters_per_level, mode, use_graph, **kw)
        if mode == "gn" and polish_iters > 0:
            finest = min(self.level_ids)
            for _ in range(polish_iters):
                self.gn_step(finest, **{**kw, "irls_eps": polish_eps})

    def _run_levels(self, iters_per_level, mode, use_graph, **kw):
        per_level = isinstance(iters_per_level, (tuple, list))          # coarse -> fine when given per level
        if per_level:
            assert len(iters_per_level) == len(self.level_ids)
        for li, level in enumerate(reversed(self.level_ids)):
            n_iters = iters_per_level[li] if per_level else iters_per_level
            if mode == "gn":
                self.lm_state[:, 1] = -1.0      # costs of different levels are not comparable
                self.lm_state[:, 4] = 0.0
            if use_graph:
                key = (level, mode, tuple(sorted(kw.items())))
                g = self._graphs.get(key)
                done = 0
                if g is None:
                    g = self._graphs[key] = self.graph(level, mode, 1, **kw)
                    done = 1                     # graph() ran one real warm-up iteration; capturing executes nothing
                for _ in range(n_iters - done):
                    g.replay()
            else:
                for _ in range(n_iters):
                    (self.gn_step if mode == "gn" else self.adam_step)(level, **kw)

    # ------------------------------------------------------------------------------------------------
    def costs(self):
        return self._costs

    def poses(self):
        return self.pose.reshape(self.M, 4, 4)

    def klds(self):
        return [self.kld[self.n_off[m]: self.n_off[m + 1]] for m in range(self.M)]

    def schedule_traffic(self, **schedule_kw):
        """ALGORITHMIC bytes of the last scheduled run (``run_scheduled(verdict=dict(count_evaluations=True), **schedule_kw)``): every
        cost evaluation of every pair in every phase (``self.evals``) priced like the level-0 pass is: 20 B per point of
        the phase's point set + 12 B per pixel of the phase's target level.  Returns dict(total_bytes, evaluations, phases=[dict(phase, level,
        stride, kind, evaluations, bytes)])."""
        assert getattr(self, "evals", None) is not None, "run_scheduled(verdict=dict(count_evaluations=True)) first"
        sched = self.schedule(**schedule_kw)
        ev = self.evals.cpu().numpy().astype(np.float64)
        where = {self.desc[l].data_ptr(): (l, 1) for l in list(self.desc.keys())}
        where.update({lay.desc.data_ptr(): key for key, lay in self.coarse.items()})
        where.update({lay.desc.data_ptr(): key for key, lay in getattr(self, "_fine", {}).items()})
        phases, total = [], 0.0
        for p in range(sched.n_phases):
            level, stride = where[sched.phase[p].pairs]
            pts = np.asarray(self.Ps if stride == 1 else self.coarse[(level, stride)].points, dtype=np.float64)
            hw = np.asarray(self.level_hw[level], dtype=np.float64)
            per_pair = 20.0 * pts + 12.0 * hw[:, 0] * hw[:, 1]
            fl = sched.phase[p].flags
            kind = ("adam" if fl & _lib.SP_PHASE_ADAM else "pose-only" if fl & _lib.SP_PHASE_POSE_ONLY else
                    "damped" if (fl >> _lib.SP_PHASE_DEPTH_DAMP_SHIFT) & 0xff else "joint")
            if p == sched.n_phases - 1 and stride == 1:
                kind = "polish"
            nbytes = float((ev[:, p] * per_pair).sum())
            total += nbytes
            phases.append(dict(phase=p, level=int(level), stride=int(stride), kind=kind, attempt=("third" if p < max(sched.retry_entry, 0) and sched.retry2_entry >= 0 else
                                                                                                   "second" if p < sched.entry else "first"),
                               evaluations=int(ev[:, p].sum()), bytes=nbytes, bytes_per_evaluation=float(per_pair.mean())))
        return dict(total_bytes=total, evaluations=int(ev.sum()), phases=phases)

    def algorithmic_bytes(self, level):
        """20 B per segment pixel + 12 B per target-image pixel, per pair per iteration."""
        return sum(20 * P + 12 * h * w for P, (h, w) in zip(self.Ps, self.level_hw[level]))
