/*
 * sp_hip.h -- C ABI of libsp_hip.so: the MI355X (gfx950) implementation of SuperPrimitive's
 * per-segment photometric pose-and-depth optimisation hot path.
 *
 * The reference (makezur/super_primitive) is 100 % Python: the interface this library sits behind is a set
 * of module-level Python functions, not an FFI.  Each entry point below names the reference function(s)
 * whose arithmetic it replaces (paths relative to the reference tree); the Python wrappers in
 * super_primitive_amd/ keep those functions' names and signatures and call these symbols through ctypes.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing here synchronises the host;
 *   - every function returns 0 on success, a hipError_t (>0) from the runtime, or a negative SP_E* code;
 *   - all arithmetic is fp32 (fp64 only inside the small per-pair finalise / solve kernels);
 *   - the library allocates nothing: workspaces are caller-owned (sizes via sp_*_workspace_floats()).
 *
 * Data layout of a compacted source keyframe ("segment table", built once per keyframe):
 *   pix[P]      uint32  bit31 = source-pixel validity, bits 30..16 = row, bits 15..0 = col; points ordered
 *                       (segment, row, col) exactly like torch.where(keypoint_regions) in
 *                       core/dense_optim.py:103
 *   seg_off[N+1] int32  CSR offsets of the segments into pix / src4
 *   src4[P]     float4  {I_src.r, I_src.g, I_src.b, L}: the source image of ONE pyramid level sampled at the
 *                       point's own pixel (core/dense_optim.py:315-317) and the base log-depth
 *                       logdepth_perseg[n,row,col]
 *   kp_L[N]     float   logdepth_perseg[n, kp_row, kp_col] (core/dense_optim.py:51-64)
 *   tiles[T]    int32x4 {pair, segment, first point, point count}; a tile never straddles two segments
 *   seg_tile_off[N+1]   CSR offsets of the segments into tiles
 * Target images are packed HWC3 (r,g,b adjacent, 12 bytes per texel) per pyramid level so one bilinear tap is one
 * 12-byte load and a level costs exactly the algorithmic 12 B per pixel of HBM traffic.
 */
#ifndef SP_HIP_H
#define SP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_ABI_VERSION 15

#define SP_EINVAL (-1)   /* bad argument (null pointer, non-positive size, ...) */
#define SP_ELIMIT (-2)   /* size outside what the kernels support (H or W > 32767, N > 65535, ...) */

/* floats per partial record: mode 0 -- one tile of sp_photo_cost_grad (workspace = n_tiles * B * 16) or one span of
 * sp_pairs_cost; mode 1 -- one span of sp_pairs_cost */
#define SP_GRAD_PARTIAL_FLOATS 16
#define SP_GN_PARTIAL_FLOATS   32
#define SP_GNA_PARTIAL_FLOATS  48   /* mode 2: Gauss-Newton with the affine brightness pair among the unknowns */

int sp_abi_version(void);

/* ------------------------------------------------------------------------------------------------------
 * Segment table construction (replaces the per-iteration torch.where / dense (N,H,W) passes of
 * core/dense_optim.py:38-114 by a once-per-keyframe compaction).
 * ---------------------------------------------------------------------------------------------------- */

/* Pass 1: row_counts[N*H] = exclusive per-segment scan of the per-row mask counts (scratch kept for pass 2),
 * counts[N] = pixels per segment, seg_off[N+1] = exclusive scan of counts (seg_off[N] = P, which the host
 * reads once to size the table).  masks: N*H*W bytes (torch.bool). */
int sp_mask_count(const uint8_t* masks, int N, int H, int W, int32_t* row_counts, int32_t* counts,
                  int32_t* seg_off, void* stream);

/* Pass 2: fill pix (row/col, validity bit cleared), base log-depth baseL[P] and kp_L[N].  row_off = the
 * row_counts array produced by sp_mask_count.  keypoints: (N,2) normalised (row,col) as in
 * image/keyframe.py:20-75; the keypoint pixel is round_half_even(0.5*(dim-1)*(kp+1))
 * (tool/point_utils.py:37-40).  Order inside a segment is row-major, like torch.where. */
int sp_table_fill(const uint8_t* masks, const float* logdepth, const float* keypoints, int N, int H, int W,
                  const int32_t* seg_off, const int32_t* row_off, uint32_t* pix, float* baseL, float* kp_L,
                  void* stream);

/* Sample one source pyramid level at every table point and pack {rgb, L}.  set_validity != 0: also compute the
 * source validity bit of pix with the reference's formula (core/dense_optim.py:143-162 applied to the source
 * frame, :315-317); it depends on the geometry grid only (not on the level), so it is set by the first sampling
 * after the table is built and pix is read-only afterwards.  img: planar (3,Hl,Wl) f32.  K: 9 floats row-major. */
int sp_table_sample_source(uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                           const float* kld, int N, int P, int H, int W, const float* img, int Hl, int Wl,
                           const float* K, float* src4, int set_validity, void* stream);

/* planar (B,3,H,W) f32 -> packed (B,H,W,3) f32 */
int sp_pack_rgb(const float* chw, int B, int H, int W, float* hwc3, void* stream);

/* One pyramid step of image/gaussian_pyramid.py:53-85: reflect-pad 1, 3x3 binomial /16, keep even rows and
 * columns.  in: planar (C,H,W); out: planar (C,ceil(H/2),ceil(W/2)). */
int sp_blur_decimate(const float* in, int C, int H, int W, float* out, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Batched preparation: the table / pyramid / sampling passes above for MANY keyframes per launch (job records in DEVICE
 * memory, one grid row per record).  PairBatch builds hundreds of frame pairs with ~12 launches and ONE host
 * synchronisation (the per-segment counts, which size the tables) instead of ~15 launches and a synchronisation per pair.
 * A "table" is the compacted point set of one keyframe on a pixel lattice: stride 1 = every mask pixel (the tables above),
 * stride s > 1 = the mask pixels whose row and column are multiples of s (coarse levels of per-pair schedules).
 *   sp_prepare_count : row_counts (scratch, N*H) and counts[N] of every lattice of every keyframe, and the masks as packed bit
 *                      words (SpPrepTable.bits)                                                     [masks read once]
 *   -- host: reads counts, lays the segments out (runs padded to multiples of 256), allocates pix / baseL, uploads seg_off --
 *   sp_prepare_fill  : pix / baseL at seg_off[n] + rank inside the segment; kp_L[N] where kp_L != NULL
 *   sp_prepare_blur  : one pyramid step (sp_blur_decimate) of every job image
 *   sp_prepare_pack  : planar (3,H,W) -> HWC3 of every job image
 *   sp_prepare_blur_pack: both at once for three-channel images (the target frames)
 *   sp_prepare_gather: n small float vectors, one DEVICE pointer each (src[i], off[i + 1] - off[i] floats), into one flat array
 *                      (out + off[i]); src and off are device arrays.  What the host side otherwise does with one torch.cat /
 *                      torch.stack over hundreds of per-pair tensors (the intrinsics and initial log-depths of a batch: 0.7 us of
 *                      interpreter time per tensor); no counterpart in the reference, which optimises one pair per call
 *   sp_prepare_sample: sp_table_sample_source of every job at all its levels, and the source-validity bit of pix.  Every
 *                      segment's run must start at a multiple of 256 (the padded layout of the many-pairs work list; of 64 when
 *                      SpPrepSample.granule is 64);
 *                      positions beyond counts[n] of a run are padding and are written as {pix 0, src4 0} = invalid points,
 *                      so pix / src4 need no prior clearing
 * ---------------------------------------------------------------------------------------------------- */
#define SP_PREP_MAX_STRIDES 4
#define SP_PREP_MAX_LEVELS 4
typedef struct SpPrepTable {     /* one keyframe: its masks are read once per pass for all its lattices */
    const uint8_t* masks;        /* (N,H,W) bool */
    const float* logdepth;       /* (N,H,W) */
    const float* keypoints;      /* (N,2), used when kp_L != NULL */
    float* kp_L;                 /* N or NULL, out of sp_prepare_fill */
    int32_t* row_counts[SP_PREP_MAX_STRIDES];   /* N*H scratch each: written by sp_prepare_count, read by sp_prepare_fill */
    int32_t* counts[SP_PREP_MAX_STRIDES];       /* N each, out of sp_prepare_count */
    const int32_t* seg_off[SP_PREP_MAX_STRIDES];/* N each: first table position of every segment (host-made, padded layout) */
    uint32_t* pix[SP_PREP_MAX_STRIDES];         /* out of sp_prepare_fill */
    float* baseL[SP_PREP_MAX_STRIDES];          /* out of sp_prepare_fill: the log-depth of every table point -- or all NULL (ABI 13): no copy is made, the
                                                   sampling pass reads the keyframe's dense array instead (SP_PREP_DENSE_L): 8 bytes per lattice point of
                                                   writes and re-reads less, and the fill pass loses its log-depth loads */
    int32_t stride[SP_PREP_MAX_STRIDES];
    int32_t N, H, W, n_strides;
    uint32_t* bits;              /* N*H*(W/16) words or NULL: the masks as one bit word per 16 pixels, written by sp_prepare_count on
                                    its fast path (W a multiple of 16 up to 1024, masks 16-byte aligned, strides in {1,2,4,8,16}) and
                                    read by sp_prepare_fill INSTEAD of the masks (4 instead of 16 bytes per 16 pixels); NULL, or a
                                    keyframe off the fast path: the fill pass reads the masks.  logdepth must be 16-byte aligned
                                    when bits is given.  The words of rows WITHOUT a set pixel are left unwritten when stride[0] is 1
                                    (the fill pass finds such rows empty through their counts and never reads their words): scratch
                                    between the two passes, not an output */
    const int32_t* boxes;        /* NULL, or N x {row0, col0, row1, col1} (half open, pixels): a HINT from whoever made the masks -- SAM's
                                    frontend computes and NMS-filters exactly these (frontend/segment/mask_generation.py:93,155-180) --
                                    that segment n has no set pixel outside its box.  The count pass then reads the masks inside
                                    the boxes only (the 16-pixel pieces that meet them), ~1 MB of a 640x480x64 keyframe's 19.7 MB;
                                    boxes are clamped to the image, an empty or inverted box is an empty segment.  A mask pixel that
                                    breaks the contract is not seen when its ROW is outside the box or its 16-pixel piece does not
                                    meet the box columns, and seen otherwise (deterministic either way).  Fast path only (see
                                    bits); ignored on the general path, which scans everything */
} SpPrepTable;                   /* 240 bytes */
typedef struct SpPrepSample {    /* one table, sampled at up to SP_PREP_MAX_LEVELS pyramid levels in one pass; sets pix bit 31 */
    uint32_t* pix;
    const float* baseL;          /* [P] the table's own log-depths; under SP_PREP_DENSE_L: the keyframe's dense (N,H,W) array (N H W < 2^32) */
    const int32_t* seg_off;      /* N, relative to pix */
    const int32_t* counts;       /* N: real points of every segment */
    const float* kp_L;
    const float* kld;
    const float* K;              /* 9 floats, full-resolution intrinsics */
    const float* image[SP_PREP_MAX_LEVELS];     /* planar (3,Hl,Wl) source image of each level */
    float* src4[SP_PREP_MAX_LEVELS];            /* (P,4) out, one per level */
    int32_t Hl[SP_PREP_MAX_LEVELS], Wl[SP_PREP_MAX_LEVELS];
    int32_t N, P, H, W, n_levels;
    int32_t granule;             /* low 16 bits: padding granule of the runs: 0 / 256, or 64 for wave-span tables (SP_COST_WAVE_SPANS);
                                    | SP_PREP_DEPTH_TABLE: write exp(L) instead of L into src4.w (SP_COST_DEPTH_TABLE)
                                    | SP_PREP_DENSE_L: baseL is the dense (N,H,W) array, a point's log-depth sits at (its segment, row, column) */
} SpPrepSample;                  /* 176 bytes */
#define SP_PREP_DEPTH_TABLE 0x10000
#define SP_PREP_DENSE_L 0x20000
typedef struct SpPrepImage {
    const float* in;             /* (C,H,W) planar */
    float* out;                  /* blur: (C,ceil(H/2),ceil(W/2)); pack: (H,W,3) */
    int32_t H, W;
} SpPrepImage;                   /* 24 bytes */
int sp_prepare_count(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream);
/* (ABI 15) sp_prepare_count for batches whose keyframes ALL carry the segment-box hint (SpPrepTable.boxes; mask_generation.py:93,155-180 computes
 * the boxes of SAM's masks): same outputs, bit for bit; a workgroup tests 64 blocks of 64 rows against the boxes at once and only walks those
 * that meet one (the pass is otherwise bound by the latency of 184 k two-load chains per 384 keyframes).  Keyframes without boxes are handled
 * too, correctly but slowly. */
int sp_prepare_count_boxed(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream);
int sp_prepare_fill(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream);
int sp_prepare_sample(const SpPrepSample* jobs, int n_jobs, int max_P, void* stream);
/* (ABI 15) sp_prepare_sample for TWO tables of every keyframe in one launch: jobs_a[i] and jobs_b[i] (i < n_jobs) belong to the same keyframe and
 * take consecutive workgroups, so that a source image both tables gather from (the all-points table and the stride-2 lattice, both sampled at
 * level 0) is fetched from memory once.  Same outputs as two sp_prepare_sample calls, bit for bit. */
int sp_prepare_sample_pairs(const SpPrepSample* jobs_a, int max_P_a, const SpPrepSample* jobs_b, int max_P_b, int n_jobs, void* stream);
int sp_prepare_blur(const SpPrepImage* jobs, int n_jobs, int C, int max_out_pixels, void* stream);
int sp_prepare_pack(const SpPrepImage* jobs, int n_jobs, int max_pixels, void* stream);
/* One pyramid step of THREE-CHANNEL images together with their packed forms (ABI 14): what sp_prepare_blur followed by sp_prepare_pack
 * computes, bit for bit, without the packing pass's re-read of the planar levels.  out (planar level l + 1), packed_out (level l + 1 as
 * HWC3) and packed_in (level l as HWC3) may each be NULL. */
typedef struct SpPrepImagePack {
    const float* in;             /* (3,H,W) planar, level l */
    float* out;
    float* packed_in;
    float* packed_out;
    int32_t H, W;
} SpPrepImagePack;               /* 40 bytes */
int sp_prepare_blur_pack(const SpPrepImagePack* jobs, int n_jobs, int max_out_pixels, void* stream);
int sp_prepare_gather(const float* const* src, const long long* off, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * The cost: one source keyframe against B target frames (B = 1: core/dense_optim.py:265-363
 * photomeric_cost and :365-403 photomeric_cost_precomputed; B >= 1: core/dense_optim_batch.py:50-147).
 * One fused pass computes residual[b] = mean_{ch,pt} |(I_src - I_trg')*mask| AND its first-order information.
 *
 *   mode 0 (gradient): d residual[b] / d {kld (N), pose rows 0..2 (3x4), affine (a_s,b_s,a_t,b_t)}
 *                      -- what autograd produces in the reference's Adam loops;
 *   mode 1 (Gauss-Newton): IRLS-weighted normal equations of the same residuals over the left SE(3) tangent
 *                      [tau,phi] and the per-segment log-depths (arrow structure); see sp_gn_* below.
 *
 * K_src: 9 floats. K_trg: B*9. pose: B*16 row-major (target <- source). aff_src: 2 floats or NULL,
 * aff_trg: B*2 or NULL (both or neither).  zmin: 1e-7 for the single-target functions, 1e-6 for the batch
 * function (dense_optim.py:146 vs dense_optim_batch.py:15).
 * Outputs (mode 0): residual[B], g_kld[B*N], g_pose[B*16] (row 3 = 0), g_aff[B*4].
 * ---------------------------------------------------------------------------------------------------- */
int sp_photo_cost_grad(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L,
                       const int32_t* tiles, const int32_t* seg_tile_off, int n_tiles, int N, int P, int H, int W,
                       const float* K_src, const float* kld, const float* trg3, int Hl, int Wl,
                       const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                       float zmin, float* workspace, float* residual, float* g_kld, float* g_pose, float* g_aff,
                       void* stream);

/* The same cost for an EXPLICIT list of source points -- core/dense_optim.py:365-403 photomeric_cost_precomputed fed
 * with a dict that did not come from this library's table (built by hand, filtered, or round-tripped through
 * tool/etc.py dict_cpu / a queue / a checkpoint): xyz (n,3) = src_pts and rgb (n,3) = src_pixels^T of the points whose
 * src_valid_mask is set (the caller compacts; invalid source points contribute exact zeros in the reference), 24 bytes
 * per point (tracking variant).  P_total = the ORIGINAL number of points: the residual is a mean
 * over 3 * P_total values (core/dense_optim.py:249-253).  No log-depth unknowns: outputs residual[B], g_pose[B*16],
 * g_aff[B*4].  workspace: sp_points_workspace_floats(n_points, B) floats. */
int sp_points_workspace_floats(int n_points, int B);
int sp_points_cost_grad(const float* xyz, const float* rgb, int n_points, int P_total, int H, int W, const float* trg3,
                        int Hl, int Wl, const float* K_trg, const float* pose, int B, const float* aff_src,
                        const float* aff_trg, float zmin, float* workspace, float* residual, float* g_pose, float* g_aff,
                        void* stream);

/* Per-point diagnostics of the same pass (collect_stats > 0 in the reference, core/dense_optim.py:347-361).
 * Any output pointer may be NULL.  Only every stride-th table point is reported (stride 1 = the reference's
 * tensors; > 1 = the down-sampled channel for a visualiser): with Q = ceil(P / stride) rows the
 * shapes are src_pts (Q,3); trg_pts (B,Q,3); src_rgb (3,Q); trg_rgb (B,3,Q) after brightness compensation;
 * raw (B,3,Q) = (I_src - I_trg')*mask; src_valid (Q) u8; trg_valid (B,Q) u8; seg_ids (Q) int64. */
int sp_photo_stats(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L, int N, int P,
                   int H, int W, const float* K_src, const float* kld, const float* trg3, int Hl, int Wl,
                   const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                   float zmin, float* src_pts, float* trg_pts, float* src_rgb, float* trg_rgb, float* raw,
                   uint8_t* src_valid, uint8_t* trg_valid, int64_t* seg_ids, int stride, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Many independent frame pairs per launch (BASELINE.json configs 2 and 5; the throughput path).  Each pair
 * has its own segment table, target image, intrinsics, pose and log-depths, described by one SpPair record
 * in device memory.  No host synchronisation anywhere: one optimiser iteration is sp_pairs_cost + one
 * sp_pairs_*_step launch and can be captured in a hipGraph.
 *
 * Work list.  The tables of this path are PADDED: every segment's run of points in pix / src4 is extended to a
 * multiple of 256 with invalid points (pix word 0, src4 zeros; they contribute exact zeros).
 *   chunks[C] int32x4 {pair, segment, start, count}: a run of `count` (multiple of 256) points of one segment,
 *             `start` = index of its first point in the pair's padded arrays; chunks of a pair are consecutive
 *             and contiguous in memory (segment-major, like the table);
 *   spans[S]  int32x4 {first chunk, number of chunks, points, pair}: what one workgroup processes -- a run of
 *             consecutive chunks of one pair; pair-level sums are reduced once per span, segment-level sums once
 *             per chunk and wave.
 * Partials, two arrays:
 *   span_partials[S]      one record of (SP_GRAD_PARTIAL_FLOATS | SP_GN_PARTIAL_FLOATS) floats per span: the sums that
 *                         belong to the PAIR (mode 0: as sp_photo_cost_grad's tile record with column 13 unused;
 *                         mode 1: [0] sum|r|, [1..21] H_pp upper triangle, [22..27] b_p, [28] valid points);
 *   seg_partials[4 * C]   one record of (SP_GRAD_SEG_FLOATS | SP_GN_SEG_FLOATS) floats per (chunk, wave), record
 *                         4 * chunk + wave: the sums that belong to the SEGMENT (mode 0: d/dkld; mode 1: h_pd(6), D, b_d, and -- ABI 13 --
 *                         [8] the record's own sum |r|, [9] its valid points: the cost PER SEGMENT, what the verdict's within-pair
 *                         test reads; [10], [11] unused).
 * The solvers sum records in index order.  Per pair they need its span range (SpPair.tile0, n_tiles), the index of
 * its first segment record (rec0) and, per segment, the record range of its chunks (seg_tile_off, relative to rec0).
 * ---------------------------------------------------------------------------------------------------- */
#define SP_GRAD_SEG_FLOATS 1
#define SP_GN_SEG_FLOATS 12
#define SP_GNA_SEG_FLOATS 12
typedef struct SpPair {
    const uint32_t* pix;      /* [P_padded] */
    const float*    src4;     /* [P_padded*4] for the level being optimised */
    const float*    kp_L;     /* [N] */
    const float*    trg3;     /* [Hl*Wl*3] packed HWC3, for the level being optimised */
    float*          kld;      /* [N]  optimisation variable */
    float*          pose;     /* [16] optimisation variable (target <- source) */
    float*          aff;      /* [4]  {a_s,b_s,a_t,b_t} or NULL */
    const int32_t*  seg_tile_off; /* [N+1] offsets into this pair's segment records, relative to rec0 */
    float K_src[4];           /* fx fy cx cy */
    float K_trg[4];
    int32_t N, P, H, W, Hl, Wl;   /* P = number of REAL points (the cost is a mean over 3 P values) */
    int32_t tile0;            /* first span of this pair */
    int32_t n_tiles;          /* number of spans (workgroups) of this pair */
    float zmin;
    int32_t rec0;             /* first segment record of this pair (= 4 * its first chunk) */
} SpPair;

/* HOST helpers (no device work): the work list above for a whole batch.  pc[s] / seg_pos[s]: padded run length and pair-relative
 * position of every segment (all pairs concatenated), n_off[m]: first segment of pair m (n_pairs + 1 entries).  Chunks: pieces of at
 * most tile_points points (granule-aligned, nearly equal) of every segment; spans: greedy runs of consecutive chunks of one pair up
 * to span_points points.  sp_host_work_list_chunks returns the number of chunks (sizes the arrays: chunks and spans 4 int32 per
 * entry, spans at most as many as chunks); sp_host_work_list fills chunks, spans, seg_tile_off (per pair N + 1 record offsets, 4
 * records per chunk; sum N + n_pairs entries, pair m at sto_off[m]), c_off / s_off (first chunk / span of every pair) and returns
 * the number of spans. */
int sp_host_work_list_chunks(const long long* pc, int n_segs, int tile_points, int granule);
/* The padded layout of every lattice of a batch from the per-segment point counts (counts[l * n_segs + s], as sp_prepare_count leaves
 * them): pc = the count rounded up to the granule, seg_pos = the segment's first point relative to its pair's table, p_off[l * (n_pairs +
 * 1) + m] = first point of pair m in the lattice's flat arrays, seg_off[l * 2 n_segs ...] = the segment positions as int32, first inside
 * the flat array (sp_prepare_fill), then relative to the pair (sp_prepare_sample, SpPair.seg_tile_off's companion), points[l * n_pairs +
 * m] = real points of pair m.  One pass where the numpy form takes a dozen array operations per lattice. */
int sp_host_layout(const int32_t* counts, int n_lattices, int n_segs, const long long* n_off, int n_pairs, int granule,
                   long long* pc, long long* seg_pos, long long* p_off, int32_t* seg_off, long long* points);
int sp_host_work_list(const long long* pc, const long long* seg_pos, const long long* n_off, int n_pairs, int span_points,
                      int tile_points, int granule /* 256, or 64 for wave spans */, int records_per_chunk /* 4, or 1 for wave spans */,
                      int32_t* chunks, int32_t* spans, int32_t* seg_tile_off, long long* sto_off, long long* c_off, long long* s_off);

/* WAVE SPANS (mode | SP_COST_WAVE_SPANS, modes 0 and 1): a work list whose spans belong to single WAVES instead of workgroups -- the
 * tables are padded to multiples of 64 points instead of 256, chunk counts are multiples of 64, and there is ONE segment record per
 * chunk (record index = chunk index) instead of four.  For batches of many small ragged segments (SAM-like masks): at 1200
 * segments of ~280 pixels the 256-point granule pads 40 %, the 64-point granule 11 %.  Four consecutive spans share a workgroup. */
#define SP_COST_WAVE_SPANS 0x100
/* DEPTH TABLES (mode | SP_COST_DEPTH_TABLE, modes 0 and 1, with or without wave spans): src4.w of the tables holds exp(L) -- the
 * source depth at the segment's seed -- instead of L, and a point's depth is src4.w * exp(kld - kp_L): the exponential is taken once
 * per chunk (a scalar) instead of once per point (v_exp_f32 is a quarter-rate transcendental; -1.8 % kernel time on the headline
 * workload, profiles/r04_kernel_experiments.txt).  Same value up to 2 ulp of the product.  sp_prepare_sample writes such tables with
 * SP_PREP_DEPTH_TABLE. */
#define SP_COST_DEPTH_TABLE 0x200

/* mode 0 / 1 as above.  mode 2 = mode 1 plus the affine brightness pair of the TARGET frame as two more unknowns (a_t, b_t; the
 * source frame's pair enters with the opposite sign): residual columns j_a = gain * I_trg(sample), j_b = -1.  Span record
 * (SP_GNA_PARTIAL_FLOATS): [0..28] as mode 1, [29..31] H_aa = {aa, ab, bb}, [32,33] b_a, [34..39] H_{a,pose}, [40..45] H_{b,pose};
 * segment record (SP_GNA_SEG_FLOATS): [0..7] as mode 1, [8] H_{a,depth}, [9] H_{b,depth}.  Consumed by sp_window_gn_step. */
int sp_pairs_cost(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int mode, float irls_eps,
                  float* span_partials, float* seg_partials, void* stream);

/* Adam step on {kld, left SE(3) tangent, affine} of every pair from the mode-0 partials: reduces the tile
 * partials (fixed order, fp64), loss = |residual| like odometery/two_frame_sfm.py:201-206, maps d/dpose to the
 * tangent of Exp(a)*T, applies torch.optim.Adam semantics (betas 0.9/0.999, eps 1e-8, bias correction) with
 * per-group learning rates, retracts pose <- Exp(step)*pose.  state: per pair (2*(N+6+2)+2) floats, zeroed
 * by the caller before the first step.  losses: [n_pairs] written every step. */
int sp_pairs_adam_step(const SpPair* pairs, int n_pairs, int max_N, const float* span_partials, const float* seg_partials,
                       float lr_kld, float lr_pose, float lr_aff, float* state, float* losses, void* stream);

/* Gauss-Newton / Levenberg-Marquardt step from the mode-1 partials: per pair, reduce tiles, eliminate the
 * N diagonal log-depth unknowns (Schur complement onto the 6x6 pose block), solve in fp64, update
 * pose <- Exp(delta)*pose and kld += delta_d.  lm_state: per pair SP_LM_STATE_FLOATS floats {lambda, cost of
 * the last accepted point (init -1), n accepted, n rejected, rejected-last-call flag, last cost seen, the valid points of that
 * evaluation / P, and -- schedules only -- how the pair's last finished phase ended: +iterations = on its cap, -iterations = by
 * its convergence test};
 * lambda adapts on device (cost up -> step undone from the backup, lambda*=lm_up, re-evaluated next call;
 * cost down -> lambda = max(lambda*lm_down, lm_min)).  backup: per pair (16+max_N) floats.  costs: [n_pairs]. */
#define SP_LM_STATE_FLOATS 8
int sp_pairs_gn_step(const SpPair* pairs, int n_pairs, int max_N, const float* span_partials, const float* seg_partials,
                     float lm_up, float lm_down, float lm_min, float* lm_state, float* backup, float* costs, void* stream);

/* Per-pair convergence on the device (the counterpart of the reference's relative-loss early stop, odometery/odometery.py:907-915,
 * for a batch whose pairs need different numbers of iterations).  sp_pairs_gn_step_conv marks done[pair] = 1 -- and leaves the pair
 * at its current point -- when the step that led to the evaluated point was accepted and lowered the cost by less than
 * conv_tol * cost; it returns immediately for pairs already marked.  sp_pairs_cost_active is sp_pairs_cost whose workgroups return
 * at once for spans of marked pairs: an iteration then costs time only for the pairs still moving.  done: n_pairs int32, zeroed
 * by the caller (per pyramid level).  conv_tol <= 0 / done == NULL: exactly sp_pairs_gn_step / sp_pairs_cost. */
int sp_pairs_cost_active(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int mode, float irls_eps,
                         float* span_partials, float* seg_partials, const int32_t* done, void* stream);
int sp_pairs_gn_step_conv(const SpPair* pairs, int n_pairs, int max_N, const float* span_partials, const float* seg_partials,
                          float lm_up, float lm_down, float lm_min, float* lm_state, float* backup, float* costs, float conv_tol,
                          int32_t* done, void* stream);

/* A whole coarse-to-fine schedule PER PAIR on the device.  Phase p of the schedule runs on the descriptors `pairs` of one pyramid
 * level (every phase describes the same n_pairs pairs) over the work list (chunks, spans) of that level's point set -- a coarse
 * level may run on a decimated copy of the source points with its own, shorter work list -- with IRLS epsilon irls_eps; a pair
 * moves on to phase p + 1 when an accepted step lowers its cost by less than conv_tol * cost, or after max_iters iterations, and
 * is finished at phase n_phases.  phase[n_pairs] / iters[n_pairs] (int32, zeroed by the caller) hold every pair's position; the
 * host only issues (sp_pairs_schedule_cost, sp_pairs_schedule_gn_step) until min(phase) == n_phases.  Pairs advance
 * independently: one slow pair no longer keeps the others at a coarse level, and finished pairs cost nothing.
 * sp_pairs_schedule_cost runs the Gauss-Newton cost pass over every DISTINCT work list (phases sharing `spans` share the list) in
 * ONE launch -- up to four lists of the same kind (wave spans or workgroup spans), one after the other in block order; otherwise a
 * launch per list; partial buffers of different work lists must not alias.  The struct lives in host memory. */
#define SP_MAX_PHASES 12
/* SP_PHASE_POSE_ONLY: the phase moves the pose alone, the log-depths stay where they are (the solver skips the Schur complement
 * and the depth update; the cost pass is unchanged).  From the reference's own starting distribution (pose off by SE3.Random(sigma =
 * 0.05), depth seeds log(2 + 2 rand), odometery/two_frame_sfm.py:77-81,103-105) a joint Gauss-Newton step lets single segments run
 * away in depth while the pose is still wrong -- the total cost keeps falling, so LM never objects; the reference's Adam moves the
 * pose 10x faster than the depths (lr 1e-2 vs 1e-3, :116-123) and so aligns the pose first.  A pose-only phase at the coarsest
 * level is the Gauss-Newton counterpart (tools/gn_model.py, profiles/r03_sigma05_sweep.txt). */
#define SP_PHASE_POSE_ONLY 1
/* SP_PHASE_WAVE_SPANS: the phase's work list is wave-granular (see SP_COST_WAVE_SPANS). */
#define SP_PHASE_WAVE_SPANS 2
/* SP_PHASE_DEPTH_TABLE: the phase's tables are depth tables (see SP_COST_DEPTH_TABLE); all phases of one schedule alike. */
#define SP_PHASE_DEPTH_TABLE 4
/* SP_PHASE_DEPTH_DAMP(k), k = 0..255: the phase's Gauss-Newton step damps the log-depth block by k / 8 on top of the LM lambda --
 * (H_dd (1 + lambda + k/8)) -- so that the depths follow the pose at a fraction of their Gauss-Newton step, the way the reference's
 * Adam moves them at a tenth of the pose's rate (odometery/two_frame_sfm.py:116-123: lr 1e-3 against 1e-2).  On near-planar scenes
 * (a narrow depth range: the two-fold ambiguity of a plane's homography) the undamped joint step walks into the second solution from
 * the reference's own starts; a damped coarse phase in front of the undamped ones does not (profiles/r05_reference_start.txt). */
#define SP_PHASE_DEPTH_DAMP_SHIFT 8
#define SP_PHASE_DEPTH_DAMP(k) (((k) & 0xff) << SP_PHASE_DEPTH_DAMP_SHIFT)
/* SP_PHASE_ADAM (ABI 13): the phase's update is THE REFERENCE'S OWN OPTIMISER instead of a Gauss-Newton step -- one torch.optim.Adam
 * step (betas 0.9 / 0.999, eps 1e-8, bias correction) on {left pose tangent, log-depths} at the rates SpSchedule.adam_lr_pose /
 * adam_lr_kld (odometery/two_frame_sfm.py:116-123: 1e-2 / 1e-3), pose <- Exp(step) pose.  The gradient is that of the phase's cost pass:
 * b_p and b_d of the IRLS normal equations are sum sign(r) J wherever |r| > irls_eps, i.e. the gradient of the reference's mean |r|
 * (two_frame_sfm.py:201) up to the smoothing inside irls_eps; no Hessian is used, no step is ever rejected.  The phase ends on max_iters
 * (conv_tol is ignored).  Moments: SpSchedule.adam_state, per pair / slot 2 + 2 (max_N + 8) floats laid out like sp_pairs_adam_step's;
 * they persist over consecutive phases (the reference keeps one optimiser over its three pyramid levels, :128-155) and are zeroed whenever
 * a pair (re)starts an attempt.  What it is for: the THIRD attempt of a pair whose two Gauss-Newton attempts failed their verdict
 * (SpSchedule.retry2_entry) -- Gauss-Newton's path from a few of the reference's own starts in ten thousand leads into the second solution
 * of a near-plane's homography under every damping tried, the reference's Adam path from the same starts does not (goldens g20y). */
#define SP_PHASE_ADAM 8
/* SP_PHASE_PREDICTED_EXIT (ABI 13): a pair also leaves the phase right after a step that is PREDICTED to buy less than conv_tol of its cost
 * -- the first-order change of sum |r| along the step, b . delta, which the solver has at hand -- instead of only after the NEXT evaluation
 * has shown that it did: one cost evaluation per phase saved (the all-points polish's is a tenth of a frame pair's work).  Skipped while
 * the LM lambda is above 1e-2 or the phase damps the depth block (a short step is then the damping's doing).  The step is applied. */
#define SP_PHASE_PREDICTED_EXIT 16
typedef struct SpPhase {
    const SpPair* pairs;
    const int32_t* chunks;
    const int32_t* spans;
    float* span_partials;        /* n_spans * SP_GN_PARTIAL_FLOATS */
    float* seg_partials;         /* 4 * n_chunks * SP_GN_SEG_FLOATS */
    int32_t n_spans;             /* 0 = an empty point set: nothing is launched, its pairs pass through the phase */
    int32_t max_iters;
    float irls_eps;
    float conv_tol;
    int32_t flags;               /* SP_PHASE_* */
    int32_t next;                /* the phase a pair moves on to: 0 = the following one (p + 1), k > p = phase k (skips phases p+1..k-1) */
} SpPhase;               /* 64 bytes */
/* entry: the phase a fresh pair starts in.  retry_entry: the phase a pair RESTARTS in -- from its initial pose and log-depths, once --
 * when its verdict at the end of the schedule says it failed (SpVerdict below); -1 = no second attempt.  The phases in front of
 * `entry` belong to the later attempts only: with phases {R0, R1 (next = J), P0, J, ..., polish}, entry = 2 and retry_entry = 0 a
 * first attempt runs P0, J, ..., polish and a second one R0, R1, J, ..., polish.  retry2_entry (ABI 13): where a pair restarts, from
 * its initial values again, when the SECOND attempt fails its verdict too -- -1 = no third attempt; with {A0, A1 (next = J'), R0, ...}
 * and retry2_entry = 0 the third attempt runs A0, A1, J', ..., polish (A* = SP_PHASE_ADAM phases in REFERENCE_START_SCHEDULE). */
typedef struct SpSchedule {
    SpPhase phase[SP_MAX_PHASES];
    int32_t n_phases;
    int32_t entry;
    int32_t retry_entry;
    int32_t retry2_entry;
    float adam_lr_pose, adam_lr_kld;     /* SP_PHASE_ADAM phases */
    float* adam_state;                   /* [pairs or slots][2 + 2 (max_N + 8)], device; NULL when no phase is SP_PHASE_ADAM */
} SpSchedule;            /* 800 bytes */

/* THE VERDICT of a scheduled run, per pair, on the device.  The reference asserts finiteness every iteration and nothing else
 * (core/dense_optim.py:311,321,340-343); a Gauss-Newton schedule that ends in the wrong basin (about one of the reference's own
 * starts in a thousand, odometery/two_frame_sfm.py:77-84,103-105) would otherwise hand back a wrong pose silently.  The solver
 * workgroup that takes a pair out of its last phase writes
 *   status[pair]  SP_STATUS_* bits; 0 = converged at the first attempt, SP_STATUS_RETRIED alone = converged at the second;
 *   diag[pair * SP_DIAG_FLOATS]  {final cost, max_n |kld_n - kld0_n|, valid points / points of the last evaluated cost,
 *                                 iterations spent in the last phase, attempts made, cost at the first evaluation,
 *                                 median and maximum over the pair's segments of the segment's mean |r| (0 = not judged)}
 * and, when (status & retry_mask) and the schedule has a retry_entry and the pair has made one attempt, puts pose and log-depths
 * back to pose0 / kld0, resets its LM state (lambda = lam0) and restarts it at retry_entry IN THE SAME LAUNCH -- the pair keeps its
 * slot, the resident set stays full.  A pair's verdict depends on the pair alone (no batch statistics): bitwise what it is alone.
 * pose0 / kld0: the initial values, laid out like the batch's own pose / kld arrays (pose_base / kld_base: element 0 of those). */
#define SP_STATUS_NONFINITE   1   /* a non-finite pose entry, log-depth or cost */
#define SP_STATUS_LAST_CAP    2   /* the last phase ended on its iteration cap, not by its convergence test */
#define SP_STATUS_DEPTH_RANGE 4   /* some |kld - kld0| > kld_bound: a segment's depth ran away */
#define SP_STATUS_COST        8   /* final cost > cost_bound (absolute), or > cost_ratio * the cost at the first evaluation; the host layer also sets it
                                   * for a final cost far above the batch median (optim/pair_batch.py VERDICT_DEFAULTS['cost_outlier']) */
#define SP_STATUS_VALID       16  /* valid points of the last evaluated cost < valid_min * points */
#define SP_STATUS_SEGMENTS    32  /* (ABI 13) the cost is UNEVEN over the pair's own segments: the worst segment's mean |r| > seg_max_ratio x the median
                                   * segment's, or the pair's cost > seg_mean_ratio x the median segment's -- what the second solution of a near-plane's
                                   * homography looks like from the inside (some segments explained, others not); needs no other pair to compare with */
#define SP_STATUS_RETRIED     0x100   /* the verdict is a later attempt's (second or third) */
#define SP_STATUS_ADAM        0x400   /* ... the third one's: the pair went through the schedule's SP_PHASE_ADAM phases (retry2_entry) */
#define SP_STATUS_UNFINISHED  0x200   /* the run ended (max_rounds) before the pair left its last phase */
#define SP_DIAG_FLOATS 8
typedef struct SpVerdict {
    int32_t* status;             /* [pairs] written when a pair finishes (an attempt that is retried leaves nothing) */
    float* diag;                 /* [pairs * SP_DIAG_FLOATS] */
    int32_t* attempts;           /* [pairs] zeroed by the caller */
    const float* pose0;          /* [pairs * 16] */
    const float* kld0;           /* like the batch's flat log-depth array */
    const float* pose_base;
    const float* kld_base;
    float kld_bound;             /* <= 0: not tested */
    float cost_bound;            /* <= 0: not tested */
    float cost_ratio;            /* <= 0: not tested */
    float valid_min;             /* <= 0: not tested */
    int32_t retry_mask;          /* status bits that send a pair into its next attempt */
    float lam0;                  /* LM damping a later attempt starts with */
    float seg_max_ratio;         /* <= 0: not tested.  Segments with fewer than SP_VERDICT_SEGMENT_POINTS valid points in the last cost pass are not */
    float seg_mean_ratio;        /* judged (of the pair's first SP_VERDICT_SEGMENTS segments); pairs with fewer than SP_VERDICT_MIN_SEGMENTS judged segments are not tested */
    int32_t* evals;              /* [pairs * SP_MAX_PHASES] or NULL, zeroed by the caller: cost evaluations of every pair in every phase (diagnostics:
                                  * bench.py prices a scheduled run's algorithmic bytes with it, roofline_schedule) */
    float seg_product;           /* <= 0: not tested.  (cost / median - 1) x (worst / median) above this: a moderately raised cost TOGETHER WITH a clear
                                  * outlier segment -- neither ratio alone separates such a pair from a converged one with noisy small segments */
    float pad_;
} SpVerdict;             /* 104 bytes */
#define SP_VERDICT_SEGMENTS 2048
#define SP_VERDICT_SEGMENT_POINTS 64
#define SP_VERDICT_MIN_SEGMENTS 8
int sp_pairs_schedule_cost(const SpSchedule* sched, const int32_t* phase, void* stream);
int sp_pairs_schedule_gn_step(const SpSchedule* sched, int n_pairs, int max_N, float lm_up, float lm_down, float lm_min,
                              float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters, const SpVerdict* verdict /* or NULL */,
                              void* stream);

/* The host loop of a scheduled run, natively: issues (sp_pairs_schedule_cost, sp_pairs_schedule_gn_step) up to max_rounds times and,
 * every check_every iterations, reads min(phase) back (one tiny kernel, a 4-byte copy into flag_host -- PINNED host memory -- and a
 * synchronisation of `stream` only) to stop once every pair has finished; work lists whose phases all lie behind that minimum are
 * no longer launched (pairs only move forward).  One foreign call per scheduled run instead of ~100: a
 * Python caller issues nothing per iteration, so several batches can run their schedules from several host threads without
 * contending for the interpreter lock (optim/pair_stream.py), and the launch-bound tail of a schedule runs at the rate of the
 * runtime's launch path.  flag_dev / flag_host: EIGHT int32 each (device scratch / pinned host memory; ABI 13: {min phase, queue head, attempts left, busy slots, occupied phases, ...}).  Returns the number of iterations launched (>= 0), SP_EINVAL, or
 * -(1000 + hipError_t) for a runtime error.  This is the one entry point that synchronises the host (with `stream` only). */
int sp_pairs_schedule_run(const SpSchedule* sched, int n_pairs, int max_N, float lm_up, float lm_down, float lm_min,
                          float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters, int check_every,
                          int max_rounds, int32_t* flag_dev, int32_t* flag_host, const SpVerdict* verdict /* or NULL */, void* stream);

/* SLOT-LEVEL CONTINUOUS BATCHING of a scheduled run.  n_queue pairs are resident (tables, unknowns, descriptors of every phase's
 * level / lattice: qpairs[p][n_queue], and the batch's work lists, which cover all n_queue pairs); n_slots <= n_queue of them are
 * worked on at a time -- the schedule's phase[p].pairs are the SLOT descriptors (slot_pairs[p] = phase[p].pairs, writable, n_slots
 * records, initially copies of the first n_slots pairs'), and lm_state / backup / costs / phase / iters are per slot.  The cost pass
 * is launched over VIRTUAL spans: max_spans[p] per slot (the largest span count of any pair in phase p's work list; a multiple of 4
 * for wave spans); virtual span v belongs to slot v / max_spans[p] and is span  slot descriptor's tile0 + v % max_spans[p]  of the
 * batch's list when that is below the descriptor's n_tiles, nothing otherwise.  So pairs of DIFFERENT padded layouts (ragged
 * segment sets: SAM masks, frontend/process_frame.py:207-250) share the slots, and every pair's partial sums are taken over its own
 * spans in its own order.  The solver launch that finishes a pair files its result under the pair's index (q_costs[pair], q_lm[pair *
 * SP_LM_STATE_FLOATS], the verdict's status / diag; pose and log-depths live in the pair's own storage), takes the next waiting pair
 * from `head` (device int32, initialised to n_slots; one atomic per finished pair), copies that pair's descriptor of every phase over
 * the slot's and restarts the slot at the schedule's entry phase with lambda = lam0.  The resident set stays full until the queue is
 * empty: no launch works on a thinning batch except the very last ones.  Which pair lands in which slot depends on timing; every
 * pair's result does not (pairs never interact; bitwise what the pair gives with all pairs resident).  slot_pair: device [n_slots],
 * initialised 0..n_slots-1.  phase[p].n_spans is ignored (n_slots * max_spans[p] virtual spans are launched).
 * flag_dev / flag_host: EIGHT int32 each (min phase, queue head, attempts left, busy slots, bit mask of the phases that hold a pair, spare).  Otherwise as sp_pairs_schedule_run; pairs still in a slot when the
 * run ends on max_rounds get SP_STATUS_UNFINISHED in verdict->status (when given) and their state as it is. */
typedef struct SpQueue {
    const SpPair* qpairs[SP_MAX_PHASES];
    SpPair* slot_pairs[SP_MAX_PHASES];
    int32_t max_spans[SP_MAX_PHASES];
    int32_t n_queue;
    int32_t pad_;
    int32_t* head;
    int32_t* slot_pair;
    float* q_costs;
    float* q_lm;
    float lam0;
    int32_t pad2_;
    int32_t* active;             /* (ABI 13) device scratch [n_slots] or NULL: the slots that still work on a pair, listed at every poll -- once the queue
                                  * is empty and at most an eighth of the slots are busy, the launches of a round go over those alone (the tail of a run:
                                  * a third attempt iterates 1500 rounds by itself) */
} SpQueue;               /* 296 bytes */
int sp_pairs_schedule_run_queue(const SpSchedule* sched, const SpQueue* queue, int n_slots, int max_N, float lm_up, float lm_down,
                                float lm_min, float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters,
                                int check_every, int max_rounds, int32_t* flag_dev, int32_t* flag_host, const SpVerdict* verdict /* or NULL */,
                                void* stream);

/* One optimiser iteration of every pair as a SINGLE launch: the workgroup that completes the last span of a pair
 * runs that pair's update in place (same arithmetic, same fixed reduction order as sp_pairs_cost followed by
 * sp_pairs_adam_step / sp_pairs_gn_step -- results are bitwise identical).  arrivals: n_pairs int32, zeroed once by
 * the caller (the kernel leaves it zeroed).  Other arguments as in the two-launch forms. */
int sp_pairs_adam_iterate(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int n_pairs, int max_N,
                          float* span_partials, float* seg_partials, int32_t* arrivals, float lr_kld, float lr_pose, float lr_aff, float* state,
                          float* losses, void* stream);
int sp_pairs_gn_iterate(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int n_pairs, int max_N,
                        float irls_eps, float* span_partials, float* seg_partials, int32_t* arrivals, float lm_up, float lm_down,
                        float lm_min,
                        float* lm_state, float* backup, float* costs, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused SE(3)-Adam optimiser of the reference's driver loops (SURVEY.md section 8(b) "sp_adam_se3_step", 8(f) N1):
 * torch.optim.Adam + lietorch retr()/matrix() + the pose bookkeeping around one cost evaluation, for
 *   odometery/two_frame_sfm.py:116-123,150-207 (persistent tangent, pose = Exp(a) X, no update on the very first
 *   iteration), odometery/odometery.py:300-312,375-407 (tracking: zero-reset tangent, T_supp <- T_supp inv(Exp(d))) and
 *   odometery/odometery.py:576-648,756-915 (windowed mapping: several source keyframes, relative poses
 *   D_trg inv(T_trg) T_src inv(D_src), first keyframe fixed, oldest keyframe's depths frozen when the window is full,
 *   fold-in + renormalise_se3 + tangent reset every iteration, relative-loss early stop).
 *
 * The window is a graph.  NODES = poses; BLOCKS = per-keyframe log-depth vectors; EDGES = photometric terms
 * (source keyframe -> target frame).  Edge e IS pair e of an SpPair array (one array per pyramid level, same kld / pose /
 * aff pointers in all of them): SpPair.kld = its block's kld, SpPair.pose = a 16-float slot and SpPair.aff = a 4-float slot
 * {a_src, b_src, a_trg, b_trg} that sp_window_compose / sp_window_step fill.  One iteration =
 *     sp_pairs_cost(mode 0) over all edges  ->  sp_window_step                                   (3 launches, no host sync)
 * ---------------------------------------------------------------------------------------------------- */
typedef struct SpWindowNode {
    float T[16];         /* kind 0: camera-to-world pose, updated in place by the fold-in; kind 1: the constant group element X */
    float a[6];          /* tangent [tau, phi]: zero between iterations (kind 0) / the persistent parameter (kind 1) */
    float m[6], v[6];    /* Adam moments of the tangent */
    float aff[2];        /* affine brightness (a, b) of this frame */
    float aff_m[2], aff_v[2];
    float lr_pose;       /* 0 = pose not optimised (first keyframe, odometery.py:589-592; source-only frames) */
    float lr_aff;        /* 0 = affine not optimised */
    int32_t kind;        /* 0: edge pose = inv(T_trg) T_src evaluated at zero tangents, T <- T inv(Exp(d)) after every step;
                            1: edge pose = Exp(a) X (two-frame SfM; such a node is only ever a target) */
    int32_t flags;       /* bit 0: renormalise_se3 after every iteration (odometery.py:868,879) */
} SpWindowNode;          /* 176 bytes */

typedef struct SpWindowEdge {
    int32_t src_node;    /* pose node of the source keyframe, -1 = identity (two-frame SfM) */
    int32_t trg_node;
    int32_t block;       /* log-depth block of the source keyframe */
    float weight;        /* loss = sum_e weight_e * (abs_loss ? |residual_e| : residual_e) */
} SpWindowEdge;

typedef struct SpWindowBlock {
    float* kld;          /* [N] -- the SpPair.kld of every edge whose source is this keyframe */
    float* m;            /* [N] Adam moments */
    float* v;
    int32_t N;
    float lr;            /* 0 = frozen (oldest keyframe of a full window, odometery.py:594-603; tracking) */
} SpWindowBlock;         /* 32 bytes */

/* doubles of scratch sp_window_step needs */
int sp_window_scratch_doubles(int n_edges, int max_N);

/* Write every edge's relative pose and affine slot from the current nodes (call once before the first cost pass). */
int sp_window_compose(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes, void* stream);

/* One optimiser step from the mode-0 partials of sp_pairs_cost over `pairs`.  abs_loss: 1 = sum_e w_e |residual_e|
 * (two_frame_sfm.py:201-202), 0 = sum_e w_e residual_e (odometery.py:394,845-850).  skip_first: no parameter update on
 * iteration 0 (two_frame_sfm.py:203).  rel_tol > 0: once |loss - previous loss| / previous loss < rel_tol the window
 * freezes (later calls return without touching anything), like the break at odometery.py:907-915.
 * state: 12 floats, zeroed by the caller {Adam step count, iterations done, previous loss, converged flag, last loss, -,
 *   beta1^t and beta2^t as two doubles in [6..9]}; setting [0] = 0 restarts Adam's bias correction;
 * losses[max_losses]: loss of iteration i (evaluated BEFORE its update) at index i. */
int sp_window_step(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                   const SpWindowBlock* blocks, int n_blocks, int max_N, const float* span_partials, const float* seg_partials,
                   double* scratch, int abs_loss, int skip_first, float rel_tol, float* state, float* losses, int max_losses,
                   void* stream);

/* Gauss-Newton / Levenberg-Marquardt step of the SAME window graph (BASELINE.json north_star: "Gauss-Newton/LM solve on SE(3) (+)
 * log-depth"; the reference's loops it accelerates: odometery/odometery.py:375-407 tracking -- 6 pose + 2 affine unknowns, keyframe
 * depths fixed -- and :756-915 windowed mapping -- K poses + sum N log-depths + affines, first keyframe fixed (:589-592), oldest
 * depths frozen (:594-603), fold-in T <- T inv(Exp(D)) + renormalise_se3 after the step (:861-882)).  One iteration =
 *     sp_pairs_cost(mode 2, irls_eps) over all edges  ->  sp_window_gn_step                       (4 launches, no host sync)
 * Unknowns: the pose tangent of every node with lr_pose > 0, the affine pair of every node with lr_aff > 0 (the reduced camera system,
 * fp64: in LDS as a packed triangle up to 192 scalars -- the reference's own window, config/tum/odom_desk.yaml window_size 5 with two
 * supporting frames per keyframe and two running ones, all poses and affine pairs free (odometery.py:523-575,611-616), is 14 free
 * nodes = 112 -- and through global scratch up to 512 = 64 nodes), the log-depths of every block with lr > 0 (eliminated by a Schur
 * complement, like the per-pair solver; a depth unknown of keyframe k only couples with the frames k is matched against, and only
 * those columns are stored and summed).  No camera unknown at all is allowed (the 'supp' mapping of odometery.py:576-648: only the
 * latest keyframe's depths are free).  flags bit 0: pose-only step (all depth blocks treated as frozen); bit 1 (ABI 13): PREDICTED EXIT --
 * with conv_tol > 0 the window also freezes right after a step that is predicted to buy less than conv_tol of the loss (the first-order
 * change b . delta along the step; lambda <= 1e-2 only), without the evaluation that would confirm it: see SP_PHASE_PREDICTED_EXIT.
 * Bit 2 (ABI 14): the caller's word that NO depth block moves (every SpWindowBlock.lr is 0: frame-to-keyframe tracking) -- like bit 0 the
 * Schur-term launch is then left out (its terms are all zero): three launches per iteration instead of four.
 * LM: lambda adapts on the device exactly like sp_pairs_gn_step (loss up -> previous step undone from nodes_backup / kld_backup,
 * lambda *= lm_up, re-evaluated next call; loss down -> lambda = max(lambda lm_down, lm_min); a failed factorisation counts as a
 * rejected step: lambda *= lm_up, the same point is evaluated and solved again); conv_tol > 0: an accepted step that
 * lowered the loss by less than conv_tol * loss freezes the window (later calls return at once), like the relative-loss break at
 * odometery.py:907-915.
 * n_unknowns: the camera unknowns of the window (6 per node with lr_pose > 0 + 2 per node with lr_aff > 0), which sizes the scratch
 *   and selects the kernel; a window that turns out to have more freezes with state[9] = 1.
 * state: 16 floats {lambda (set by the caller, e.g. 1e-4), loss at the last accepted point (init -1), accepted, rejected,
 *   rejected-last flag, iterations, converged flag, last loss, failed solves, too-many-unknowns flag, ...}; losses[max_losses]: loss of
 *   call i.  scratch: sp_window_gn_scratch_doubles(n_edges, n_blocks, sum_N, max_N, n_unknowns) doubles; nodes_backup: n_nodes nodes;
 *   kld_backup: sum_N floats (blocks in order; sum_N = total log-depths of all blocks). */
int sp_window_gn_scratch_doubles(int n_edges, int n_blocks, int sum_N, int max_N, int n_unknowns);
/* where in the scratch the update kernel leaves 16 time stamps of its phases (100 MHz ticks; diagnostics, tools/window_bench.py) */
int sp_window_gn_profile_offset(int n_edges, int n_blocks, int sum_N, int max_N, int n_unknowns);
int sp_window_gn_step(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                      const SpWindowBlock* blocks, int n_blocks, int sum_N, int max_N, int n_unknowns, const float* span_partials,
                      const float* seg_partials, double* scratch, SpWindowNode* nodes_backup, float* kld_backup, int flags,
                      float lm_up, float lm_down, float lm_min, float conv_tol, float* state, float* losses, int max_losses,
                      void* stream);

/* S WINDOWS PER LAUNCH (ABI 13, config 3's throughput form: S independent sequences side by side, as PairBatch is config 2's).  One window's
 * Gauss-Newton iteration is four small dependent launches that leave the chip empty; here every launch covers all the windows -- the cost
 * pass goes over the S work lists one after the other in block order, the reduce / Schur / update kernels take the window from blockIdx.z
 * (grids sized for the largest window) -- and the host loop polls ALL states with one copy.  windows: host array, one record per window
 * (what sp_window_gn_run takes as arguments); args_dev: device scratch of n_windows * sp_window_gn_multi_bytes() bytes; states_dev / states_host
 * (pinned): n_windows * 16 floats.  The update kernel's instantiation is that of the window with the most camera unknowns.  Per window the
 * arithmetic is sp_window_gn_run's (bitwise: tests/test_gpu_window_gn.py).  With conv_tol > 0 the loop ends when EVERY window froze; a frozen
 * window's launches return at once.  Returns the iterations launched or a negative error. */
typedef struct SpWindowGn {
    const SpPair* pairs;             /* the window's edges as frame pairs, at the pyramid level of this phase */
    const int32_t* chunks;
    const int32_t* spans;
    const SpWindowEdge* edges;
    SpWindowNode* nodes;
    const SpWindowBlock* blocks;
    float* span_partials;
    float* seg_partials;
    double* scratch;                 /* sp_window_gn_scratch_doubles(...) */
    SpWindowNode* nodes_backup;
    float* kld_backup;
    float* state;                    /* 16 floats, as sp_window_gn_step */
    float* losses;
    int32_t n_spans, n_edges, n_nodes, n_blocks, sum_N, max_N, n_unknowns, max_losses;
} SpWindowGn;                        /* 136 bytes */
int sp_window_gn_multi_bytes(void);
int sp_window_gn_run_multi(const SpWindowGn* windows, int n_windows, float irls_eps, int flags, float lm_up, float lm_down, float lm_min,
                           float conv_tol, int max_iters, int check_every, void* args_dev, float* states_dev, float* states_host, void* stream);

/* Up to max_iters iterations -- sp_pairs_cost(mode 2, irls_eps) over the window's work list + sp_window_gn_step -- as ONE foreign call
 * (the Python loop of optim/window.py run_gn): every check_every iterations the 16-float state is copied to state_host (pinned) and this
 * stream synchronised; with conv_tol > 0 the loop ends once the window froze.  Returns the iterations launched (>= 0), SP_EINVAL, or
 * -(1000 + hipError_t). */
int sp_window_gn_run(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, float irls_eps,
                     const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes, const SpWindowBlock* blocks, int n_blocks,
                     int sum_N, int max_N, int n_unknowns, float* span_partials, float* seg_partials, double* scratch,
                     SpWindowNode* nodes_backup, float* kld_backup, int flags, float lm_up, float lm_down, float lm_min, float conv_tol,
                     float* state, float* losses, int max_losses, int max_iters, int check_every, float* state_host, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Helpers around the path
 * ---------------------------------------------------------------------------------------------------- */

/* Dense (N,H,W) seeded depths.  log_space = 1: (L + shift_n) * mask  (core/dense_optim.py:38-80
 * infer_depth_seeds);  log_space = 0: exp of that  (core/dense_optim.py:164-174 unproject_kf_to_depths). */
int sp_depth_expand(const uint8_t* masks, const float* logdepth, const float* keypoints, const float* kld, int N,
                    int H, int W, int log_space, float* out, void* stream);

/* core/ops.py:59-96 estimate_depth_diff (mean=False) on table points moved by pose: last-writer-wins z splat
 * at truncated (v,u); ties resolved by highest point index (deterministic, one of the orders the reference's
 * scatter_ may produce).  out: (H,W) zero-initialised by this call.  keys: H*W uint64 scratch. */
int sp_depth_splat(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                   const float* kld, int N, int P, int H, int W, const float* K, const float* pose,
                   unsigned long long* keys, float* out, void* stream);

/* core/ops.py:84-92 estimate_depth_diff(mean=True): scatter_reduce_('mean') at the truncated pixel, WITH the reference's "initial
 * zero counted" semantics (include_self defaults to True): a pixel hit by c points holds sum(z) / (c + 1), an untouched one 0.
 * acc: 12*H*W bytes of scratch (32.32 fixed-point sums + counts: order independent).  No reference caller passes mean=True. */
int sp_depth_splat_mean(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L, const float* kld, int N,
                        int P, int H, int W, const float* K, const float* pose, void* acc, float* out, void* stream);

/* odometery/depth_init.py:10-67 segment_based_depth_reinit.  mode 0 = mean, 1 = median (lower middle for
 * even counts, like torch.median).  est_depth (H,W).  scratch: P floats.  out_kld[N], out_visible[N] u8.
 * Invisible segments receive the (lower) median of the visible segments' values. */
int sp_segment_reinit(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L, int N,
                      int P, int H, int W, const float* est_depth, int mode, float* scratch, float* out_kld,
                      uint8_t* out_visible, void* stream);

/* depth_completion/segment_based_completion.py:21-27 render_depth_avg fused with the expansion and the
 * visible-segment filter: per-pixel mean over covering visible segments of exp(L + shift_n)
 * (visible may be NULL = all).  out_depth (H,W), out_invalid (H,W) u8.  acc: 12*H*W bytes of scratch
 * (32.32 fixed-point sums + counts: the accumulation is order-independent, hence reproducible). */
int sp_depth_average(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                     const float* kld, const uint8_t* visible, int N, int P, int H, int W, void* acc,
                     float* out_depth, uint8_t* out_invalid, void* stream);

/* The two halves of sp_depth_average, for SEGMENT-SHARDED depth completion (BASELINE.json configs[3], SURVEY.md
 * section 8(e)): every rank accumulates its own segments into acc = {sums[H*W] uint64 (32.32 fixed point), counts[H*W]
 * uint32}, the ranks all_reduce(SUM) the two integer arrays (exact and order independent, so the result is bitwise the
 * single-GPU one; a pixel is invalid iff its summed count is 0, i.e. the OR of the per-rank validity), rank-local finish. */
int sp_depth_accumulate(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                        const float* kld, const uint8_t* visible, int N, int P, int H, int W, void* acc, void* stream);
int sp_depth_average_finish(const void* acc, int H, int W, float* out_depth, uint8_t* out_invalid, void* stream);

/* The pose parameter of the reference's drivers, T[b] = Exp(a[b]) * X[b]  (a = [tau, phi] (n,6); X, T (n,4,4)):
 * lietorch's LieGroupParameter.retr().matrix() at odometery/two_frame_sfm.py:83-84, odometery/odometery.py:224-228.
 * grad_T == NULL: forward, writes T.  grad_T != NULL: backward, writes grad_a[b] = (dT/da)^T grad_T[b] (n,6). */
int sp_se3_retract(const float* a, const float* X, int n, float* T, const float* grad_T, float* grad_a, void* stream);

/* lie/lie_algebra.py:41-47 renormalise_se3 on n row-major 4x4 matrices, in place. */
int sp_renormalise_se3(float* T, int n, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Keyframe post-processing: frontend/segment/post_processer.py without cupy.
 * ---------------------------------------------------------------------------------------------------- */

/* post_processer.py:13-36 depth_discontinuity + mask_by_depth_discontinuity: depth = exp(logdepth) (-1 where
 * !valid), filter_size x filter_size max-pool (stride 1), Scharr/32 gradient magnitude (reflect padding) > threshold.
 * split = valid & !discontinuity (N,H,W) u8; disc (optional) = valid & discontinuity.  scratch: N*H*W floats. */
int sp_depth_discontinuity(const float* logdepth, const uint8_t* valid, int N, int H, int W, int filter_size,
                           float threshold, float* scratch, uint8_t* split, uint8_t* disc, void* stream);

/* post_processer.py:57-64 connected_components_batch (ndimage.label with the per-slice 4-connectivity structure):
 * labels[i] = 1 + (smallest linear index of i's component), 0 for background -- sorting components by label gives
 * scipy's scan-order numbering.  parent: N*H*W int32 scratch; sizes (optional, N*H*W int32): sizes[root] = pixels. */
int sp_label_components(const uint8_t* fg, int N, int H, int W, int32_t* parent, int32_t* labels, int32_t* sizes,
                        void* stream);

/* Compact list of components {slice, root, size} (unordered, at most cap entries; *n_parts = how many exist) and
 * bg_sizes[n] = |mask_n & !split_n| (the label-0 part post_process_kf forms, post_processer.py:127-133). */
int sp_collect_parts(const int32_t* labels, const int32_t* sizes, const uint8_t* masks, const uint8_t* split, int N, int H,
                     int W, int cap, int32_t* parts, int32_t* n_parts, int32_t* bg_sizes, void* stream);

/* Materialise K part masks (K,H,W) u8.  parts[k] = {slice, kind, root}: kind 0 = component `root` of that slice,
 * 1 = mask & !split, 2 = the original mask of the slice (post_processer.py:138-146). */
int sp_build_part_masks(const uint8_t* masks, const uint8_t* split, const int32_t* labels, int H, int W,
                        const int32_t* parts, int K, uint8_t* out, void* stream);

/* (row, col) of the kth[k]-th set pixel of mask k in raster order, i.e. torch.where(mask)[kth] as used by
 * sample_pts_in_mask (post_processer.py:67-84).  row_off: the row_counts output of sp_mask_count for these masks. */
int sp_kth_mask_pixel(const uint8_t* masks, const int32_t* row_off, int K, int H, int W, const int32_t* kth,
                      int32_t* out_rc, void* stream);

/* odometery/kf_criteria.py:7-21 translation_difference, :23-34 rotation_difference and the depth-validity ratio of
 * odometery/odometery.py:1003-1004, in one launch without a host sync.  depth: n floats (the rendered depth of the
 * latest keyframe); poses row-major 4x4.  out[4] = {#(depth > thresh)/n, scale = lower median of the valid depths
 * (torch.median; NaN when none is valid), |t_src - t_trg| / (scale + 1e-6), rotation angle of
 * inv(pose_src) pose_trg in degrees}. */
int sp_kf_criterion(const float* depth, int n, float thresh, const float* pose_src, const float* pose_trg, float* out,
                    void* stream);
/* The same four values -- bit for bit -- from a grid of workgroups instead of one (one launch per radix pass; 102 -> 17 us on a 640 x 480
 * render): ws = sp_kf_criterion_ws_words() uint32 of scratch, any content. */
int sp_kf_criterion_ws_words(void);
int sp_kf_criterion_ws(const float* depth, int n, float thresh, const float* pose_src, const float* pose_trg, uint32_t* ws, float* out,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------
 * ONE CALL PER FRAME of the monocular-odometry chain (ABI 14; BASELINE configs[2] as a sequence).  The reference's driver loop
 * (odometery/odometery.py:1018-1075) does, for every frame that is not a keyframe: track_frame against the latest keyframe (:323-449),
 * mapping(mode='supp') -- the latest keyframe's depths against its two running supporting frames (:1038-1042) -- and is_kf (:986-1016:
 * the keyframe's depth rendered in the tracked pose, validity ratio / scaled translation).  Behind Gauss-Newton windows that are built
 * once per keyframe (sp_window_gn_run) the interpreter between those steps -- three dozen small tensor operations, node arrays read back
 * and re-uploaded, a host synchronisation per step -- costs more than the kernels.  sp_chain_step runs the stages named in `stages` with
 * ALL state on the device -- poses, affine pairs and depths are read from and written to device buffers the caller names -- and the only
 * values that reach the host are the windows' 16-float LM states (the polls of sp_window_gn_run) and the 4 floats of the criterion.
 *   SP_CHAIN_TRACK     : pyramid of `image` (sp_blur_decimate) packed (sp_pack_rgb) into the tracker's target buffers; the target node's
 *                        pose / affine pair <- track_target.pose / .aff, tangents and moments cleared, edges recomposed; fresh LM state; the
 *                        phases of `track`; out_pose <- renormalise_se3(the node's pose) (lie/lie_algebra.py:41-47), out_aff <- its pair
 *   SP_CHAIN_SUPP      : (supp_images bit 0) the packed levels of supp_target[1] move to supp_target[0]; (bit 1) the tracker's packed levels
 *                        are copied into supp_target[1]; both nodes <- their pose / aff; fresh LM state; the phases of `supp`; then
 *                        kld_n floats kld_src -> kld_dst (the mapped depths into the tracker's block)
 *   SP_CHAIN_CRITERION : rel <- inv(out_pose) kf_pose; sp_depth_splat of the keyframe under rel into depth_out; sp_kf_criterion_ws(depth_out,
 *                        out_pose, kf_pose) -> crit (device) -> crit_host (pinned), and this stream is synchronised
 * Stages run in that order on `stream`.  The struct lives in HOST memory; iterations the phases took are written back to track_iters /
 * supp_iters.  Returns 0, SP_EINVAL, or what the stage's own entry point returned.
 * ---------------------------------------------------------------------------------------------------- */
#define SP_CHAIN_LEVELS 4
#define SP_CHAIN_PHASES 6
#define SP_CHAIN_TRACK 1
#define SP_CHAIN_SUPP 2
#define SP_CHAIN_CRITERION 4
typedef struct SpChainPhase {
    int32_t level;                     /* pyramid level (index into SpChainWindow.gn) */
    int32_t max_iters;
    float irls_eps, conv_tol;
} SpChainPhase;                        /* 16 bytes */
typedef struct SpChainWindow {         /* a built window with its Gauss-Newton schedule */
    SpWindowGn gn[SP_CHAIN_LEVELS];    /* the window at pyramid level l (pairs == NULL: the window has no descriptors there) */
    SpChainPhase phase[SP_CHAIN_PHASES];
    int32_t n_phases;
    int32_t check_every;               /* sp_window_gn_run's */
    int32_t flags;                     /* sp_window_gn_step's (bit 1: predicted exit, bit 2: no depth block moves) */
    float lam0;                        /* LM damping every frame starts from */
    float lm_up, lm_down, lm_min;
    int32_t check_first;               /* > 0: every phase looks at the state after this many iterations for the first time (a window that
                                        * usually converges at once -- the supplementary mapping -- should not run check_every iterations blind) */
    float* state_host;                 /* host, pinned: 16 floats */
} SpChainWindow;                       /* 680 bytes */
typedef struct SpChainTarget {         /* a target node whose frame the step replaces */
    float* packed[SP_CHAIN_LEVELS];    /* the node's packed (H_l, W_l, 3) image at pyramid level l, where the window has that level */
    const float* pose;                 /* 16 floats: the node's new camera-to-world pose */
    const float* aff;                  /* 2 floats, or NULL (no affine compensation) */
    int32_t node;
    int32_t pad_;
} SpChainTarget;                       /* 56 bytes */
typedef struct SpChainStep {
    int32_t stages;
    int32_t H, W, n_levels;            /* the frame; pyramid levels to build (level 0 = the image itself) */
    const float* image;                /* planar (3,H,W) */
    float* level[SP_CHAIN_LEVELS];     /* planar scratch of levels 1 .. n_levels - 1 ([0] unused) */
    SpChainWindow track;
    SpChainTarget track_target;
    float* out_pose;                   /* 16 floats */
    float* out_aff;                    /* 2 floats or NULL */
    SpChainWindow supp;
    SpChainTarget supp_target[2];      /* the latest keyframe's two running supporting frames, older first */
    int32_t supp_images;               /* bit 0: slot 1 -> slot 0; bit 1: this frame -> slot 1 */
    int32_t kld_n;
    const float* kld_src;
    float* kld_dst;
    /* the keyframe the criterion renders (sp_depth_splat's arguments) */
    const uint32_t* pix; const float* baseL; const int32_t* seg_off; const float* kp_L; const float* kld;
    const float* K; const float* kf_pose;
    int32_t N, P;
    unsigned long long* keys; float* depth_out;
    float* rel_pose;                   /* 16 floats scratch */
    float* crit;                       /* 4 floats */
    uint32_t* crit_ws;                 /* sp_kf_criterion_ws_words() uint32 of scratch */
    float* crit_host;                  /* host, pinned: 4 floats */
    float valid_thresh;
    int32_t track_iters, supp_iters;   /* out (host) */
    int32_t pad_;
} SpChainStep;
int sp_chain_step(SpChainStep* step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SP_HIP_H */
