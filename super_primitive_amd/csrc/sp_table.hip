// Once-per-keyframe / once-per-level preparation kernels: mask compaction into the per-segment point table,
// source-image sampling, image packing and the 3x3 binomial pyramid step.
#include "sp_device.h"

namespace {

// one wave per (segment,row): number of set mask bytes in that row.  stride > 1: only the pixels of the stride x stride
// lattice (row and column multiples of the stride) count -- the decimated point sets of sp_prepare_*
__device__ __forceinline__ void row_count(const uint8_t* __restrict__ masks, int row, int H, int W, int stride,
                                          int32_t* __restrict__ row_counts) {
    const int lane = threadIdx.x & 63;
    const uint8_t* m = masks + (size_t)row * W;
    int c = 0;
    if (stride == 1) {
        for (int x = lane; x < W; x += 64) c += m[x] != 0;
    } else if ((row % H) % stride == 0) {
        for (int x = lane * stride; x < W; x += 64 * stride) c += m[x] != 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) row_counts[row] = c;
}

__global__ __launch_bounds__(SP_BLOCK) void k_row_counts(const uint8_t* __restrict__ masks, int rows_total, int H, int W,
                                                         int32_t* __restrict__ row_counts) {
    const int row = blockIdx.x * SP_WAVES + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    row_count(masks, row, H, W, 1, row_counts);
}

// one block per segment: exclusive scan of its H row counts in place, total to counts[n]
__device__ __forceinline__ void segment_row_scan(int32_t* __restrict__ rc, int H, int32_t* __restrict__ count) {
    __shared__ int32_t part[SP_BLOCK];
    const int per = (H + SP_BLOCK - 1) / SP_BLOCK;
    const int r0 = threadIdx.x * per, r1 = min(r0 + per, H);
    int s = 0;
    for (int r = r0; r < r1; ++r) s += rc[r];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < SP_BLOCK; ++i) { const int v = part[i]; part[i] = run; run += v; }
        *count = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int r = r0; r < r1; ++r) { const int v = rc[r]; rc[r] = run; run += v; }
}

__global__ __launch_bounds__(SP_BLOCK) void k_segment_row_scan(int32_t* __restrict__ row_counts, int H,
                                                               int32_t* __restrict__ counts) {
    segment_row_scan(row_counts + (size_t)blockIdx.x * H, H, counts + blockIdx.x);
}

// single block: seg_off = exclusive scan of counts (N+1 entries)
__global__ __launch_bounds__(SP_BLOCK) void k_segment_offsets(const int32_t* __restrict__ counts, int N,
                                                              int32_t* __restrict__ seg_off) {
    __shared__ int32_t part[SP_BLOCK];
    const int per = (N + SP_BLOCK - 1) / SP_BLOCK;
    const int n0 = threadIdx.x * per, n1 = min(n0 + per, N);
    int s = 0;
    for (int n = n0; n < n1; ++n) s += counts[n];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < SP_BLOCK; ++i) { const int v = part[i]; part[i] = run; run += v; }
        seg_off[N] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int n = n0; n < n1; ++n) { seg_off[n] = run; run += counts[n]; }
}

// one wave per (segment,row): ordered compaction of the row (ballot + prefix popcount); stride as in row_count()
__device__ __forceinline__ void fill_row(const uint8_t* __restrict__ masks, const float* __restrict__ logdepth, int row_id,
                                         int H, int W, int stride, const int32_t* __restrict__ seg_off,
                                         const int32_t* __restrict__ row_off, uint32_t* __restrict__ pix,
                                         float* __restrict__ baseL) {
    const int lane = threadIdx.x & 63;
    const int n = row_id / H, r = row_id - n * H;
    if (stride > 1 && r % stride != 0) return;
    const uint8_t* m = masks + (size_t)row_id * W;
    const float* L = logdepth + (size_t)row_id * W;
    int base = seg_off[n] + row_off[row_id];
    for (int x0 = 0; x0 < W; x0 += 64 * stride) {
        const int x = x0 + lane * stride;
        const bool on = (x < W) && (m[x] != 0);
        const unsigned long long bal = __ballot(on);
        if (on) {
            const int k = base + __popcll(bal & ((1ull << lane) - 1ull));
            pix[k] = ((uint32_t)r << 16) | (uint32_t)x;
            if (baseL) baseL[k] = L[x];
        }
        base += __popcll(bal);
    }
}

__global__ __launch_bounds__(SP_BLOCK) void k_table_fill(const uint8_t* __restrict__ masks,
                                                         const float* __restrict__ logdepth, int N, int H, int W,
                                                         const int32_t* __restrict__ seg_off,
                                                         const int32_t* __restrict__ row_off,
                                                         uint32_t* __restrict__ pix, float* __restrict__ baseL) {
    const int row_id = blockIdx.x * SP_WAVES + (threadIdx.x >> 6);
    if (row_id >= N * H) return;
    fill_row(masks, logdepth, row_id, H, W, 1, seg_off, row_off, pix, baseL);
}

// tool/point_utils.py:37-40 + core/dense_optim.py:51-59: L at the rounded keypoint pixel
__device__ __forceinline__ void keypoint_L(const float* __restrict__ logdepth, const float* __restrict__ keypoints, int n, int H,
                                           int W, float* __restrict__ kp_L) {
    const float kr = rintf(0.5f * (float)(H - 1) * (keypoints[2 * n] + 1.f));      // rintf = half-to-even
    const float kc = rintf(0.5f * (float)(W - 1) * (keypoints[2 * n + 1] + 1.f));
    int r = (int)kr, c = (int)kc;
    // torch advanced indexing wraps negative indices once; anything else is an error upstream -- clamp here
    if (r < 0) r += H;
    if (c < 0) c += W;
    r = min(max(r, 0), H - 1);
    c = min(max(c, 0), W - 1);
    kp_L[n] = logdepth[((size_t)n * H + r) * W + c];
}

__global__ void k_keypoint_L(const float* __restrict__ logdepth, const float* __restrict__ keypoints, int N, int H,
                             int W, float* __restrict__ kp_L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) keypoint_L(logdepth, keypoints, n, H, W, kp_L);
}

__device__ __forceinline__ int segment_of(const int32_t* __restrict__ seg_off, int N, int i) {
    int lo = 0, hi = N;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    while (lo + 1 < N && seg_off[lo + 1] <= i) ++lo;
    return lo;
}

struct SourceGeom { float xn, yn, L; bool ok; uint32_t pw; };

// geometry of a table point in its own frame: normalised image coordinates and the validity of the sample
__device__ __forceinline__ SourceGeom source_geometry(uint32_t pix_word, float L, float shift, int H, int W, const float* __restrict__ K9) {
    const float fx = K9[0], cx = K9[2], fy = K9[4], cy = K9[5];
    SourceGeom g;
    g.pw = pix_word & 0x7fffffffu;
    g.L = L;
    const float col = (float)(g.pw & 0xffffu), row = (float)(g.pw >> 16);
    const float d = expf(L + shift);
    const float x = __fdiv_rn(__fmul_rn(__fsub_rn(col, cx), d), fx);
    const float y = __fdiv_rn(__fmul_rn(__fsub_rn(row, cy), d), fy);
    const float zinv = (fabsf(d) > 1e-6f) ? __fdiv_rn(1.0f, d) : 1e-6f;
    const float u = __fadd_rn(__fmul_rn(__fmul_rn(x, fx), zinv), cx);
    const float v = __fadd_rn(__fmul_rn(__fmul_rn(y, fy), zinv), cy);
    const float invW = __fdiv_rn(1.0f, (float)(W - 1)), invH = __fdiv_rn(1.0f, (float)(H - 1));
    g.xn = __fsub_rn(__fmul_rn(__fmul_rn(2.f, u), invW), 1.f);
    g.yn = __fsub_rn(__fmul_rn(__fmul_rn(2.f, v), invH), 1.f);
    g.ok = (fabsf(g.xn) <= 0.99f) && (fabsf(g.yn) <= 0.99f) && (d > 1e-7f);
    return g;
}

// bilinear sample of one pyramid level at that position: {r, g, b, L}.  Taps outside the level read as zero (grid_sample's zeros
// padding); the addresses are clamped into the level and the value selected afterwards, so the loads of a point have no control
// flow between them and those of several points can be in flight together.  The two taps of an image row come from ONE 8-byte
// load (levels at least 2 pixels wide): 6 load instructions per point and level instead of 12.
struct Tap2 { float a, b; };
__device__ __forceinline__ Tap2 load_tap2(const float* p) {
    typedef float f32x2_t __attribute__((ext_vector_type(2), aligned(4)));
    const f32x2_t v = *reinterpret_cast<const f32x2_t*>(p);
    return Tap2{v.x, v.y};
}
__device__ __forceinline__ float4 source_sample(const SourceGeom& g, const float* __restrict__ img, int Hl, int Wl) {
    const float ix = __fmul_rn(__fmul_rn(__fadd_rn(g.xn, 1.f), 0.5f), (float)(Wl - 1));
    const float iy = __fmul_rn(__fmul_rn(__fadd_rn(g.yn, 1.f), 0.5f), (float)(Hl - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx = ix - fx0, wy = iy - fy0;
    const bool xa = x0 >= 0 && x0 < Wl, xb = x0 + 1 >= 0 && x0 + 1 < Wl, ya = y0 >= 0 && y0 < Hl, yb = y0 + 1 >= 0 && y0 + 1 < Hl;
    const int yc0 = min(max(y0, 0), Hl - 1), yc1 = min(max(y0 + 1, 0), Hl - 1);
    float rgb[3];
    if (Wl >= 2) {
        // the pair of columns (xs, xs + 1) inside the row that holds whichever of x0, x0 + 1 exist: x0 itself unless x0 is the last
        // column (then x0 is the pair's second entry) or the column left of the first (then x0 + 1 is its first)
        const int xs = min(max(x0, 0), Wl - 2);
        const bool at = x0 == xs;
        const int i0 = yc0 * Wl + xs, i1 = yc1 * Wl + xs;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* pl = img + (size_t)ch * Hl * Wl;
            const Tap2 t0 = load_tap2(pl + i0), t1 = load_tap2(pl + i1);
            const float nw = (xa && ya) ? (at ? t0.a : t0.b) : 0.f, ne = (xb && ya) ? (at ? t0.b : t0.a) : 0.f;
            const float sw = (xa && yb) ? (at ? t1.a : t1.b) : 0.f, se = (xb && yb) ? (at ? t1.b : t1.a) : 0.f;
            // same weight products and summation order as ATen's grid_sampler_2d (nw, ne, sw, se)
            float acc = nw * ((1.f - wx) * (1.f - wy));
            acc += ne * (wx * (1.f - wy));
            acc += sw * ((1.f - wx) * wy);
            acc += se * (wx * wy);
            rgb[ch] = acc;
        }
        return make_float4(rgb[0], rgb[1], rgb[2], g.L);
    }
    const int xc0 = min(max(x0, 0), Wl - 1), xc1 = min(max(x0 + 1, 0), Wl - 1);
    const int i00 = yc0 * Wl + xc0, i01 = yc0 * Wl + xc1, i10 = yc1 * Wl + xc0, i11 = yc1 * Wl + xc1;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float* pl = img + (size_t)ch * Hl * Wl;
        const float t00 = pl[i00], t01 = pl[i01], t10 = pl[i10], t11 = pl[i11];
        const float nw = (xa && ya) ? t00 : 0.f, ne = (xb && ya) ? t01 : 0.f, sw = (xa && yb) ? t10 : 0.f, se = (xb && yb) ? t11 : 0.f;
        float acc = nw * ((1.f - wx) * (1.f - wy));
        acc += ne * (wx * (1.f - wy));
        acc += sw * ((1.f - wx) * wy);
        acc += se * (wx * wy);
        rgb[ch] = acc;
    }
    return make_float4(rgb[0], rgb[1], rgb[2], g.L);
}

// Source-side sampling exactly as the reference does it (get_pixels on the source's own points,
// core/dense_optim.py:143-162,315-317): IEEE divisions, same operation order, so the validity bit matches.
// The validity bit is a property of the geometry grid (0.99 band on the point's own pixel; depth enters only through the
// last bit of the re-projection): it is written once, when the table is built, and is shared by all pyramid levels --
// later level samplings leave pix untouched.
__device__ __forceinline__ void sample_point(uint32_t* __restrict__ pix, const float* __restrict__ baseL, int i, int n,
                                             const float* __restrict__ kp_L, const float* __restrict__ kld, int H, int W,
                                             const float* __restrict__ img, int Hl, int Wl, const float* __restrict__ K9,
                                             float4* __restrict__ src4, int set_validity) {
    const SourceGeom g = source_geometry(pix[i], baseL[i], kld[n] - kp_L[n], H, W, K9);
    src4[i] = source_sample(g, img, Hl, Wl);
    if (set_validity) pix[i] = g.pw | (g.ok ? 0x80000000u : 0u);
}

__global__ __launch_bounds__(SP_BLOCK) void k_sample_source(uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                                            const int32_t* __restrict__ seg_off,
                                                            const float* __restrict__ kp_L, const float* __restrict__ kld,
                                                            int N, int P, int H, int W, const float* __restrict__ img,
                                                            int Hl, int Wl, const float* __restrict__ K9,
                                                            float4* __restrict__ src4, int set_validity) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= P) return;
    sample_point(pix, baseL, i, segment_of(seg_off, N, i), kp_L, kld, H, W, img, Hl, Wl, K9, src4, set_validity);
}

__device__ __forceinline__ void pack_texel(const float* __restrict__ p, int HW, int i, float* __restrict__ out) {
    float* q = out + (size_t)i * SP_TEXEL_FLOATS;
    q[0] = p[i]; q[1] = p[HW + i]; q[2] = p[2 * (size_t)HW + i];
}

__global__ __launch_bounds__(SP_BLOCK) void k_pack_rgb(const float* __restrict__ chw, int HW, float* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= HW) return;
    pack_texel(chw + (size_t)blockIdx.y * 3 * HW, HW, i, out + (size_t)blockIdx.y * HW * SP_TEXEL_FLOATS);
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// image/gaussian_pyramid.py:53-85: reflect pad 1, depthwise [1 2 1;2 4 2;1 2 1]/16, keep [::2, ::2]
__device__ __forceinline__ float blur_decimate_at(const float* __restrict__ p, int H, int W, int Wo, int i) {
    const int yo = i / Wo, xo = i - yo * Wo;
    const float wgt[3] = {1.f, 2.f, 1.f};
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = reflect1(2 * yo + dy - 1, H);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int x = reflect1(2 * xo + dx - 1, W);
            acc = fmaf(wgt[dy] * wgt[dx] * (1.f / 16.f), p[(size_t)y * W + x], acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(SP_BLOCK) void k_blur_decimate(const float* __restrict__ in, int H, int W, int Ho, int Wo,
                                                            float* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= Ho * Wo) return;
    out[(size_t)blockIdx.y * Ho * Wo + i] = blur_decimate_at(in + (size_t)blockIdx.y * H * W, H, W, Wo, i);
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched preparation (sp_prepare_*): the same per-row / per-point work, one launch for MANY keyframes -- blockIdx.y picks
// the job record.  A pipeline that optimises hundreds of new frame pairs per batch is otherwise bound by the ~15 tiny
// launches per pair above, not by the optimiser.
// ---------------------------------------------------------------------------------------------------------------------
// Device views of the job records (include/sp_hip.h): the same bytes, the pointers typed as global memory.  A pointer read out
// of a record is otherwise a generic pointer to the compiler -- flat_load / flat_store, 64-bit address arithmetic in vector
// registers for every access, vmcnt and lgkmcnt both held -- and every access of these passes goes through one.
#define SP_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T* generic(SP_GLOBAL T* p) { return (T*)p; }      // for the helpers shared with the per-keyframe kernels
__device__ __forceinline__ uint4 load4(const SP_GLOBAL u32x4* p) { const u32x4 v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 load4(const SP_GLOBAL f32x4* p) { const f32x4 v = *p; return make_float4(v.x, v.y, v.z, v.w); }
// ... of data that is read exactly once (the masks): non-temporal, so that 20 MB per keyframe do not push the row counts and bit words
// the next passes read out of the L2
__device__ __forceinline__ uint4 load4_once(const SP_GLOBAL u32x4* p) { const u32x4 v = __builtin_nontemporal_load(p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void store4(SP_GLOBAL f32x4* p, const float4& v) { const f32x4 w = {v.x, v.y, v.z, v.w}; *p = w; }

struct PrepTable {
    const SP_GLOBAL uint8_t* masks;
    const SP_GLOBAL float* logdepth;
    const SP_GLOBAL float* keypoints;
    SP_GLOBAL float* kp_L;
    SP_GLOBAL int32_t* row_counts[SP_PREP_MAX_STRIDES];
    SP_GLOBAL int32_t* counts[SP_PREP_MAX_STRIDES];
    const SP_GLOBAL int32_t* seg_off[SP_PREP_MAX_STRIDES];
    SP_GLOBAL uint32_t* pix[SP_PREP_MAX_STRIDES];
    SP_GLOBAL float* baseL[SP_PREP_MAX_STRIDES];
    int32_t stride[SP_PREP_MAX_STRIDES];
    int32_t N, H, W, n_strides;
    SP_GLOBAL uint32_t* bits;
    const SP_GLOBAL int32_t* boxes;
};
struct PrepSample {
    SP_GLOBAL uint32_t* pix;
    const SP_GLOBAL float* baseL;
    const SP_GLOBAL int32_t* seg_off;
    const SP_GLOBAL int32_t* counts;
    const SP_GLOBAL float* kp_L;
    const SP_GLOBAL float* kld;
    const SP_GLOBAL float* K;
    const SP_GLOBAL float* image[SP_PREP_MAX_LEVELS];
    SP_GLOBAL float* src4[SP_PREP_MAX_LEVELS];
    int32_t Hl[SP_PREP_MAX_LEVELS], Wl[SP_PREP_MAX_LEVELS];
    int32_t N, P, H, W, n_levels;
    int32_t granule;
};
struct PrepImage {
    const SP_GLOBAL float* in;
    SP_GLOBAL float* out;
    int32_t H, W;
};
static_assert(sizeof(PrepTable) == sizeof(SpPrepTable) && sizeof(PrepSample) == sizeof(SpPrepSample) && sizeof(PrepImage) == sizeof(SpPrepImage),
              "device views of the job records");
static_assert(offsetof(PrepTable, bits) == offsetof(SpPrepTable, bits) && offsetof(PrepTable, stride) == offsetof(SpPrepTable, stride)
              && offsetof(PrepSample, granule) == offsetof(SpPrepSample, granule) && offsetof(PrepSample, src4) == offsetof(SpPrepSample, src4), "device views of the job records");
__device__ __forceinline__ const PrepTable& table_of(const SpPrepTable* tables, int i) { return reinterpret_cast<const PrepTable*>(tables)[i]; }

// bit 7 of every byte of w that is non-zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) { return ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u); }

// which of the 4 pixels x0 .. x0+3 (x0 a multiple of 4) lie on the stride lattice, as a nonzero_bytes() mask
__device__ __forceinline__ uint32_t lattice_bytes(int x0, int stride) {
    if (stride == 1) return 0x80808080u;
    if (stride == 2) return 0x00800080u;
    if (stride == 4) return 0x00000080u;
    uint32_t sel = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) sel |= ((x0 + j) % stride == 0) ? (0x80u << (8 * j)) : 0u;
    return sel;
}

// Mask word of 16 consecutive pixels (one 16-byte piece of a mask row) as written by the count pass and consumed by the fill
// pass: pixel 4 q + j of the piece sits at bit 8 j + q, i.e. m = (nz0 >> 7) | (nz1 >> 6) | (nz2 >> 5) | (nz3 >> 4) with nz_q =
// nonzero_bytes(word q) -- four shifts and two three-input ORs, and ((m >> q) & 0x01010101) << 7 gives nz_q back.  Two bits of
// storage per pixel (the bits 4-7 of every byte stay clear); the fill pass then reads 4 bytes per 16 pixels instead of 16.
__device__ __forceinline__ uint32_t piece_bits(uint32_t nz0, uint32_t nz1, uint32_t nz2, uint32_t nz3) {
    return (nz0 >> 7) | (nz1 >> 6) | (nz2 >> 5) | (nz3 >> 4);
}
// the pixels of a 16-pixel piece that lie on the stride lattice, in piece_bits() layout (strides that divide 16 only)
__device__ __forceinline__ uint32_t lattice_piece(int stride) {
    return stride == 1 ? 0x0f0f0f0fu : (stride == 2 ? 0x000f000fu : (stride == 4 ? 0x0000000fu : (stride == 8 ? 0x00000005u : 0x00000001u)));
}

// Every lattice of the keyframe in the same pass over the masks.  Fast path (rows of W = 16 k <= 1024 pixels, 16-byte aligned,
// lattice strides 1 / 2 / 4 / 8 / 16): a wave takes SP_PREP_WAVE_ROWS consecutive (segment,row) rows -- ONE contiguous run of
// 16-byte pieces in memory -- and its lanes walk that run densely, a piece per lane and trip, whatever the row length (a lane per
// piece of ONE row left 24 of 64 lanes idle at W = 640, and the pass was bound by the vector ALU: 97 % busy at 0.60 of the HBM
// roofline).  Per piece: one packed bit word (written to SpPrepTable.bits: the fill pass reads those instead of the masks) and
// its four lattice counts as the bytes of one word in LDS; then four lanes per row add up the row's words.
#define SP_PREP_WAVE_ROWS 16
#define SP_PREP_LOADS 8             /* 16-byte loads a lane has in flight */
// does this keyframe take the fast path of the count pass (and, if it has a bits array, of the fill pass)?
__device__ __forceinline__ bool prep_fast_path(const PrepTable& t) {
    bool fixed = true;
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) {
        const int st = k < t.n_strides ? t.stride[k] : 1;
        fixed = fixed && (st == 1 || st == 2 || st == 4 || st == 8 || st == 16);
    }
    return fixed && (t.W & 15) == 0 && ((uintptr_t)t.masks & 15) == 0 && t.W <= 1024;
}

// What the count pass needs of a keyframe's record, as a PRIVATE copy made once per workgroup: through a reference to the record in device
// memory the compiler re-reads a field after every store (it cannot know the stores leave the record alone) -- a chain of scalar-load round
// trips inside every block of rows, and most of a block's time when the segment boxes leave it 96 pieces to read.
struct CountTable {
    const SP_GLOBAL uint8_t* masks;
    const SP_GLOBAL float* logdepth;
    SP_GLOBAL int32_t* row_counts[SP_PREP_MAX_STRIDES];
    int32_t stride[SP_PREP_MAX_STRIDES];
    int32_t N, H, W, n_strides;
    SP_GLOBAL uint32_t* bits;
    const SP_GLOBAL int32_t* boxes;
};
__device__ __forceinline__ CountTable count_table(const PrepTable& g) {
    CountTable t;
    t.masks = g.masks; t.logdepth = g.logdepth; t.N = g.N; t.H = g.H; t.W = g.W; t.n_strides = g.n_strides; t.bits = g.bits; t.boxes = g.boxes;
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) { t.row_counts[k] = g.row_counts[k]; t.stride[k] = g.stride[k]; }
    return t;
}

// (one block of SP_WAVES x SP_PREP_WAVE_ROWS rows; every early return is taken by the whole workgroup)
__device__ __forceinline__ void prep_row_counts_block(const CountTable& t, int vb) {
    const int rows = t.N * t.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qpr = t.W >> 4;
    constexpr int BLOCK_ROWS = SP_WAVES * SP_PREP_WAVE_ROWS;
    if (vb * BLOCK_ROWS >= rows) return;
    // (the box hint only counts pixels INSIDE the boxes; the fill pass honours that only on its bit-word path (it then fills
    //  from the bit words this pass writes) -- a keyframe that k_prep_fill will scan mask by mask ignores the hint, so that a mask pixel
    //  outside its box, a contract violation, is still counted before it is filled)
    const bool fill_reads_bits = t.bits && ((uintptr_t)t.logdepth & 15) == 0 && (long long)t.N * t.H < (1ll << 22) && t.H <= 1024;
    const SP_GLOBAL int32_t* const boxes = fill_reads_bits ? t.boxes : nullptr;
    if (boxes && t.H >= BLOCK_ROWS) {
        // SEGMENT BOXES (SpPrepTable.boxes), the whole workgroup first (its 64 rows lie in at most two segments): most workgroups of
        // a boxed keyframe are outside every box -- a segment covers a small part of the image -- and leave with three coalesced
        // stores of zero counts, before any LDS
        const int B0 = vb * BLOCK_ROWS, B1 = min(B0 + BLOCK_ROWS, rows) - 1;
        if (B0 >= rows) return;
        const int H = t.H, n0 = B0 / H, n1 = B1 / H;
        bool any = false;
        {
            const int lo = B0 - n0 * H, hi = (n1 == n0 ? B1 - n0 * H : H - 1) + 1;      // rows [lo, hi) of segment n0
            const int r0 = boxes[4 * n0], c0 = boxes[4 * n0 + 1], r1 = boxes[4 * n0 + 2], c1 = boxes[4 * n0 + 3];
            any = max(r0, lo) < min(r1, hi) && max(c0, 0) < min(c1, t.W);
        }
        if (n1 != n0) {
            const int hi = B1 - n1 * H + 1;                                           // rows [0, hi) of segment n1
            const int r0 = boxes[4 * n1], c0 = boxes[4 * n1 + 1], r1 = boxes[4 * n1 + 2], c1 = boxes[4 * n1 + 3];
            any = any || (max(r0, 0) < min(r1, hi) && max(c0, 0) < min(c1, t.W));
        }
        if (!any) {
            const int row = B0 + (int)threadIdx.x;
            if ((int)threadIdx.x < BLOCK_ROWS && row < rows) {
#pragma unroll
                for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
                    if (k < t.n_strides) t.row_counts[k][row] = 0;
            }
            if (t.bits && t.stride[0] != 1)      // (the bit words of empty rows are only read when lattice 0 is not the full one: see below)
                for (int p = threadIdx.x; p < (B1 - B0 + 1) * qpr; p += SP_BLOCK) t.bits[(size_t)B0 * qpr + p] = 0u;
            return;
        }
    }
    const int row_base = (vb * SP_WAVES + wave) * SP_PREP_WAVE_ROWS;
    const int n_rows = min(SP_PREP_WAVE_ROWS, rows - row_base);          // (<= 0: a wave past the end, which still meets the barrier)
    // ... then per wave: lane r < n_rows holds the box of the wave's row r -- is the row inside its segment's box rows, and which
    // columns.  A wave none of whose rows is inside any box reads nothing and writes zero counts; the others walk only the 16-pixel
    // pieces [q0, q1) of their rows that some box of theirs meets (the rest of those rows' bit words are written as zeros), and of
    // those load the ones the row's own box meets.
    int my_c0 = 0, my_c1 = t.W;
    unsigned long long row_in = ~0ull;
    int q0 = 0, q1 = qpr;
    if (boxes) {
        bool in = false;
        if (lane < n_rows) {
            const int row = row_base + lane, H = t.H;
            int n = (int)((float)row * (1.f / (float)H)), r = row - n * H;      // (rows < 2^22 on this path: the quotient is off by one at most)
            if (r < 0) { --n; r += H; }
            if (r >= H) { ++n; r -= H; }
            const int r0 = boxes[4 * n], c0 = boxes[4 * n + 1], r1 = boxes[4 * n + 2], c1 = boxes[4 * n + 3];
            my_c0 = max(c0, 0); my_c1 = min(c1, t.W);
            in = r >= r0 && r < r1 && my_c0 < my_c1;
        }
        row_in = __ballot(in);
        int lo = in ? (my_c0 >> 4) : qpr, hi = in ? ((my_c1 + 15) >> 4) : 0;
#pragma unroll
        for (int o = 1; o < SP_PREP_WAVE_ROWS; o <<= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
        q0 = __builtin_amdgcn_readfirstlane(lo); q1 = __builtin_amdgcn_readfirstlane(hi);
    }
    const bool skip = row_in == 0ull;
    const int nq = skip ? 1 : q1 - q0;                                       // pieces walked per row
    const int n_pieces = skip ? 0 : n_rows * nq;
    const uint32_t q_magic = (1u << 20) / (uint32_t)nq + 1u;                 // p / nq = (p * q_magic) >> 20 for p < 2^20 / nq (p < 1024 here)
    __shared__ uint32_t s_c[SP_WAVES][SP_PREP_WAVE_ROWS * 64];
    uint32_t sel[SP_PREP_MAX_STRIDES];
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) sel[k] = k < t.n_strides ? lattice_piece(t.stride[k]) : 0u;
    const SP_GLOBAL u32x4* mq = (const SP_GLOBAL u32x4*)t.masks + (size_t)max(row_base, 0) * qpr;
    SP_GLOBAL uint32_t* bits = t.bits ? t.bits + (size_t)row_base * qpr : nullptr;
    if (boxes && bits && !skip && lane < qpr && (lane < q0 || lane >= q1)) {      // the walked rows' bit words outside [q0, q1): zeros
        for (int r = 0; r < n_rows; ++r) bits[r * qpr + lane] = 0u;
    }
    // The bit words of EMPTY rows are never read when lattice 0 is the full one (the fill pass finds those rows empty through their counts):
    // without boxes -- every piece of every row is walked -- they are 85 % of the words, 1.6 of the pass's 9.5 GB per 384 keyframes.  A
    // row's emptiness is known only once all its pieces are counted, so the words are KEPT in registers (at most 1024 pieces per wave: 16
    // per lane) and stored after the row sums, for the non-empty rows only.
    constexpr int TRIPS = (SP_PREP_WAVE_ROWS * 64 + SP_PREP_LOADS * 64 - 1) / (SP_PREP_LOADS * 64);
    const bool lazy_bits = bits && !boxes && t.stride[0] == 1;               // (wave-uniform)
    uint32_t mk[TRIPS][SP_PREP_LOADS];
#pragma unroll
    for (int tr = 0; tr < TRIPS; ++tr) {
        const int p0 = tr * SP_PREP_LOADS * 64;
#pragma unroll
        for (int u = 0; u < SP_PREP_LOADS; ++u) mk[tr][u] = 0u;
        if (p0 >= n_pieces) continue;
        uint4 w[SP_PREP_LOADS];
#pragma unroll
        for (int u = 0; u < SP_PREP_LOADS; ++u) {
            const int p = p0 + u * 64 + lane;
            bool want = p < n_pieces;
            int at = p;                                                      // the piece's index in the wave's run of rows
            if (boxes) {
                const int row_l = min((int)(((uint32_t)min(p, 1023) * q_magic) >> 20), SP_PREP_WAVE_ROWS - 1), xq = q0 + (p - row_l * nq);
                const int c0 = __shfl(my_c0, row_l, 64), c1 = __shfl(my_c1, row_l, 64);
                want = want && ((row_in >> row_l) & 1ull) != 0ull && 16 * xq < c1 && 16 * xq + 16 > c0;
                at = row_l * qpr + xq;
            }
            w[u] = want ? load4_once(mq + at) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < SP_PREP_LOADS; ++u) {
            const int p = p0 + u * 64 + lane;
            if (p0 + u * 64 >= n_pieces) break;
            const uint32_t m = piece_bits(nonzero_bytes(w[u].x), nonzero_bytes(w[u].y), nonzero_bytes(w[u].z), nonzero_bytes(w[u].w));
            if (p < n_pieces) {
                mk[tr][u] = m;
                if (bits && !lazy_bits) {
                    int at = p;
                    if (boxes) { const int row_l = (int)(((uint32_t)p * q_magic) >> 20); at = row_l * qpr + q0 + (p - row_l * nq); }
                    bits[at] = m;
                }
                s_c[wave][p] = (uint32_t)__popc(m & sel[0]) | ((uint32_t)__popc(m & sel[1]) << 8) | ((uint32_t)__popc(m & sel[2]) << 16)
                               | ((uint32_t)__popc(m & sel[3]) << 24);
            }
        }
    }
    // (rows of a skipped wave are found empty through lattice 0's row counts and their bit words never read -- unless the first lattice
    //  is not the full one: then the fill pass looks at every row, and the words must be there)
    if (skip && bits && t.stride[0] != 1)
        for (int p = lane; p < n_rows * qpr; p += 64) bits[p] = 0u;
    __syncthreads();
    // four lanes per row: bytes -> two words of 16-bit fields (a row's count is at most 1024), summed over the row's walked pieces
    const int row_l = lane >> 2, sub = lane & 3;
    uint32_t a0 = 0u, a1 = 0u;
    if (row_l < n_rows && !skip)
        for (int i = sub; i < nq; i += 4) {
            const uint32_t c = s_c[wave][row_l * nq + i];
            a0 += (c & 0xffu) | ((c & 0xff00u) << 8);
            a1 += ((c >> 16) & 0xffu) | ((c >> 8) & 0xff0000u);
        }
    a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
    a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
    if (sub == 0 && row_l < n_rows) {
        const int row = row_base + row_l, r = row % t.H;
        const int cnt[SP_PREP_MAX_STRIDES] = {(int)(a0 & 0xffffu), (int)(a0 >> 16), (int)(a1 & 0xffffu), (int)(a1 >> 16)};
#pragma unroll
        for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
            if (k < t.n_strides) t.row_counts[k][row] = (r % t.stride[k] == 0) ? cnt[k] : 0;
    }
    if (lazy_bits) {
        // (bit 4 r of the ballot: row r of the wave holds a pixel -- lattice 0 is the full one here, its count is the row's)
        const unsigned long long full = __ballot(sub == 0 && row_l < n_rows && (a0 & 0xffffu) != 0u);
        if (full != 0ull) {
#pragma unroll
            for (int tr = 0; tr < TRIPS; ++tr)
#pragma unroll
                for (int u = 0; u < SP_PREP_LOADS; ++u) {
                    const int p = tr * SP_PREP_LOADS * 64 + u * 64 + lane;
                    if (p < n_pieces) {
                        const int rl = (int)(((uint32_t)p * q_magic) >> 20);
                        if ((full >> (4 * rl)) & 1ull) bits[p] = mk[tr][u];
                    }
                }
        }
    }
}

__global__ __launch_bounds__(SP_BLOCK) void k_prep_row_counts(const SpPrepTable* __restrict__ tables) {
    const PrepTable& g = table_of(tables, blockIdx.y);
    if (!prep_fast_path(g)) return;          // (k_prep_row_counts_general takes those keyframes)
    const CountTable t = count_table(g);
    prep_row_counts_block(t, (int)blockIdx.x);
}

// The count pass of BOXED keyframes (sp_prepare_count_boxed; round 6, late): ONE WORKGROUP PER SEGMENT.  With the segment-box hint
// k_prep_row_counts reads 1.15 MB of a keyframe's 19.7 MB of masks, but stays organised by blocks of 64 (segment,row) rows: 480 workgroups per
// keyframe, most of which leave after a test behind two dependent scalar loads, the others walking 16 rows x ~6 pieces per wave behind three
// more round trips -- 0.33-0.49 ms per 384 keyframes at 0.10 of the HBM roofline, bound by the LATENCY of its chains (several blocks per
// workgroup, tested a block per lane at once: 0.31-0.42 ms; a private copy of the record against scalar re-loads: no change).  A segment's
// box is one rectangle: its workgroup loads the box, has ALL the rectangle's 16-byte pieces in flight at once (68 rows x 7 pieces for a grid
// segment: two loads per thread), writes their bit words (zeros for the rest of the box rows' words: the fill pass reads whole rows) and adds
// the pieces' lattice counts to their rows' counters in LDS (integer adds: any order gives the same sums), then stores the counts of the
// segment's H rows -- zeros outside the box rows.  Same row counts, same bit words as k_prep_row_counts, bit for bit.  A keyframe without
// boxes (or whose fill pass does not read the bit words) takes the whole frame as every segment's box: correct, but meant for batches whose
// keyframes all carry boxes.
__global__ __launch_bounds__(SP_BLOCK) void k_prep_row_counts_boxed(const SpPrepTable* __restrict__ tables) {
    const PrepTable& g = table_of(tables, blockIdx.y);
    if (!prep_fast_path(g)) return;          // (k_prep_row_counts_general takes those keyframes)
    const CountTable t = count_table(g);
    const int n = (int)blockIdx.x;
    if (n >= t.N) return;
    const int H = t.H, W = t.W, qpr = W >> 4;
    const bool fill_reads_bits = t.bits && ((uintptr_t)t.logdepth & 15) == 0 && (long long)t.N * H < (1ll << 22) && H <= 1024;
    int r0 = 0, r1 = H, c0 = 0, c1 = W;
    if (fill_reads_bits && t.boxes) {
        r0 = min(max(t.boxes[4 * n], 0), H); c0 = max(t.boxes[4 * n + 1], 0);
        r1 = min(max(t.boxes[4 * n + 2], r0), H); c1 = min(t.boxes[4 * n + 3], W);
        if (c0 >= c1) r1 = r0;               // (an empty box: no row is inside)
    }
    const int q0 = c0 >> 4, q1 = r1 > r0 ? (c1 + 15) >> 4 : q0, nq = q1 - q0;
    constexpr int CHUNK = 1024;              // box rows per pass over the LDS counters (one pass unless H > 1024)
    __shared__ uint32_t s_a[2][CHUNK];       // per row of the box: the four lattice counts as 16-bit fields of two words (a row holds <= 1024 pixels)
    uint32_t sel[SP_PREP_MAX_STRIDES];
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) sel[k] = k < t.n_strides ? lattice_piece(t.stride[k]) : 0u;
    const size_t row0 = (size_t)n * H;
    const SP_GLOBAL u32x4* const mq = (const SP_GLOBAL u32x4*)t.masks + row0 * qpr;
    SP_GLOBAL uint32_t* const bits = t.bits ? t.bits + row0 * qpr : nullptr;
    // the box rows' bit words outside [q0, q1): zeros (rows outside the box are found empty through lattice 0's row counts and their words are
    // never read -- unless the first lattice is not the full one: then every row's words must be there)
    if (bits) {
        const bool all_rows = t.stride[0] != 1;
        const int ra = all_rows ? 0 : r0, rb = all_rows ? H : r1;
        for (int p = threadIdx.x; p < (rb - ra) * qpr; p += SP_BLOCK) {
            const int r = ra + p / qpr, xq = p - (r - ra) * qpr;
            if (r < r0 || r >= r1 || xq < q0 || xq >= q1) bits[(size_t)r * qpr + xq] = 0u;
        }
    }
    // the rows outside the box: zero counts
    for (int r = threadIdx.x; r < H; r += SP_BLOCK) {
        if (r >= r0 && r < r1) continue;
#pragma unroll
        for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
            if (k < t.n_strides) t.row_counts[k][row0 + r] = 0;
    }
    const uint32_t nq_magic = nq > 1 ? (0xffffffffu / (uint32_t)nq + 1u) : 0u;
    for (int ra = r0; ra < r1; ra += CHUNK) {
        const int rows_c = min(CHUNK, r1 - ra), n_pieces = rows_c * nq;
        __syncthreads();                     // (the previous chunk's counters have been read)
        for (int r = threadIdx.x; r < rows_c; r += SP_BLOCK) { s_a[0][r] = 0u; s_a[1][r] = 0u; }
        __syncthreads();
        for (int p0 = 0; p0 < n_pieces; p0 += SP_PREP_LOADS * SP_BLOCK) {
            uint4 w[SP_PREP_LOADS];
            int rl[SP_PREP_LOADS], xq[SP_PREP_LOADS];
#pragma unroll
            for (int u = 0; u < SP_PREP_LOADS; ++u) {
                const int p = p0 + u * SP_BLOCK + (int)threadIdx.x;
                const int pc = min(p, n_pieces - 1);
                int q = pc;                                      // row of the piece inside the chunk: pc / nq
                if (nq > 1) {
                    q = (int)__umulhi((uint32_t)pc, nq_magic);
                    if ((q + 1) * nq <= pc) ++q;                 // (the magic quotient is exact or one off)
                    if (q * nq > pc) --q;
                }
                rl[u] = q; xq[u] = q0 + (pc - q * nq);
                w[u] = p < n_pieces ? load4_once(mq + (size_t)(ra + rl[u]) * qpr + xq[u]) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < SP_PREP_LOADS; ++u) {
                const int p = p0 + u * SP_BLOCK + (int)threadIdx.x;
                if (p < n_pieces) {
                    const uint32_t m = piece_bits(nonzero_bytes(w[u].x), nonzero_bytes(w[u].y), nonzero_bytes(w[u].z), nonzero_bytes(w[u].w));
                    if (bits) bits[(size_t)(ra + rl[u]) * qpr + xq[u]] = m;
                    const uint32_t a0 = (uint32_t)__popc(m & sel[0]) | ((uint32_t)__popc(m & sel[1]) << 16);
                    const uint32_t a1 = (uint32_t)__popc(m & sel[2]) | ((uint32_t)__popc(m & sel[3]) << 16);
                    if (a0) atomicAdd(&s_a[0][rl[u]], a0);
                    if (a1) atomicAdd(&s_a[1][rl[u]], a1);
                }
            }
        }
        __syncthreads();
        for (int rr = threadIdx.x; rr < rows_c; rr += SP_BLOCK) {
            const int r = ra + rr;
            const uint32_t a0 = s_a[0][rr], a1 = s_a[1][rr];
            const int cnt[SP_PREP_MAX_STRIDES] = {(int)(a0 & 0xffffu), (int)(a0 >> 16), (int)(a1 & 0xffffu), (int)(a1 >> 16)};
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
                if (k < t.n_strides) t.row_counts[k][row0 + r] = (r % t.stride[k] == 0) ? cnt[k] : 0;
        }
    }
}

// General path (any width, alignment and stride): SP_PREP_ROWS rows at a time, word or byte loads.  Its own kernel -- the two
// paths in one cost the fast one its occupancy (138 vector registers against 56) -- on a bounded grid per keyframe whose waves
// walk the rows, so that the launch costs nothing when every keyframe is on the fast path.
#define SP_PREP_GENERAL_BLOCKS 64
#define SP_PREP_ROWS 4
__global__ __launch_bounds__(SP_BLOCK) void k_prep_row_counts_general(const SpPrepTable* __restrict__ tables) {
    const PrepTable& t = table_of(tables, blockIdx.y);
    if (prep_fast_path(t)) return;
    const int rows = t.N * t.H;
    const int lane = threadIdx.x & 63;
    const int wpr = t.W >> 2;
    for (int row0 = (blockIdx.x * SP_WAVES + (threadIdx.x >> 6)) * SP_PREP_ROWS; row0 < rows; row0 += gridDim.x * SP_WAVES * SP_PREP_ROWS) {
        float acc[SP_PREP_ROWS * SP_PREP_MAX_STRIDES];
#pragma unroll
        for (int i = 0; i < SP_PREP_ROWS * SP_PREP_MAX_STRIDES; ++i) acc[i] = 0.f;
        if ((t.W & 3) == 0 && ((uintptr_t)t.masks & 3) == 0 && wpr <= 256) {
            const SP_GLOBAL uint32_t* mw = (const SP_GLOBAL uint32_t*)t.masks + (size_t)row0 * wpr;
            uint32_t w[SP_PREP_ROWS][4];
#pragma unroll
            for (int rr = 0; rr < SP_PREP_ROWS; ++rr)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xw = q * 64 + lane;
                    w[rr][q] = (row0 + rr < rows && xw < wpr) ? mw[(size_t)rr * wpr + xw] : 0u;
                }
#pragma unroll
            for (int rr = 0; rr < SP_PREP_ROWS; ++rr)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t nz = nonzero_bytes(w[rr][q]);
#pragma unroll
                    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
                        acc[rr * SP_PREP_MAX_STRIDES + k] += (float)__popc(nz & (k < t.n_strides ? lattice_bytes(4 * (q * 64 + lane), t.stride[k]) : 0u));
                }
        } else {
#pragma unroll
            for (int rr = 0; rr < SP_PREP_ROWS; ++rr) {
                if (row0 + rr >= rows) break;
                const SP_GLOBAL uint8_t* m = t.masks + (size_t)(row0 + rr) * t.W;
                for (int x = lane; x < t.W; x += 64) {
                    const bool on = m[x] != 0;
#pragma unroll
                    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
                        if (k < t.n_strides) acc[rr * SP_PREP_MAX_STRIDES + k] += (on && x % t.stride[k] == 0) ? 1.f : 0.f;
                }
            }
        }
        int pos;
        bool ok;
        wave_sum_to_lanes<SP_PREP_ROWS * SP_PREP_MAX_STRIDES>(acc, lane, pos, ok);
        const int rr = pos / SP_PREP_MAX_STRIDES, k = pos % SP_PREP_MAX_STRIDES, row = row0 + rr;
        if (ok && k < t.n_strides && row < rows) t.row_counts[k][row] = ((row % t.H) % t.stride[k] == 0) ? (int)acc[0] : 0;
    }
}

// inclusive prefix sum over the 64 lanes of a wave in six DPP additions (row shifts by 1 / 2 / 4 / 8 inside the rows of 16 lanes, then
// lane 15 of a row broadcast into the next row, lane 31 into the upper half): no LDS traffic, no barrier
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

// Exclusive scan of the H row counts of every (segment, lattice) in place, total to counts[n]: one WAVE each, 64 rows per trip, the
// prefix sum across the lanes by DPP, the carry in a scalar -- no LDS, no barrier.  (A workgroup per segment with thread 0 adding up 256
// partial sums one after the other -- the per-keyframe kernel's form -- took 45 us per 128 keyframes, 7 % of the count pass.)
__global__ __launch_bounds__(SP_BLOCK) void k_prep_row_scan(const SpPrepTable* __restrict__ tables) {
    const PrepTable& t = table_of(tables, blockIdx.y);
    const int lane = threadIdx.x & 63, n = blockIdx.x * SP_WAVES + (threadIdx.x >> 6);
    const int k = blockIdx.z;
    if (n >= t.N || k >= t.n_strides) return;
    const int H = t.H;
    SP_GLOBAL int32_t* rc = t.row_counts[k] + (size_t)n * H;
    uint32_t carry = 0u;
    if (H <= 512) {                  // all trips' loads in flight together
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (64 * i + lane < H) ? (uint32_t)rc[64 * i + lane] : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (64 * i >= H) break;
            const uint32_t incl = wave_inclusive_scan(v[i]);
            if (64 * i + lane < H) rc[64 * i + lane] = (int32_t)(carry + incl - v[i]);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    } else {
        for (int r0 = 0; r0 < H; r0 += 64) {
            const int r = r0 + lane;
            const uint32_t v = r < H ? (uint32_t)rc[r] : 0u;
            const uint32_t incl = wave_inclusive_scan(v);
            if (r < H) rc[r] = (int32_t)(carry + incl - v);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    if (lane == 0) t.counts[k][n] = (int32_t)carry;
}

#define SP_FILL_ROWS 256

// does the fill pass of this keyframe read the bit words the count pass wrote (k_prep_fill_bits) instead of the masks (k_prep_fill)?
// (k_prep_row_counts spells the same condition out for its use of the segment boxes)
__device__ __forceinline__ bool prep_fill_bits_path(const PrepTable& t) {
    return t.bits && prep_fast_path(t) && ((uintptr_t)t.logdepth & 15) == 0 && (long long)t.N * t.H < (1ll << 22) && t.H <= 1024;
}

// the set pixels i = 0, STEP, 2 STEP, ... of an octet, in order, into the staging list from position q on
template <int STEP>
__device__ __forceinline__ void stage_octet(uint2* stage, uint32_t sel, int q, uint32_t pw0, const float (&Lq)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 8; i += STEP) {
        const uint32_t bit = (sel >> i) & 1u;
        stage[bit ? q : 512 + lane] = make_uint2(pw0 + (uint32_t)i, __float_as_uint(Lq[i]));
        q += (int)bit;
    }
}

// Ordered compaction on the bit words of the count pass, by OCTETS (8 pixels = half a bit word), every WAVE on its own rows.
// A row at a time -- a lane per 4 pixels, ballots for the ranks -- is ~330 instructions per row of which a segment fills a fifth of
// the lanes (bound by instruction issue at 0.11-0.19 of the HBM roofline); octets of 32 rows through block-wide stages (list, scan
// across the four waves, deltas: five barriers per 256 octets) left the waves parked three quarters of the time (0.25).  The start
// of every ROW in every lattice's table is known from the row counts, so rows -- and waves -- are independent: a wave takes every
// fourth row of the workgroup's SP_FILL_ROWS (a segment's non-empty rows are consecutive: they spread evenly over the waves),
// keeps the non-empty ones, and runs batches of SP_FILL_WAVE_ROWS of them through (a) the rows' bit words into LDS, (b) the ordered
// list of the batch's non-empty octets (ballots), (c) 64 listed octets at a time, one per lane: the log-depths of the octet (two
// 16-byte loads, requested one chunk ahead), its point counts per lattice, a wave-wide prefix sum of those (DPP) = the rank of its
// first point in list order, minus the value at the row's first octet = the rank inside the row, plus the row's start = where its
// points go, in torch.where order.  All LDS is private to the wave: no barrier anywhere.
#define SP_FILL_WAVE_ROWS 16
__global__ __launch_bounds__(SP_BLOCK) void k_prep_fill_bits(const SpPrepTable* __restrict__ tables) {
    const PrepTable& t = table_of(tables, blockIdx.y);
    if (!prep_fill_bits_path(t)) return;
    const int rows = t.N * t.H;
    const int row_base = blockIdx.x * SP_FILL_ROWS;
    if (row_base >= rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RB = SP_FILL_WAVE_ROWS;
    constexpr int LIST = 1152;                                        // (14 rows of 640 pixels a batch; fewer when they are longer)
    __shared__ uint16_t s_list[SP_WAVES][LIST];                       // octet = lane that holds the row << 7 | octet of the row
    __shared__ int s_delta[SP_WAVES][SP_PREP_MAX_STRIDES][64];        // per row (by the lane that holds it): table position minus list rank
    __shared__ uint2 s_stage[SP_WAVES][512 + 64];                     // the points of a chunk of 64 octets in list order {pixel word, log-depth}; + a slot per lane for the unset pixels
    static_assert(sizeof(uint16_t) * SP_WAVES * LIST + sizeof(int) * SP_WAVES * SP_PREP_MAX_STRIDES * 64 + sizeof(uint2) * SP_WAVES * (512 + 64) <= 32768,
                  "five workgroups per CU");
    const unsigned long long below = (1ull << lane) - 1ull;
    // One lane per row: is it empty?  The row's start in every lattice's table is requested together with the counts that say so.
    // What the pass needs of a row -- its index, its row of the image, its starts -- STAYS in the registers of that lane and is
    // fetched by lane index (readlane for the batch's loads, ds_bpermute per listed octet): no LDS arrays for it (6 KB a workgroup).
    uint32_t my_row;                                 // row | row of the image << 22
    int my_b[SP_PREP_MAX_STRIDES];                   // first table position of the row per lattice, -1: not on the lattice
    unsigned long long rem;                          // lanes whose rows are still to do
    {
        const int row = row_base + SP_WAVES * lane + wave;
        const bool in_range = row < rows;
        // segment and image row without an integer division per lane: rows < 2^22 are exact in fp32, the quotient is off by one at most
        const int H = t.H;
        int n = (int)((float)row * (1.f / (float)H)), r = row - n * H;
        if (r < 0) { --n; r += H; }
        if (r >= H) { ++n; r -= H; }
        if (!in_range) n = 0;
#pragma unroll
        for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)      // (the strides of this path are powers of two)
            my_b[k] = (in_range && k < t.n_strides && (r & (t.stride[k] - 1)) == 0) ? t.seg_off[k][n] + t.row_counts[k][row] : -1;
        bool todo = in_range;
        if (todo && t.stride[0] == 1) {      // lattice 0 holds every mask pixel: its row count says whether the row is empty
            const SP_GLOBAL int32_t* rc = t.row_counts[0];
            const int next = (r + 1 < H) ? rc[row + 1] : t.counts[0][n];
            todo = next != rc[row];
        }
        my_row = in_range ? ((uint32_t)row | ((uint32_t)r << 22)) : 0u;
        rem = __ballot(todo);
    }
    if (rem == 0ull) return;
    // (record fields in registers before the first store: the compiler cannot know that the stores leave the record alone)
    SP_GLOBAL uint32_t* pix_k[SP_PREP_MAX_STRIDES];
    SP_GLOBAL float* baseL_k[SP_PREP_MAX_STRIDES];
    uint32_t lmask[SP_PREP_MAX_STRIDES];            // pixels of an octet on lattice k, bit i = pixel i (strides 8, 16: pixel 0, and only
    bool even_only[SP_PREP_MAX_STRIDES];            //  of even octets for 16)
    int step_k[SP_PREP_MAX_STRIDES];
    const int n_strides = t.n_strides;
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) {
        pix_k[k] = t.pix[k]; baseL_k[k] = t.baseL[k];
        const int st = t.stride[k];
        lmask[k] = k < n_strides ? (st == 1 ? 0xffu : (st == 2 ? 0x55u : (st == 4 ? 0x11u : 0x01u))) : 0u;
        even_only[k] = st == 16;
        step_k[k] = st >= 8 ? 8 : st;
    }
    const SP_GLOBAL uint32_t* const bits_p = t.bits;
    const SP_GLOBAL float* const logdepth_p = t.logdepth;
    // Round 6: baseL == NULL -- the table carries NO copy of the log-depths (the sampling pass reads them from the keyframe's dense array
    // at the point's own (segment, row, column): SP_PREP_DENSE_L): this pass then neither loads the octets' log-depths -- its longest
    // dependent chain -- nor writes 4 bytes per lattice point, and the sampling pass reads 4 bytes per point either way
    const bool with_L = baseL_k[0] != nullptr;
    const int W = t.W, qpr = W >> 4;
    const int rb = min(RB, LIST / (2 * qpr));

    struct Batch {               // up to RB rows: the lanes that hold them (the lowest set bits of `lanes`) and their bit words (a lane per word)
        int n_rows;
        unsigned long long lanes;
        uint32_t bw[RB];
    };
    auto request = [&](Batch& b) {      // the next rb rows still to do: their bit words requested
        b.n_rows = 0;
        b.lanes = rem;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            b.bw[i] = 0u;
            if (i < rb && rem != 0ull) {
                const int src = __builtin_ctzll(rem);
                rem &= rem - 1ull;
                b.n_rows = i + 1;
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, src) & 0x3fffffu;
                if (lane < qpr) b.bw[i] = bits_p[row * (uint32_t)qpr + (uint32_t)lane];
            }
        }
    };
    struct Octet {               // one listed octet of a chunk, its bit word and log-depths requested
        int src, oct, r;
        bool valid, first;
        uint32_t word;
        float4 La, Lb;
    };
    int n_w = 0;
    auto fetch = [&](int g0, Octet& o) {
        const int g = g0 + lane;
        o.valid = g < n_w;
        const int c = o.valid ? s_list[wave][g] : s_list[wave][0];
        o.oct = c & 127; o.src = c >> 7;
        o.first = o.valid && (g == 0 || (s_list[wave][g - 1] >> 7) != o.src);            // first octet of its row
        const uint32_t rr = (uint32_t)__shfl((int)my_row, o.src, 64);
        o.r = (int)(rr >> 22);
        // (the bit word again, from the L2 this time -- the wave just read it -- next to the log-depths: no LDS copy of the batch's words)
        o.word = bits_p[(rr & 0x3fffffu) * (uint32_t)qpr + (uint32_t)(o.oct >> 1)];
        const SP_GLOBAL float* Lp = logdepth_p + ((rr & 0x3fffffu) * (uint32_t)W + (uint32_t)(8 * o.oct));
        o.La = o.Lb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (with_L) {                                        // (wave-uniform)
            o.La = load4((const SP_GLOBAL f32x4*)Lp);        // (both halves, whatever the bits say, and for the lanes past the list the first
            o.Lb = load4((const SP_GLOBAL f32x4*)(Lp + 4));  //  octet of the list: no control flow around the loads)
        }
    };
    Batch bt, bt_next;
    request(bt);
    while (bt.n_rows > 0) {
        // (a) the ordered list of its non-empty octets: row by row from the registers, a lane per bit word = two octets
        n_w = 0;
        unsigned long long walk = bt.lanes;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i < bt.n_rows) {
                const int src = __builtin_ctzll(walk);
                walk &= walk - 1ull;
                const bool lo = (bt.bw[i] & 0x03030303u) != 0u, hi = (bt.bw[i] & 0x0c0c0c0cu) != 0u;
                const unsigned long long bal_lo = __ballot(lo), bal_hi = __ballot(hi);
                const int at = n_w + __popcll(bal_lo & below) + __popcll(bal_hi & below);
                const int head = src << 7;
                if (lo) s_list[wave][at] = (uint16_t)(head | (2 * lane));
                if (hi) s_list[wave][at + (lo ? 1 : 0)] = (uint16_t)(head | (2 * lane + 1));
                n_w += __popcll(bal_lo) + __popcll(bal_hi);
            }
        }
        request(bt_next);                                   // (the next batch's bit words are on their way during this one's chunks)
        // (b) 64 listed octets at a time
        int run[SP_PREP_MAX_STRIDES] = {0, 0, 0, 0};        // points of the batch before this chunk, per lattice
        Octet cur, nxt;
        if (n_w > 0) fetch(0, cur);
        for (int g0 = 0; g0 < n_w; g0 += 64) {
            if (g0 + 64 < n_w) fetch(g0 + 64, nxt);
            int bk[SP_PREP_MAX_STRIDES];
            uint32_t sel[SP_PREP_MAX_STRIDES];
            // pixel i of the octet sits at bit 8 i of e (i < 4) or 8 (i - 4) + 1: bring it to bit i
            const uint32_t e = (cur.word >> (2 * (cur.oct & 1))) & 0x03030303u;
            const uint32_t m8 = (((e & 0x01010101u) * 0x10204080u) >> 28) | ((((e >> 1) & 0x01010101u) * 0x10204080u) >> 24 & 0xf0u);
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) {
                bk[k] = -1;
                if (k < n_strides) bk[k] = __shfl(my_b[k], cur.src, 64);
                sel[k] = (cur.valid && bk[k] >= 0 && !(even_only[k] && (cur.oct & 1))) ? (m8 & lmask[k]) : 0u;
            }
            // wave-wide prefix sums of the four counts (two words of 16-bit fields: a chunk holds at most 512 points)
            const uint32_t p0 = (uint32_t)__popc(sel[0]) | ((uint32_t)__popc(sel[1]) << 16), p1 = (uint32_t)__popc(sel[2]) | ((uint32_t)__popc(sel[3]) << 16);
            const uint32_t i0 = wave_inclusive_scan(p0), i1 = wave_inclusive_scan(p1);
            const uint32_t e0 = i0 - p0, e1 = i1 - p1;
            const int E[SP_PREP_MAX_STRIDES] = {(int)(e0 & 0xffffu), (int)(e0 >> 16), (int)(e1 & 0xffffu), (int)(e1 >> 16)};
            if (cur.first) {
#pragma unroll
                for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) s_delta[wave][k][cur.src] = bk[k] - (run[k] + E[k]);
            }
            const float Lq[8] = {cur.La.x, cur.La.y, cur.La.z, cur.La.w, cur.Lb.x, cur.Lb.y, cur.Lb.z, cur.Lb.w};
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
            const int T[SP_PREP_MAX_STRIDES] = {(int)(t0 & 0xffffu), (int)(t0 >> 16), (int)(t1 & 0xffffu), (int)(t1 >> 16)};
            // The points of the chunk go to the tables THROUGH LDS, in list order: a lane's points sit 4 to 32 bytes from its neighbour's
            // in the table, so storing them lane by lane (one store per pixel of the octet) made 16 partial write requests of every
            // store instruction -- 58 M requests to the L2 for 128 keyframes, half its request rate, with the waves stalled on the
            // issue of the next store; staged, consecutive lanes store consecutive points (two full lines per instruction).  The
            // staged word carries the lane that holds the row in the free bits 10..15 (rows of at most 1024 pixels on this path).
            const uint32_t pw0 = ((uint32_t)cur.r << 16) | ((uint32_t)cur.src << 10) | (uint32_t)(8 * cur.oct);
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) {
                if (k >= n_strides) break;
                if (T[k] == 0) continue;
                // (no control flow around the staging stores: an unset pixel goes to the lane's own spare slot; only the pixels the
                //  lattice can hold are looked at)
                if (step_k[k] == 1) stage_octet<1>(&s_stage[wave][0], sel[k], E[k], pw0, Lq, lane);
                else if (step_k[k] == 2) stage_octet<2>(&s_stage[wave][0], sel[k], E[k], pw0, Lq, lane);
                else if (step_k[k] == 4) stage_octet<4>(&s_stage[wave][0], sel[k], E[k], pw0, Lq, lane);
                else stage_octet<8>(&s_stage[wave][0], sel[k], E[k], pw0, Lq, lane);
                for (int j = lane; j < T[k]; j += 64) {
                    const uint2 v = s_stage[wave][j];
                    const int dest = s_delta[wave][k][(v.x >> 10) & 0x3fu] + run[k] + j;
                    pix_k[k][(uint32_t)dest] = v.x & 0xffff03ffu;            // (non-temporal stores and log-depth loads here: 215 -> 270 us)
                    if (with_L) baseL_k[k][(uint32_t)dest] = __uint_as_float(v.y);
                }
            }
            run[0] += T[0]; run[1] += T[1]; run[2] += T[2]; run[3] += T[3];
            cur = nxt;
        }
        bt = bt_next;
    }
}

// Ordered compaction of (segment,row) rows into every lattice's table for the keyframes OFF the bit-word path (any width, alignment
// and stride; k_prep_fill_bits takes the others).  A workgroup takes SP_FILL_ROWS consecutive rows at a time; one thread per row decides
// from the row counts alone whether its row is empty (a segment covers a small part of the image: ~85 % of its mask rows are empty and
// are not read again) and, if not, fetches where the row's points start in every lattice's table: those loads are in flight together,
// once per round, and the row loop takes row and starts from LDS.  The four waves share the non-empty rows.  Word path: a lane owns 4
// consecutive pixels; its rank inside the row is the prefix sum over lower lanes of their set-pixel counts (0..4), taken from three
// ballots of the count's bits.
__global__ __launch_bounds__(SP_BLOCK) void k_prep_fill(const SpPrepTable* __restrict__ tables) {
    const PrepTable& t = table_of(tables, blockIdx.y);
    if (prep_fill_bits_path(t)) return;       // (k_prep_fill_bits takes those keyframes)
    const int rows = t.N * t.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int s_rows[SP_FILL_ROWS];                             // row id << 16 | row of the image
    __shared__ int s_r[SP_FILL_ROWS];
    __shared__ int s_base[SP_PREP_MAX_STRIDES][SP_FILL_ROWS];       // first table position of the row per lattice, -1: not on the lattice
    __shared__ int s_cnt[SP_WAVES];
    const unsigned long long below = (1ull << lane) - 1ull;
    // (a bounded grid per keyframe whose workgroups walk the rows: the launch costs nothing when every keyframe is on the bits path)
    for (int row_base = blockIdx.x * SP_FILL_ROWS; row_base < rows; row_base += gridDim.x * SP_FILL_ROWS) {
    __syncthreads();                                                 // (the previous round is done with the lists)
    {   // one thread per row: is it empty?  then a block-wide ordered compaction of the non-empty ones.  The row's start in every
        // lattice's table is requested together with the counts that say whether it is empty (not after: one round trip to
        // memory per workgroup less, for 24 bytes per row more)
        const int row = row_base + (int)threadIdx.x;
        const bool in_range = row < rows && (int)threadIdx.x < SP_FILL_ROWS;
        const int n = in_range ? row / t.H : 0, r = row - n * t.H;
        int b[SP_PREP_MAX_STRIDES];
#pragma unroll
        for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
            b[k] = (in_range && k < t.n_strides && r % t.stride[k] == 0) ? t.seg_off[k][n] + t.row_counts[k][row] : -1;
        bool todo = in_range;
        if (todo && t.stride[0] == 1) {      // lattice 0 holds every mask pixel: its row count says whether the row is empty
            const SP_GLOBAL int32_t* rc = t.row_counts[0];
            const int next = (r + 1 < t.H) ? rc[row + 1] : t.counts[0][n];
            todo = next != rc[row];
        }
        const unsigned long long bal = __ballot(todo);
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_cnt[w];
        if (todo) {
            const int slot = off + __popcll(bal & below);
            s_rows[slot] = row;
            s_r[slot] = r;
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) s_base[k][slot] = b[k];
        }
        __syncthreads();
    }
    const int s_n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (s_n == 0) continue;
    const bool words = (t.W & 3) == 0 && ((uintptr_t)t.masks & 3) == 0;
    // The fields of the record the row loop uses, read ONCE into registers: the compiler cannot know that the stores through
    // pix / baseL leave the record alone and read a field again after every one of them -- 200 scalar loads in the kernel, each
    // a wait inside the row loop, which is where the waves were parked.
    SP_GLOBAL uint32_t* pix_k[SP_PREP_MAX_STRIDES];
    SP_GLOBAL float* baseL_k[SP_PREP_MAX_STRIDES];
    int stride_k[SP_PREP_MAX_STRIDES];
#pragma unroll
    for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) { pix_k[k] = t.pix[k]; baseL_k[k] = t.baseL[k]; stride_k[k] = t.stride[k]; }
    const SP_GLOBAL float* const logdepth_p = t.logdepth;
    const SP_GLOBAL uint8_t* const masks_p = t.masks;
    const int W = t.W;
    // One row's compaction into every lattice's table.  A lane owns the 4-pixel groups xw = q * 64 + lane (q < 4: rows of up to
    // 1024 pixels); nz[q] = nonzero_bytes() form of its mask bits, Lv[q] = the 4 log-depths of the group.
    int base[SP_PREP_MAX_STRIDES];
    int r = 0;
    auto row_begin = [&](int slot) {
        r = __builtin_amdgcn_readfirstlane(s_r[slot]);
#pragma unroll
        for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) base[k] = __builtin_amdgcn_readfirstlane(s_base[k][slot]);
    };
    // the groups w0 + q * 64 + lane (q < 4) of the row begun with row_begin(), left to right
    auto emit_groups = [&](int w0, const uint32_t (&nz)[4], const float4 (&Lv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (__ballot(nz[q] != 0u) == 0ull) continue;
            const int xw = w0 + q * 64 + lane;
            const float Lq[4] = {Lv[q].x, Lv[q].y, Lv[q].z, Lv[q].w};
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k) {
                if (base[k] < 0) continue;
                const uint32_t sel = nz[q] & lattice_bytes(4 * xw, stride_k[k]);
                const int cnt = __popc(sel);
                const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
                int pos = base[k] + __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((sel >> (8 * j + 7)) & 1u) {
                        pix_k[k][pos] = ((uint32_t)r << 16) | (uint32_t)(4 * xw + j);
                        if (baseL_k[k]) baseL_k[k][pos] = Lq[j];
                        ++pos;
                    }
                base[k] += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
            }
        }
    };
    for (int i = wave; i < s_n; i += SP_WAVES) {
        const int row_id = s_rows[i];
        if (!words) {
#pragma unroll
            for (int k = 0; k < SP_PREP_MAX_STRIDES; ++k)
                if (k < t.n_strides)
                    fill_row(generic(t.masks), generic(t.logdepth), row_id, t.H, W, stride_k[k], generic(t.seg_off[k]), generic(t.row_counts[k]),
                             generic(pix_k[k]), generic(baseL_k[k]));
            continue;
        }
        const SP_GLOBAL uint32_t* mw = (const SP_GLOBAL uint32_t*)(masks_p + (size_t)row_id * W);
        const SP_GLOBAL float* L = logdepth_p + (size_t)row_id * W;
        row_begin(i);
        for (int w0 = 0; w0 < (W >> 2); w0 += 256) {       // (4 x 64 groups at a time)
            uint32_t nz[4];
            float4 Lv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xw = w0 + q * 64 + lane;
                nz[q] = xw < (W >> 2) ? nonzero_bytes(mw[xw]) : 0u;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ((nz[q] >> (8 * j + 7)) & 1u) ? L[4 * xw + j] : 0.f;
                Lv[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
            emit_groups(w0, nz, Lv);
        }
    }
    }
}

__global__ void k_prep_keypoint_L(const SpPrepTable* __restrict__ tables) {
    const PrepTable& t = table_of(tables, blockIdx.y);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < t.N && t.kp_L) keypoint_L(generic(t.logdepth), generic(t.keypoints), n, t.H, t.W, generic(t.kp_L));
}

// every pyramid level of one table in one pass: segment search, depth and validity once per point.  Padding positions of a
// segment's run are written as {pix 0, src4 0} = invalid points (the arrays need no prior clearing).  A workgroup takes
// SP_SAMPLE_BLOCKS consecutive 256-point blocks: the scalar segment search (a chain of dependent loads) is repeated only when a
// block leaves the current segment's padded run.
template <int SP_SAMPLE_BLOCKS>
__global__ __launch_bounds__(SP_BLOCK) void k_prep_sample(const SpPrepSample* __restrict__ jobs, int blocks_per_job, int total_blocks,
                                                          const SpPrepSample* __restrict__ jobs_b, int blocks_per_job_b) {
    // One-dimensional grid over (table, block) pairs, rounded up to a multiple of 8.  Workgroups are dealt to the 8 XCDs round robin
    // in dispatch order: taken as they come, the blocks of one table -- which gather from the same source image -- run on all
    // eight and every XCD's L2 fetches that image.  Chunked order: every XCD gets a contiguous range of the pairs.
    const int v = xcd_chunked_tile(blockIdx.x, total_blocks);
    if (v >= total_blocks) return;
    // TWO TABLES OF A KEYFRAME IN ONE LAUNCH (sp_prepare_sample_pairs, jobs_b != NULL): record i of `jobs` and record i of `jobs_b` belong to the
    // same keyframe and take consecutive workgroups -- both tables gather from the same source image (the all-points table and the stride-2
    // lattice are both sampled at level 0), which a launch per table fetches from memory twice: 3.7 MB of the sampling pass's ~24 MB per pair
    // ... INTERLEAVED: the tables list their segments in the same order and in the same proportions, so workgroup g of the smaller one and
    // workgroups r g .. r g + r - 1 of the larger (r = the ratio of their sizes, rounded up) gather from the same region of the image: they take
    // consecutive workgroups, close in time on one XCD (the image of a pair is as large as an XCD's L2: a table after the other finds it gone)
    int job, bx;
    bool second = false;
    if (blocks_per_job_b > 0) {
        const int r = (blocks_per_job + blocks_per_job_b - 1) / blocks_per_job_b, per = blocks_per_job_b * (r + 1);
        job = v / per;
        const int rem = v - job * per, grp = rem / (r + 1), within = rem - grp * (r + 1);
        second = within == r;
        bx = second ? grp : grp * r + within;
        if (!second && bx >= blocks_per_job) return;
    } else {
        job = v / blocks_per_job;
        bx = v - job * blocks_per_job;
    }
    const PrepSample& j = reinterpret_cast<const PrepSample*>(second ? jobs_b : jobs)[job];
    // padding granule of the table: 256 (a whole 256-point block lies in one segment) or 64 (wave spans: every WAVE's 64 points
    // do; the search then runs per wave)
    const int lane_off = (j.granule & 0xffff) == 64 ? (int)(threadIdx.x & ~63u) : 0;
    const bool depth_table = (j.granule & SP_PREP_DEPTH_TABLE) != 0;      // src4.w = exp(L) for the cost kernels' depth-table form
    const bool dense_L = (j.granule & SP_PREP_DENSE_L) != 0;              // baseL = the keyframe's dense (N,H,W) log-depths, read at (segment, row, column)
    // The pass is bound by memory latency, not bytes or arithmetic (waves parked on s_waitcnt 87 % of their cycles when every
    // point went load -> geometry -> 12 taps -> store on its own): the SP_SAMPLE_BLOCKS points of a thread go through each stage
    // together -- all table words requested, then all taps of a level, then the stores.
    int idx[SP_SAMPLE_BLOCKS], seg[SP_SAMPLE_BLOCKS];
    bool in_table[SP_SAMPLE_BLOCKS], live[SP_SAMPLE_BLOCKS];
    float shift[SP_SAMPLE_BLOCKS];
    {
        int n = 0, first = 0, count = 0, next_first = -1;
        float sh = 0.f;
#pragma unroll
        for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k) {
            const int b0 = (bx * SP_SAMPLE_BLOCKS + k) * SP_BLOCK;
            const int i0 = __builtin_amdgcn_readfirstlane(b0 + lane_off);        // first point of this wave's / block's unit
            idx[k] = b0 + (int)threadIdx.x;
            if (i0 < j.P && (k == 0 || i0 >= next_first)) {          // (per wave when the granule is 64: every wave tracks its own run)
                n = segment_of(generic(j.seg_off), j.N, i0);
                first = j.seg_off[n];
                count = j.counts[n];
                sh = j.kld[n] - j.kp_L[n];
                next_first = n + 1 < j.N ? j.seg_off[n + 1] : j.P;
                if (next_first <= first) next_first = j.P;
            }
            in_table[k] = idx[k] < j.P;
            live[k] = in_table[k] && idx[k] - first < count;       // (else padding of the segment's run: an invalid point)
            shift[k] = sh;
            seg[k] = n;
        }
    }
    SP_GLOBAL uint32_t* const pix = j.pix;             // (record fields in registers before the first store: see k_prep_fill)
    const int n_levels = j.n_levels;
    uint32_t pw[SP_SAMPLE_BLOCKS];
    float L[SP_SAMPLE_BLOCKS];
#pragma unroll
    for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k) {
        pw[k] = live[k] ? pix[idx[k]] : 0u;
        if (!dense_L) L[k] = live[k] ? j.baseL[idx[k]] : 0.f;
    }
    if (dense_L) {
        // (consecutive table points are consecutive columns of one mask row: the lanes of a wave read runs of consecutive floats)
#pragma unroll
        for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k) {
            const uint32_t at = ((uint32_t)seg[k] * (uint32_t)j.H + ((pw[k] >> 16) & 0x7fffu)) * (uint32_t)j.W + (pw[k] & 0xffffu);
            L[k] = live[k] ? j.baseL[at] : 0.f;
        }
    }
    SourceGeom g[SP_SAMPLE_BLOCKS];
#pragma unroll
    for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k) g[k] = source_geometry(pw[k], L[k], shift[k], j.H, j.W, generic(j.K));
    for (int l = 0; l < n_levels; ++l) {
        const SP_GLOBAL float* const image = j.image[l];
        SP_GLOBAL f32x4* const out = (SP_GLOBAL f32x4*)j.src4[l];
        const int Hl = j.Hl[l], Wl = j.Wl[l];
        float4 v[SP_SAMPLE_BLOCKS];
#pragma unroll
        for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k) {
            v[k] = source_sample(g[k], generic(image), Hl, Wl);
            if (depth_table) v[k].w = fast_exp(v[k].w);
        }
#pragma unroll
        for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k)
            if (in_table[k]) store4(out + idx[k], live[k] ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
#pragma unroll
    for (int k = 0; k < SP_SAMPLE_BLOCKS; ++k)
        if (in_table[k]) pix[idx[k]] = live[k] ? (g[k].pw | (g[k].ok ? 0x80000000u : 0u)) : 0u;
}

// One pyramid step of a batch of planar images.  Rows of W = 4 k pixels (16-byte aligned): a thread makes TWO neighbouring outputs
// from one 16-byte load and one 4-byte load (the column left of it) of each of the three input rows -- 6 loads per 2 outputs, the
// wide ones contiguous over the wave; one output per thread is 9 four-byte loads 8 bytes apart, and the pass sat in the load
// issue (issue stalls 43 % of the wave cycles, 0.43 of the HBM roofline).  Same products and summation order as
// blur_decimate_at().  The grid covers max_out_pixels / 2 threads per plane: other shapes take two outputs per thread, one at a time.
__global__ __launch_bounds__(SP_BLOCK) void k_prep_blur(const SpPrepImage* __restrict__ jobs) {
    const PrepImage& j = reinterpret_cast<const PrepImage*>(jobs)[blockIdx.z];
    const int Ho = (j.H + 1) / 2, Wo = (j.W + 1) / 2;
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    const SP_GLOBAL float* in = j.in + (size_t)blockIdx.y * j.H * j.W;
    SP_GLOBAL float* out = j.out + (size_t)blockIdx.y * Ho * Wo;
    if ((j.W & 3) == 0 && ((uintptr_t)j.in & 15) == 0 && ((uintptr_t)j.out & 7) == 0) {
        const int half = Wo >> 1;
        if (i >= Ho * half) return;
        const int yo = i / half, t = i - yo * half;
        const float wgt[3] = {1.f, 2.f, 1.f};
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const SP_GLOBAL float* row = in + (size_t)reflect1(2 * yo + dy - 1, j.H) * j.W;
            const float4 v = load4((const SP_GLOBAL f32x4*)(row + 4 * t));
            const float left = row[max(4 * t - 1, 0)];
            const float c0 = t > 0 ? left : v.y;            // column -1 reflects onto column 1
            const float w0 = wgt[dy] * wgt[0] * (1.f / 16.f), w1 = wgt[dy] * wgt[1] * (1.f / 16.f), w2 = wgt[dy] * wgt[2] * (1.f / 16.f);
            a0 = fmaf(w0, c0, a0);  a0 = fmaf(w1, v.x, a0);  a0 = fmaf(w2, v.y, a0);
            a1 = fmaf(w0, v.y, a1); a1 = fmaf(w1, v.z, a1);  a1 = fmaf(w2, v.w, a1);
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 o = {a0, a1};
        *(SP_GLOBAL f32x2*)(out + (size_t)yo * Wo + 2 * t) = o;
        return;
    }
    const int n_threads = gridDim.x * SP_BLOCK;
    for (int o = i; o < Ho * Wo; o += n_threads) out[o] = blur_decimate_at(generic(in), j.H, j.W, Wo, o);
}

// One pyramid step of a batch of THREE-CHANNEL images AND their packed (HWC3) forms in the same pass (round 6): the target frames of a
// batch are blurred level by level and every level is also packed for the cost kernel -- as two passes the packing re-reads what the pyramid
// step has just had in registers (12 B per pixel and level of the set-up's traffic, one launch).  A thread makes two neighbouring outputs of
// all three channels exactly like k_prep_blur (same loads, products and summation order) and, from the same registers, writes
//   out        planar level l + 1 (may be NULL: nobody blurs the last level further)
//   packed_out level l + 1 packed (or NULL)
//   packed_in  level l packed (or NULL): rows 2 yo, 2 yo + 1, columns 4 t .. 4 t + 3 -- the 2 x 4 pixels of its 3 x 5 window that no other thread owns
// Pure copies and the same fused multiply-adds: bit-identical to sp_prepare_blur followed by sp_prepare_pack.
struct PrepImagePack {
    const SP_GLOBAL float* in;
    SP_GLOBAL float* out;
    SP_GLOBAL float* packed_in;
    SP_GLOBAL float* packed_out;
    int32_t H, W;
};
static_assert(sizeof(PrepImagePack) == sizeof(SpPrepImagePack), "device view of the job record");
__global__ __launch_bounds__(SP_BLOCK) void k_prep_blur_pack(const SpPrepImagePack* __restrict__ jobs) {
    const PrepImagePack& j = reinterpret_cast<const PrepImagePack*>(jobs)[blockIdx.z];
    const int H = j.H, W = j.W, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t HW = (size_t)H * W, HWo = (size_t)Ho * Wo;
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if ((W & 3) == 0 && ((uintptr_t)j.in & 15) == 0 && ((uintptr_t)j.out & 7) == 0 && ((uintptr_t)j.packed_in & 15) == 0 && ((uintptr_t)j.packed_out & 7) == 0) {
        const int half = Wo >> 1;
        if (i >= Ho * half) return;
        const int yo = i / half, t = i - yo * half;
        const float wgt[3] = {1.f, 2.f, 1.f};
        float a0[3] = {0.f, 0.f, 0.f}, a1[3] = {0.f, 0.f, 0.f};
        float4 own[2][3];                                   // rows 2 yo and 2 yo + 1 of the window, per channel
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const SP_GLOBAL float* in = j.in + c * HW;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const SP_GLOBAL float* row = in + (size_t)reflect1(2 * yo + dy - 1, H) * W;
                const float4 v = load4((const SP_GLOBAL f32x4*)(row + 4 * t));
                const float left = row[max(4 * t - 1, 0)];
                const float c0 = t > 0 ? left : v.y;            // column -1 reflects onto column 1
                const float w0 = wgt[dy] * wgt[0] * (1.f / 16.f), w1 = wgt[dy] * wgt[1] * (1.f / 16.f), w2 = wgt[dy] * wgt[2] * (1.f / 16.f);
                a0[c] = fmaf(w0, c0, a0[c]);  a0[c] = fmaf(w1, v.x, a0[c]);  a0[c] = fmaf(w2, v.y, a0[c]);
                a1[c] = fmaf(w0, v.y, a1[c]); a1[c] = fmaf(w1, v.z, a1[c]);  a1[c] = fmaf(w2, v.w, a1[c]);
                if (dy > 0) own[dy - 1][c] = v;
            }
        }
        if (j.out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x2 o = {a0[c], a1[c]};
                *(SP_GLOBAL f32x2*)(j.out + c * HWo + (size_t)yo * Wo + 2 * t) = o;
            }
        }
        if (j.packed_out) {                                 // two texels = 24 contiguous bytes
            SP_GLOBAL f32x2* q = (SP_GLOBAL f32x2*)(j.packed_out + ((size_t)yo * Wo + 2 * t) * SP_TEXEL_FLOATS);
            const f32x2 q0 = {a0[0], a0[1]}, q1 = {a0[2], a1[0]}, q2 = {a1[1], a1[2]};
            q[0] = q0; q[1] = q1; q[2] = q2;
        }
        if (j.packed_in) {                                  // four texels of each owned row = 48 contiguous bytes
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = 2 * yo + r;
                if (y >= H) break;                          // (odd H: the window's last row is a reflection, nobody's own)
                const float4 R = own[r][0], G = own[r][1], B = own[r][2];
                SP_GLOBAL f32x4* q = (SP_GLOBAL f32x4*)(j.packed_in + ((size_t)y * W + 4 * t) * SP_TEXEL_FLOATS);
                store4(q, make_float4(R.x, G.x, B.x, R.y));
                store4(q + 1, make_float4(G.y, B.y, R.z, G.z));
                store4(q + 2, make_float4(B.z, R.w, G.w, B.w));
            }
        }
        return;
    }
    // other shapes: an output / a texel per thread and trip
    const int n_threads = gridDim.x * SP_BLOCK;
    for (int o = i; o < Ho * Wo; o += n_threads) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = blur_decimate_at(generic(j.in) + c * HW, H, W, Wo, o);
            if (j.out) j.out[c * HWo + o] = v;
            if (j.packed_out) j.packed_out[(size_t)o * SP_TEXEL_FLOATS + c] = v;
        }
    }
    if (j.packed_in)
        for (int o = i; o < H * W; o += n_threads) pack_texel(generic(j.in), H * W, o, generic(j.packed_in));
}

// many small float vectors (one pointer each) -> one flat array: out[off[i] .. off[i + 1]) = src[i][0 .. off[i + 1] - off[i])
__global__ __launch_bounds__(64) void k_prep_gather(const float* const* __restrict__ src, const long long* __restrict__ off, float* __restrict__ out) {
    const float* p = src[blockIdx.x];
    const long long o0 = off[blockIdx.x], n = off[blockIdx.x + 1] - o0;
    for (long long i = threadIdx.x; i < n; i += 64) out[o0 + i] = p[i];
}

__global__ __launch_bounds__(SP_BLOCK) void k_prep_pack(const SpPrepImage* __restrict__ jobs) {
    const PrepImage& j = reinterpret_cast<const PrepImage*>(jobs)[blockIdx.y];
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i < j.H * j.W) pack_texel(generic(j.in), j.H * j.W, i, generic(j.out));
}

}  // namespace

extern "C" {

int sp_mask_count(const uint8_t* masks, int N, int H, int W, int32_t* row_counts, int32_t* counts, int32_t* seg_off,
                  void* stream) {
    if (!masks || !row_counts || !counts || !seg_off || N <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    if (H > 32767 || W > 65535) return SP_ELIMIT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = N * H;
    hipLaunchKernelGGL(k_row_counts, dim3((rows + SP_WAVES - 1) / SP_WAVES), dim3(SP_BLOCK), 0, s, masks, rows, H, W, row_counts);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_segment_row_scan, dim3(N), dim3(SP_BLOCK), 0, s, row_counts, H, counts);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_segment_offsets, dim3(1), dim3(SP_BLOCK), 0, s, counts, N, seg_off);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_table_fill(const uint8_t* masks, const float* logdepth, const float* keypoints, int N, int H, int W,
                  const int32_t* seg_off, const int32_t* row_off, uint32_t* pix, float* baseL, float* kp_L,
                  void* stream) {
    if (!masks || !logdepth || !keypoints || !seg_off || !row_off || !pix || !baseL || !kp_L) return SP_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = N * H;
    hipLaunchKernelGGL(k_table_fill, dim3((rows + SP_WAVES - 1) / SP_WAVES), dim3(SP_BLOCK), 0, s, masks, logdepth, N, H, W,
                       seg_off, row_off, pix, baseL);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_keypoint_L, dim3((N + 63) / 64), dim3(64), 0, s, logdepth, keypoints, N, H, W, kp_L);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_table_sample_source(uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                           const float* kld, int N, int P, int H, int W, const float* img, int Hl, int Wl,
                           const float* K, float* src4, int set_validity, void* stream) {
    if (!pix || !baseL || !seg_off || !kp_L || !kld || !img || !K || !src4) return SP_EINVAL;
    if (N <= 0 || P <= 0 || H < 2 || W < 2 || Hl < 1 || Wl < 1) return SP_EINVAL;
    hipLaunchKernelGGL(k_sample_source, dim3((P + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), pix, baseL, seg_off, kp_L, kld, N, P, H, W, img, Hl, Wl, K,
                       reinterpret_cast<float4*>(src4), set_validity);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pack_rgb(const float* chw, int B, int H, int W, float* hwc3, void* stream) {
    if (!chw || !hwc3 || B <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_pack_rgb, dim3((H * W + SP_BLOCK - 1) / SP_BLOCK, B), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), chw, H * W, hwc3);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_blur_decimate(const float* in, int C, int H, int W, float* out, void* stream) {
    if (!in || !out || C <= 0 || H < 2 || W < 2) return SP_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(k_blur_decimate, dim3((Ho * Wo + SP_BLOCK - 1) / SP_BLOCK, C), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), in, H, W, Ho, Wo, out);
    SP_CHECK_LAUNCH();
    return 0;
}


static_assert(sizeof(SpPrepTable) == 240 && sizeof(SpPrepSample) == 176 && sizeof(SpPrepImage) == 24, "preparation job records are part of the ABI");

// ---- batched preparation ----
static int check_grid(long x, long y) { return (x <= 0 || y <= 0 || y > 65535) ? SP_EINVAL : 0; }

static int prepare_count(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, bool boxed, void* stream) {
    if (!tables || check_grid(max_rows, n_tables) || max_N <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int per_block = SP_WAVES * SP_PREP_WAVE_ROWS;
    if (boxed) {
        hipLaunchKernelGGL(k_prep_row_counts_boxed, dim3(max_N, n_tables), dim3(SP_BLOCK), 0, s, tables);
    } else hipLaunchKernelGGL(k_prep_row_counts, dim3((max_rows + per_block - 1) / per_block, n_tables), dim3(SP_BLOCK), 0, s, tables);
    SP_CHECK_LAUNCH();
    const int general_rows = SP_WAVES * SP_PREP_ROWS;
    hipLaunchKernelGGL(k_prep_row_counts_general, dim3(std::min((max_rows + general_rows - 1) / general_rows, SP_PREP_GENERAL_BLOCKS), n_tables), dim3(SP_BLOCK), 0, s, tables);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_prep_row_scan, dim3((max_N + SP_WAVES - 1) / SP_WAVES, n_tables, SP_PREP_MAX_STRIDES), dim3(SP_BLOCK), 0, s, tables);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_count(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream) {
    return prepare_count(tables, n_tables, max_rows, max_N, false, stream);
}

int sp_prepare_count_boxed(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream) {
    return prepare_count(tables, n_tables, max_rows, max_N, true, stream);
}

int sp_prepare_fill(const SpPrepTable* tables, int n_tables, int max_rows, int max_N, void* stream) {
    if (!tables || check_grid(max_rows, n_tables) || max_N <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_prep_fill_bits, dim3((max_rows + SP_FILL_ROWS - 1) / SP_FILL_ROWS, n_tables), dim3(SP_BLOCK), 0, s, tables);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_prep_fill, dim3(std::min((max_rows + SP_FILL_ROWS - 1) / SP_FILL_ROWS, SP_PREP_GENERAL_BLOCKS), n_tables), dim3(SP_BLOCK), 0, s, tables);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_prep_keypoint_L, dim3((max_N + 63) / 64, n_tables), dim3(64), 0, s, tables);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_sample(const SpPrepSample* jobs, int n_jobs, int max_P, void* stream) {
    if (!jobs || check_grid(max_P, n_jobs)) return SP_EINVAL;
    constexpr int K = 4;               // points per thread (384 pairs of 640x480x64: 1 -> 1.86 ms, 2 -> 1.46-1.51, 3 -> 1.40, 4 -> 1.40-1.42)
    const long per_job = (max_P + SP_BLOCK * K - 1) / (SP_BLOCK * K), total = per_job * n_jobs;
    if (total + 7 > 0x7fffffffL) return SP_ELIMIT;
    hipLaunchKernelGGL((k_prep_sample<K>), dim3((unsigned)((total + 7) / 8 * 8)), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), jobs, (int)per_job, (int)total,
                       (const SpPrepSample*)nullptr, 0);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_sample_pairs(const SpPrepSample* jobs_a, int max_P_a, const SpPrepSample* jobs_b, int max_P_b, int n_jobs, void* stream) {
    if (!jobs_a || !jobs_b || check_grid(max_P_a, n_jobs) || max_P_b <= 0) return SP_EINVAL;
    constexpr int K = 4;
    const long per_a = (max_P_a + SP_BLOCK * K - 1) / (SP_BLOCK * K), per_b = (max_P_b + SP_BLOCK * K - 1) / (SP_BLOCK * K);
    const long total = per_b * ((per_a + per_b - 1) / per_b + 1) * n_jobs;             // (k_prep_sample's interleaved order: per_b groups of r + 1 workgroups)
    if (total + 7 > 0x7fffffffL) return SP_ELIMIT;
    hipLaunchKernelGGL((k_prep_sample<K>), dim3((unsigned)((total + 7) / 8 * 8)), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), jobs_a, (int)per_a, (int)total,
                       jobs_b, (int)per_b);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_blur(const SpPrepImage* jobs, int n_jobs, int C, int max_out_pixels, void* stream) {
    if (!jobs || check_grid(max_out_pixels, n_jobs) || C <= 0 || C > 65535) return SP_EINVAL;
    const int threads = (max_out_pixels + 1) / 2;          // (k_prep_blur: two outputs per thread)
    hipLaunchKernelGGL(k_prep_blur, dim3((threads + SP_BLOCK - 1) / SP_BLOCK, C, n_jobs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), jobs);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_blur_pack(const SpPrepImagePack* jobs, int n_jobs, int max_out_pixels, void* stream) {
    if (!jobs || check_grid(max_out_pixels, n_jobs)) return SP_EINVAL;
    const int threads = (max_out_pixels + 1) / 2;          // (two outputs of three channels per thread)
    hipLaunchKernelGGL(k_prep_blur_pack, dim3((threads + SP_BLOCK - 1) / SP_BLOCK, 1, n_jobs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), jobs);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_pack(const SpPrepImage* jobs, int n_jobs, int max_pixels, void* stream) {
    if (!jobs || check_grid(max_pixels, n_jobs)) return SP_EINVAL;
    hipLaunchKernelGGL(k_prep_pack, dim3((max_pixels + SP_BLOCK - 1) / SP_BLOCK, n_jobs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), jobs);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_prepare_gather(const float* const* src, const long long* off, int n, float* out, void* stream) {
    if (!src || !off || !out || n <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_prep_gather, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), src, off, out);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
