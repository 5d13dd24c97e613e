// Device-side building blocks shared by the libsp_hip.so kernels (gfx950 / CDNA4 only).
// Numeric conventions are the reference's; each helper cites the lines it restates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sp_hip.h"

#define SP_BLOCK 256            // 4 wavefronts of 64
#define SP_WAVES (SP_BLOCK / 64)

// Explicit global (address space 1) pointers: pointers that reach a kernel through a struct in memory are
// otherwise treated as generic and compile to flat_load (slower path, also occupies the LDS counter).
typedef const uint32_t __attribute__((address_space(1)))* gptr_u32;
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: loadable straight from address space 1
typedef const f32x4 __attribute__((address_space(1)))* gptr_f4;
typedef const float __attribute__((address_space(1)))* gptr_f32;

// streaming (read-once) 16-byte load: non-temporal so the source stream does not evict the target images from L2
__device__ __forceinline__ float4 ntload4(gptr_f4 p) {
    const f32x4 v = __builtin_nontemporal_load(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// exp(x) = exp2(x * log2 e) on the hardware exp2 (v_exp_f32, 1 ulp): 2 instructions instead of ocml expf's ~14.
// The product is split hi/lo so the argument of exp2 carries no extra rounding: relative error <= ~1.5 ulp for
// the log-depth range (|x| < 80), the same class as expf itself.
__device__ __forceinline__ float fast_exp(float x) {
    const float log2e_hi = 1.44269502162933349609375f, log2e_lo = 1.92596299112661746e-8f;
    const float hi = x * log2e_hi;
    const float lo = fmaf(x, log2e_hi, -hi) + x * log2e_lo;     // exact low part of the product + tail of log2 e
    return __builtin_amdgcn_exp2f(hi) * (1.0f + 0.693147180559945f * lo);
}

#define SP_CHECK_LAUNCH()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// sp_pairs_schedule_cost() from a phase every pair is known to have reached (sp_cost.hip; used by sp_pairs_schedule_run in sp_solver.hip)
struct SpSchedule;
struct SpQueue;
// one work list of a launch over MANY (sp_window_gn_run_multi: the cost pass of S windows in one launch); device memory, ordered by first_block
struct MultiList {
    const SpPair* pairs;
    const int32_t* chunks;
    const int32_t* spans;
    float* partials;
    float* seg_partials;
    int32_t n_spans;
    int32_t first_block;      // its workgroups are blockIdx.x - first_block (a multiple of 8)
};
__attribute__((visibility("hidden"))) int cost_pairs_multi(const MultiList* lists_dev, int n_lists, int total_blocks, int mode, float irls_eps, void* stream);
__attribute__((visibility("hidden"))) int schedule_cost_from(const SpSchedule* sched, const int32_t* phase, void* stream, int first_phase, const SpQueue* queue, int n_slots,
                                                              const int32_t* active, int n_active, uint32_t idle_mask);

// ---------------------------------------------------------------------------------------------------
// wave64 / block reductions (fixed order -> bitwise reproducible run to run)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Reduce NV per-thread accumulators over the block; thread k < NV ends up holding the total of value k.
//
// Wave stage = recursive halving ("reduce-scatter" over lanes): at lane distance 32, 16, ... 1 the two halves
// of every lane pair split the surviving values between them, swap the halves they give up and add -- NV/2 +
// NV/4 + ... cross-lane moves in total (41 for NV = 40) instead of 6 per value (240) for a butterfly per value.
// The order of additions is fixed by the lane numbering, so results stay bitwise reproducible.
template <int N, int OFF, class T>
__device__ __forceinline__ void halve_step(T* v, bool upper) {
    constexpr int H = (N + 1) / 2;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const T lo = v[k];
        const T hi = (k + H < N) ? v[k + H] : T(0);
        const T keep = upper ? hi : lo;
        const T give = upper ? lo : hi;
        v[k] = keep + __shfl_xor(give, OFF, 64);
    }
}

// Wave stage: reduce NV per-lane values over the 64 lanes.  Afterwards lane `lane` holds the total of value `pos` in
// acc[0] if `ok` (each of the NV totals lives in exactly one lane).
template <int NV, class T>
__device__ __forceinline__ void wave_sum_to_lanes(T (&acc)[NV], int lane, int& pos, bool& ok) {
    static_assert(NV <= 64, "one value per lane at most");
    constexpr int N0 = NV, N1 = (N0 + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
    halve_step<N0, 32>(acc, lane & 32);
    halve_step<N1, 16>(acc, lane & 16);
    halve_step<N2, 8>(acc, lane & 8);
    halve_step<N3, 4>(acc, lane & 4);
    halve_step<N4, 2>(acc, lane & 2);
    halve_step<N5, 1>(acc, lane & 1);
    // which of the NV values this lane now holds in acc[0] (walk the levels backwards), if any
    pos = 0;
    ok = true;
    if (lane & 1)  { pos += (N5 + 1) / 2; ok = ok && pos < N5; }
    if (lane & 2)  { pos += (N4 + 1) / 2; ok = ok && pos < N4; }
    if (lane & 4)  { pos += (N3 + 1) / 2; ok = ok && pos < N3; }
    if (lane & 8)  { pos += (N2 + 1) / 2; ok = ok && pos < N2; }
    if (lane & 16) { pos += (N1 + 1) / 2; ok = ok && pos < N1; }
    if (lane & 32) { pos += (N0 + 1) / 2; ok = ok && pos < N0; }
}

template <int NV>
__device__ __forceinline__ float block_sum_to_thread(float (&acc)[NV], float* lds /* SP_WAVES*NV floats */, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    int pos;
    bool ok;
    wave_sum_to_lanes<NV>(acc, lane, pos, ok);
    if (ok) lds[wave * NV + pos] = acc[0];
    __syncthreads();
    float total = 0.f;
    if (tid < NV) {
#pragma unroll
        for (int w = 0; w < SP_WAVES; ++w) total += lds[w * NV + tid];
    }
    return total;
}
template <int NV>
__device__ __forceinline__ float block_sum_to_thread(float (&acc)[NV], float* lds) {
    return block_sum_to_thread<NV>(acc, lds, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// XCD-aware tile order: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md, observed, used for
// speed only).  Give every XCD one contiguous chunk of the tile list so that the tiles of one frame pair --
// which gather from the same target image -- share a single XCD's L2 instead of pulling that image into
// all eight.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_chunked_tile(int b, int n) {
    const int per = (n + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    return t;  // caller checks t < n
}

// ---------------------------------------------------------------------------------------------------
// geometry of one table point
// ---------------------------------------------------------------------------------------------------
struct Cam {
    float fx, fy, cx, cy;
};

struct PointGeom {
    float px, py, pz;   // point in the source camera
    float qx, qy, qz;   // point in the target camera
    float zinv;         // guarded 1/qz
    bool  zguard;       // |qz| > 1e-6 (gradient flows through 1/z only then, core/ops.py:33-34)
    float ix, iy;       // sampling position in LEVEL pixels
    bool  valid;        // target validity & source validity
};

// pix word: bit31 source validity, bits 30..16 row, bits 15..0 col
__device__ __forceinline__ void decode_pix(uint32_t w, float& col, float& row, bool& src_ok) {
    col = (float)(w & 0xffffu);
    row = (float)((w >> 16) & 0x7fffu);
    src_ok = (w >> 31) != 0u;
}

// core/dense_optim.py:19-35 with the depth of :38-86:  d = exp(L + (kld[n] - L[n,kp])).
__device__ __forceinline__ void backproject(float col, float row, float d, const Cam& K, float ifx, float ify,
                                            float& x, float& y) {
    x = ((col - K.cx) * d) * ifx;
    y = ((row - K.cy) * d) * ify;
}

struct Warp {
    float R[9], t[3];
    Cam Kt;
    float invWm1, invHm1;   // 1/(W-1), 1/(H-1) of the GEOMETRY grid
    float sx, sy;           // 0.5*(Wl-1), 0.5*(Hl-1)
    float zmin;
};

// core/dense_optim.py:117-122 (rigid), core/ops.py:19-40 (guarded projection), tool/point_utils.py:31-35
// (normalise with geometry dims), core/dense_optim.py:128-130,146,160 (validity), grid_sample's
// align_corners un-normalisation ((x+1)/2*(size-1)).
__device__ __forceinline__ void warp_point(const Warp& w, float px, float py, float pz, PointGeom& g) {
    g.px = px; g.py = py; g.pz = pz;
    g.qx = fmaf(w.R[0], px, fmaf(w.R[1], py, w.R[2] * pz)) + w.t[0];
    g.qy = fmaf(w.R[3], px, fmaf(w.R[4], py, w.R[5] * pz)) + w.t[1];
    g.qz = fmaf(w.R[6], px, fmaf(w.R[7], py, w.R[8] * pz)) + w.t[2];
    g.zguard = fabsf(g.qz) > 1e-6f;
    g.zinv = g.zguard ? __builtin_amdgcn_rcpf(g.qz) : 1e-6f;
    const float u = g.qx * w.Kt.fx * g.zinv + w.Kt.cx;
    const float v = g.qy * w.Kt.fy * g.zinv + w.Kt.cy;
    const float xn = 2.f * u * w.invWm1 - 1.f;
    const float yn = 2.f * v * w.invHm1 - 1.f;
    g.valid = (fabsf(xn) <= 0.99f) && (fabsf(yn) <= 0.99f) && (g.qz > w.zmin);
    g.ix = (xn + 1.f) * w.sx;
    g.iy = (yn + 1.f) * w.sy;
}

#define SP_TEXEL_FLOATS 3      // target images are packed HWC3: r,g,b of one pixel adjacent, 12 bytes per texel

// One bilinear footprint of a packed HWC3 image.  Taps outside the image are zero (grid_sample
// padding_mode='zeros'); for valid points (0.99 band) all four taps are inside whenever Wl,Hl >= 2.
struct Taps {
    float4 t00, t10, t01, t11;
    float wx, wy;
};

__device__ __forceinline__ void fetch_taps(gptr_f32 img, int Wl, int Hl, float ix, float iy, Taps& tp) {
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    tp.wx = ix - fx0;
    tp.wy = iy - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool inx0 = (x0 >= 0) & (x0 < Wl), inx1 = (x1 >= 0) & (x1 < Wl);
    const bool iny0 = (y0 >= 0) & (y0 < Hl), iny1 = (y1 >= 0) & (y1 < Hl);
    const int cx0 = min(max(x0, 0), Wl - 1), cx1 = min(max(x1, 0), Wl - 1);
    const int cy0 = min(max(y0, 0), Hl - 1), cy1 = min(max(y1, 0), Hl - 1);
    struct Rgb { float x, y, z; };
    auto texel = [&](int yy, int xx) { gptr_f32 q = img + (size_t)(yy * Wl + xx) * SP_TEXEL_FLOATS; return Rgb{q[0], q[1], q[2]}; };
    const Rgb a = texel(cy0, cx0), b = texel(cy0, cx1);
    const Rgb c = texel(cy1, cx0), d = texel(cy1, cx1);
    const float m00 = (inx0 & iny0) ? 1.f : 0.f, m10 = (inx1 & iny0) ? 1.f : 0.f;
    const float m01 = (inx0 & iny1) ? 1.f : 0.f, m11 = (inx1 & iny1) ? 1.f : 0.f;
    tp.t00 = make_float4(a.x * m00, a.y * m00, a.z * m00, 0.f);
    tp.t10 = make_float4(b.x * m10, b.y * m10, b.z * m10, 0.f);
    tp.t01 = make_float4(c.x * m01, c.y * m01, c.z * m01, 0.f);
    tp.t11 = make_float4(d.x * m11, d.y * m11, d.z * m11, 0.f);
}

__device__ __forceinline__ float bilerp(float a00, float a10, float a01, float a11, float wx, float wy) {
    const float top = fmaf(wx, a10 - a00, a00);
    const float bot = fmaf(wx, a11 - a01, a01);
    return fmaf(wy, bot - top, top);
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ void load_cam(const float* __restrict__ K9, Cam& c) {
    c.fx = K9[0]; c.cx = K9[2]; c.fy = K9[4]; c.cy = K9[5];
}

__device__ __forceinline__ void load_pose(const float* __restrict__ T16, float (&R)[9], float (&t)[3]) {
    R[0] = T16[0]; R[1] = T16[1]; R[2] = T16[2];  t[0] = T16[3];
    R[3] = T16[4]; R[4] = T16[5]; R[5] = T16[6];  t[1] = T16[7];
    R[6] = T16[8]; R[7] = T16[9]; R[8] = T16[10]; t[2] = T16[11];
}
