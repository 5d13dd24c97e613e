// The fused photometric cost pass: per table point, depth seed -> exp -> unproject -> SE(3) -> project ->
// bilinear sample -> brightness -> masked L1, and in the same pass either the analytic gradient (mode 0) or
// the IRLS-weighted Gauss-Newton normal equations (mode 1).  One workgroup = one tile (a run of points of a
// single segment); per-tile partial sums go to a workspace and are combined in fixed order by a finalise /
// solver kernel, so results are bitwise reproducible.
//
// Replaces core/dense_optim.py:265-403 and core/dense_optim_batch.py:50-147 of the reference (about 150 ATen
// launches forward + the autograd backward) -- see include/sp_hip.h for the ABI and DESIGN.md for the layout.
#include "sp_device.h"

namespace {

// Everything a tile needs; filled from kernel arguments (single source, B targets) or from an SpPair record.
struct TileCtx {
    const uint32_t* __restrict__ pix;
    const float4* __restrict__ src4;
    const float4* __restrict__ trg4;
    Cam Ks;
    Warp w;
    float shift;      // kld[n] - kp_L[n]
    float gain, bias; // exp(-(a_t-a_s)), b_t-b_s
    int Wl, Hl;
    int start, count;
};

// -----------------------------------------------------------------------------------------------------
// mode 0: gradient accumulators
//   [0] sum |r|      [1..3] g_t      [4..12] g_R (row-major)     [13] g_kld(segment of the tile)
//   [14] d/da_t      [15] d/db_t          (all still to be scaled by 1/(3P))
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_grad(const TileCtx& c, const PointGeom& g, const float4 s,
                                                float (&acc)[SP_GRAD_PARTIAL_FLOATS]) {
    Taps tp;
    fetch_taps(c.trg4, c.Wl, c.Hl, g.ix, g.iy, tp);
    const float itr = bilerp(tp.t00.x, tp.t10.x, tp.t01.x, tp.t11.x, tp.wx, tp.wy);
    const float itg = bilerp(tp.t00.y, tp.t10.y, tp.t01.y, tp.t11.y, tp.wx, tp.wy);
    const float itb = bilerp(tp.t00.z, tp.t10.z, tp.t01.z, tp.t11.z, tp.wx, tp.wy);
    const float dr = s.x - fmaf(c.gain, itr, c.bias);
    const float dg = s.y - fmaf(c.gain, itg, c.bias);
    const float db = s.z - fmaf(c.gain, itb, c.bias);
    acc[0] += fabsf(dr) + fabsf(dg) + fabsf(db);
    const float sr = sgn(dr), sg = sgn(dg), sb = sgn(db);
    // sign-weighted channel mix of each tap, then the two bilinear slopes of the mix
    const float m00 = sr * tp.t00.x + sg * tp.t00.y + sb * tp.t00.z;
    const float m10 = sr * tp.t10.x + sg * tp.t10.y + sb * tp.t10.z;
    const float m01 = sr * tp.t01.x + sg * tp.t01.y + sb * tp.t01.z;
    const float m11 = sr * tp.t11.x + sg * tp.t11.y + sb * tp.t11.z;
    const float mx = fmaf(tp.wy, (m11 - m01) - (m10 - m00), m10 - m00);
    const float my = fmaf(tp.wx, (m11 - m10) - (m01 - m00), m01 - m00);
    // d|r|/d(u,v): -gain * slope * (level px per geometry px)
    const float gu = -c.gain * mx * (2.f * c.w.sx * c.w.invWm1);
    const float gv = -c.gain * my * (2.f * c.w.sy * c.w.invHm1);
    const float a = gu * c.w.Kt.fx * g.zinv, b = gv * c.w.Kt.fy * g.zinv;
    const float gqx = a, gqy = b;
    const float gqz = g.zguard ? -(a * g.qx + b * g.qy) * g.zinv : 0.f;
    acc[1] += gqx; acc[2] += gqy; acc[3] += gqz;
    acc[4] = fmaf(gqx, g.px, acc[4]);  acc[5] = fmaf(gqx, g.py, acc[5]);  acc[6] = fmaf(gqx, g.pz, acc[6]);
    acc[7] = fmaf(gqy, g.px, acc[7]);  acc[8] = fmaf(gqy, g.py, acc[8]);  acc[9] = fmaf(gqy, g.pz, acc[9]);
    acc[10] = fmaf(gqz, g.px, acc[10]); acc[11] = fmaf(gqz, g.py, acc[11]); acc[12] = fmaf(gqz, g.pz, acc[12]);
    // d p / d kld_n = p  =>  d q / d kld_n = R p = q - t
    acc[13] += gqx * (g.qx - c.w.t[0]) + gqy * (g.qy - c.w.t[1]) + gqz * (g.qz - c.w.t[2]);
    acc[14] += c.gain * (sr * itr + sg * itg + sb * itb);
    acc[15] -= sr + sg + sb;
}

// -----------------------------------------------------------------------------------------------------
// mode 1: Gauss-Newton accumulators over x = [tau(3), phi(3), kld_n] (left perturbation Exp(xi)*T)
//   [0] sum |r|   [1..21] H_pp upper triangle (row-major)   [22..27] b_p   [28..33] h_pd   [34] D   [35] b_d
//   [36] number of valid points   [37..39] unused
// r_ch = I_src - I_trg';  J_ch = c_ch * A,  A (2x7) shared by the channels,  c_ch = -gain*[dI/dix, dI/diy];
// IRLS weight of the L1 cost w_ch = 1/max(|r_ch|, eps)  =>  b = J^T sign(r) for |r| > eps.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_gn(const TileCtx& c, const PointGeom& g, const float4 s, float eps,
                                              float (&acc)[SP_GN_PARTIAL_FLOATS]) {
    Taps tp;
    fetch_taps(c.trg4, c.Wl, c.Hl, g.ix, g.iy, tp);
    const float tr[3][4] = {{tp.t00.x, tp.t10.x, tp.t01.x, tp.t11.x},
                            {tp.t00.y, tp.t10.y, tp.t01.y, tp.t11.y},
                            {tp.t00.z, tp.t10.z, tp.t01.z, tp.t11.z}};
    const float sv[3] = {s.x, s.y, s.z};
    float w00 = 0.f, w01 = 0.f, w11 = 0.f, v0 = 0.f, v1 = 0.f, cost = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float dx0 = tr[ch][1] - tr[ch][0], dx1 = tr[ch][3] - tr[ch][2];
        const float dy0 = tr[ch][2] - tr[ch][0], dy1 = tr[ch][3] - tr[ch][1];
        const float it = bilerp(tr[ch][0], tr[ch][1], tr[ch][2], tr[ch][3], tp.wx, tp.wy);
        const float Ix = fmaf(tp.wy, dx1 - dx0, dx0);
        const float Iy = fmaf(tp.wx, dy1 - dy0, dy0);
        const float r = sv[ch] - fmaf(c.gain, it, c.bias);
        const float ar = fabsf(r);
        cost += ar;
        const float wgt = __builtin_amdgcn_rcpf(fmaxf(ar, eps));
        w00 = fmaf(wgt * Ix, Ix, w00);
        w01 = fmaf(wgt * Ix, Iy, w01);
        w11 = fmaf(wgt * Iy, Iy, w11);
        v0 = fmaf(wgt * r, Ix, v0);
        v1 = fmaf(wgt * r, Iy, v1);
    }
    const float g2 = c.gain * c.gain;
    w00 *= g2; w01 *= g2; w11 *= g2;
    v0 *= -c.gain; v1 *= -c.gain;
    // A = diag(alpha) * Jproj * [I | -[q]x | q - t]
    const float zi = g.zguard ? g.zinv : 0.f;   // d(1/z) vanishes on the guarded branch
    const float a = (2.f * c.w.sx * c.w.invWm1) * c.w.Kt.fx * g.zinv;
    const float b = (2.f * c.w.sy * c.w.invHm1) * c.w.Kt.fy * g.zinv;
    const float ux = g.qx * zi, vy = g.qy * zi;
    const float ex = g.qx - c.w.t[0], ey = g.qy - c.w.t[1], ez = g.qz - c.w.t[2];
    const float A0[7] = {a, 0.f, -a * ux, -a * ux * g.qy, a * (g.qz + ux * g.qx), -a * g.qy, a * (ex - ux * ez)};
    const float A1[7] = {0.f, b, -b * vy, -b * (g.qz + vy * g.qy), b * vy * g.qx, b * g.qx, b * (ey - vy * ez)};
    float B0[7], B1[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        B0[j] = fmaf(w00, A0[j], w01 * A1[j]);
        B1[j] = fmaf(w01, A0[j], w11 * A1[j]);
    }
    acc[0] += cost;
    int k = 1;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { acc[k] = fmaf(A0[i], B0[j], fmaf(A1[i], B1[j], acc[k])); ++k; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[22 + i] = fmaf(A0[i], v0, fmaf(A1[i], v1, acc[22 + i]));
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[28 + i] = fmaf(A0[i], B0[6], fmaf(A1[i], B1[6], acc[28 + i]));
    acc[34] = fmaf(A0[6], B0[6], fmaf(A1[6], B1[6], acc[34]));
    acc[35] = fmaf(A0[6], v0, fmaf(A1[6], v1, acc[35]));
    acc[36] += 1.f;
}

template <int MODE>
__device__ __forceinline__ void run_tile(const TileCtx& c, float irls_eps, float* __restrict__ out, float* lds) {
    constexpr int NV = MODE == 0 ? SP_GRAD_PARTIAL_FLOATS : SP_GN_PARTIAL_FLOATS;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    const float ifx = 1.f / c.Ks.fx, ify = 1.f / c.Ks.fy;
    for (int i = threadIdx.x; i < c.count; i += SP_BLOCK) {
        const uint32_t pw = c.pix[c.start + i];
        const float4 s = c.src4[c.start + i];
        float col, row; bool src_ok;
        decode_pix(pw, col, row, src_ok);
        const float d = expf(s.w + c.shift);
        float x, y;
        backproject(col, row, d, c.Ks, ifx, ify, x, y);
        PointGeom g;
        warp_point(c.w, x, y, d, g);
        if (g.valid && src_ok && d > 1e-7f) {
            if (MODE == 0) accumulate_grad(c, g, s, reinterpret_cast<float(&)[SP_GRAD_PARTIAL_FLOATS]>(acc));
            else accumulate_gn(c, g, s, irls_eps, reinterpret_cast<float(&)[SP_GN_PARTIAL_FLOATS]>(acc));
        }
    }
    const float total = block_sum_to_thread<NV>(acc, lds);
    if (threadIdx.x < NV) out[threadIdx.x] = total;
}

__device__ __forceinline__ void fill_warp(TileCtx& c, const float* pose, const Cam& Kt, int H, int W, int Hl, int Wl,
                                          float zmin) {
    load_pose(pose, c.w.R, c.w.t);
    c.w.Kt = Kt;
    c.w.invWm1 = 1.f / (float)(W - 1);
    c.w.invHm1 = 1.f / (float)(H - 1);
    c.w.sx = 0.5f * (float)(Wl - 1);
    c.w.sy = 0.5f * (float)(Hl - 1);
    c.w.zmin = zmin;
    c.Wl = Wl; c.Hl = Hl;
}

// ------------------------------------------------------------------------------------------------
// single source keyframe, B targets: grid = (n_tiles, B)
// ------------------------------------------------------------------------------------------------
struct SingleArgs {
    const uint32_t* pix; const float4* src4; const float* kp_L; const int4* tiles;
    const float* K_src; const float* kld; const float4* trg4; const float* K_trg; const float* pose;
    const float* aff_src; const float* aff_trg;
    int n_tiles, H, W, Hl, Wl;
    float zmin;
};

__global__ __launch_bounds__(SP_BLOCK) void k_cost_single_grad(SingleArgs a, float* __restrict__ partials) {
    __shared__ float lds[SP_WAVES * SP_GRAD_PARTIAL_FLOATS];
    const int t = xcd_chunked_tile(blockIdx.x, a.n_tiles);
    if (t >= a.n_tiles) return;
    const int b = blockIdx.y;
    const int4 tile = a.tiles[t];   // {pair(unused), segment, start, count}
    TileCtx c;
    c.pix = a.pix; c.src4 = a.src4;
    c.trg4 = a.trg4 + (size_t)b * a.Hl * a.Wl;
    load_cam(a.K_src, c.Ks);
    Cam Kt; load_cam(a.K_trg + 9 * b, Kt);
    fill_warp(c, a.pose + 16 * b, Kt, a.H, a.W, a.Hl, a.Wl, a.zmin);
    c.shift = a.kld[tile.y] - a.kp_L[tile.y];
    c.gain = 1.f; c.bias = 0.f;
    if (a.aff_src) {
        c.gain = expf(-(a.aff_trg[2 * b] - a.aff_src[0]));
        c.bias = a.aff_trg[2 * b + 1] - a.aff_src[1];
    }
    c.start = tile.z; c.count = tile.w;
    run_tile<0>(c, 0.f, partials + ((size_t)b * a.n_tiles + t) * SP_GRAD_PARTIAL_FLOATS, lds);
}

// Fixed-order fp64 combination of the tile partials of one target b.
__global__ __launch_bounds__(SP_BLOCK) void k_finalise_single(const float* __restrict__ partials,
                                                               const int32_t* __restrict__ seg_tile_off, int n_tiles,
                                                               int N, int P, bool has_aff, float* residual,
                                                               float* g_kld, float* g_pose, float* g_aff) {
    const int b = blockIdx.x;
    const float* p = partials + (size_t)b * n_tiles * SP_GRAD_PARTIAL_FLOATS;
    const double scale = 1.0 / (3.0 * (double)P);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 15 global sums (all but the per-segment one): wave w handles values w, w+4, ...
    for (int k = wave; k < SP_GRAD_PARTIAL_FLOATS; k += SP_WAVES) {
        if (k == 13) continue;
        double s = 0.0;
        for (int t = lane; t < n_tiles; t += 64) s += (double)p[(size_t)t * SP_GRAD_PARTIAL_FLOATS + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            const float v = (float)(s * scale);
            if (k == 0) residual[b] = v;
            else if (k <= 3) g_pose[b * 16 + (k - 1) * 4 + 3] = v;
            else if (k <= 12) { const int e = k - 4; g_pose[b * 16 + (e / 3) * 4 + (e % 3)] = v; }
            else if (k == 14) { g_aff[b * 4 + 2] = has_aff ? v : 0.f; g_aff[b * 4 + 0] = has_aff ? -v : 0.f; }
            else { g_aff[b * 4 + 3] = has_aff ? v : 0.f; g_aff[b * 4 + 1] = has_aff ? -v : 0.f; }
        }
    }
    if (threadIdx.x < 4) g_pose[b * 16 + 12 + threadIdx.x] = 0.f;
    for (int n = threadIdx.x; n < N; n += SP_BLOCK) {
        double s = 0.0;
        for (int t = seg_tile_off[n]; t < seg_tile_off[n + 1]; ++t) s += (double)p[(size_t)t * SP_GRAD_PARTIAL_FLOATS + 13];
        g_kld[(size_t)b * N + n] = (float)(s * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// many independent pairs: grid = n_tiles_total
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(SP_BLOCK) void k_cost_pairs(const SpPair* __restrict__ pairs, const int4* __restrict__ tiles,
                                                         int n_tiles, float irls_eps, float* __restrict__ partials) {
    constexpr int NV = MODE == 0 ? SP_GRAD_PARTIAL_FLOATS : SP_GN_PARTIAL_FLOATS;
    __shared__ float lds[SP_WAVES * NV];
    const int t = xcd_chunked_tile(blockIdx.x, n_tiles);
    if (t >= n_tiles) return;
    const int4 tile = tiles[t];
    const SpPair& pr = pairs[tile.x];
    TileCtx c;
    c.pix = pr.pix;
    c.src4 = reinterpret_cast<const float4*>(pr.src4);
    c.trg4 = reinterpret_cast<const float4*>(pr.trg4);
    c.Ks = Cam{pr.K_src[0], pr.K_src[1], pr.K_src[2], pr.K_src[3]};
    const Cam Kt{pr.K_trg[0], pr.K_trg[1], pr.K_trg[2], pr.K_trg[3]};
    fill_warp(c, pr.pose, Kt, pr.H, pr.W, pr.Hl, pr.Wl, pr.zmin);
    c.shift = pr.kld[tile.y] - pr.kp_L[tile.y];
    c.gain = 1.f; c.bias = 0.f;
    if (pr.aff) {
        c.gain = expf(-(pr.aff[2] - pr.aff[0]));
        c.bias = pr.aff[3] - pr.aff[1];
    }
    c.start = tile.z; c.count = tile.w;
    run_tile<MODE>(c, irls_eps, partials + (size_t)t * NV, lds);
}

// ------------------------------------------------------------------------------------------------
// per-point diagnostics (collect_stats > 0): grid = (ceil(P/256), B)
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
    const uint32_t* pix; const float4* src4; const int32_t* seg_off; const float* kp_L;
    const float* K_src; const float* kld; const float4* trg4; const float* K_trg; const float* pose;
    const float* aff_src; const float* aff_trg;
    int N, P, H, W, Hl, Wl;
    float zmin;
    float *src_pts, *trg_pts, *src_rgb, *trg_rgb, *raw;
    uint8_t *src_valid, *trg_valid;
    int64_t* seg_ids;
};

__global__ __launch_bounds__(SP_BLOCK) void k_stats(StatsArgs a) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= a.P) return;
    const int b = blockIdx.y;
    // segment of point i: binary search in seg_off
    int lo = 0, hi = a.N;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.seg_off[mid] <= i) lo = mid; else hi = mid; }
    // (empty segments share an offset with their successor: skip forward to the one that owns i)
    while (lo + 1 < a.N && a.seg_off[lo + 1] <= i) ++lo;
    const int n = lo;
    TileCtx c;
    load_cam(a.K_src, c.Ks);
    if (a.trg4) {
        Cam Kt; load_cam(a.K_trg + 9 * b, Kt);
        fill_warp(c, a.pose + 16 * b, Kt, a.H, a.W, a.Hl, a.Wl, a.zmin);
    }
    float gain = 1.f, bias = 0.f;
    if (a.aff_src) { gain = expf(-(a.aff_trg[2 * b] - a.aff_src[0])); bias = a.aff_trg[2 * b + 1] - a.aff_src[1]; }
    const uint32_t pw = a.pix[i];
    const float4 s = a.src4[i];
    float col, row; bool src_ok;
    decode_pix(pw, col, row, src_ok);
    const float d = expf(s.w + (a.kld[n] - a.kp_L[n]));
    src_ok = src_ok && d > 1e-7f;
    float x, y;
    backproject(col, row, d, c.Ks, 1.f / c.Ks.fx, 1.f / c.Ks.fy, x, y);
    const size_t P = a.P;
    if (b == 0) {
        if (a.src_pts) { a.src_pts[3 * i] = x; a.src_pts[3 * i + 1] = y; a.src_pts[3 * i + 2] = d; }
        if (a.src_rgb) { a.src_rgb[i] = s.x; a.src_rgb[P + i] = s.y; a.src_rgb[2 * P + i] = s.z; }
        if (a.src_valid) a.src_valid[i] = src_ok;
        if (a.seg_ids) a.seg_ids[i] = n;
    }
    if (!a.trg4) return;   // source-only query (unproject_kf)
    PointGeom g;
    warp_point(c.w, x, y, d, g);
    Taps tp;
    fetch_taps(a.trg4 + (size_t)b * a.Hl * a.Wl, a.Wl, a.Hl, g.ix, g.iy, tp);
    const float it[3] = {fmaf(gain, bilerp(tp.t00.x, tp.t10.x, tp.t01.x, tp.t11.x, tp.wx, tp.wy), bias),
                         fmaf(gain, bilerp(tp.t00.y, tp.t10.y, tp.t01.y, tp.t11.y, tp.wx, tp.wy), bias),
                         fmaf(gain, bilerp(tp.t00.z, tp.t10.z, tp.t01.z, tp.t11.z, tp.wx, tp.wy), bias)};
    const float sv[3] = {s.x, s.y, s.z};
    const float m = (g.valid && src_ok) ? 1.f : 0.f;
    if (a.trg_pts) { float* q = a.trg_pts + ((size_t)b * P + i) * 3; q[0] = g.qx; q[1] = g.qy; q[2] = g.qz; }
    if (a.trg_valid) a.trg_valid[(size_t)b * P + i] = g.valid;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (a.trg_rgb) a.trg_rgb[((size_t)b * 3 + ch) * P + i] = it[ch];
        if (a.raw) a.raw[((size_t)b * 3 + ch) * P + i] = (sv[ch] - it[ch]) * m;
    }
}

}  // namespace

extern "C" {

int sp_abi_version(void) { return SP_ABI_VERSION; }

int sp_photo_cost_grad(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L,
                       const int32_t* tiles, const int32_t* seg_tile_off, int n_tiles, int N, int P, int H, int W,
                       const float* K_src, const float* kld, const float* trg4, int Hl, int Wl,
                       const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                       float zmin, float* workspace, float* residual, float* g_kld, float* g_pose, float* g_aff,
                       void* stream) {
    (void)seg_off;
    if (!pix || !src4 || !kp_L || !tiles || !seg_tile_off || !K_src || !kld || !trg4 || !K_trg || !pose ||
        !workspace || !residual || !g_kld || !g_pose || !g_aff)
        return SP_EINVAL;
    if (n_tiles <= 0 || N <= 0 || P <= 0 || B <= 0 || H < 2 || W < 2 || Hl < 1 || Wl < 1) return SP_EINVAL;
    if ((aff_src == nullptr) != (aff_trg == nullptr)) return SP_EINVAL;
    if (H > 32767 || W > 65535 || B > 65535) return SP_ELIMIT;
    SingleArgs a{pix, reinterpret_cast<const float4*>(src4), kp_L, reinterpret_cast<const int4*>(tiles),
                 K_src, kld, reinterpret_cast<const float4*>(trg4), K_trg, pose, aff_src, aff_trg,
                 n_tiles, H, W, Hl, Wl, zmin};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_cost_single_grad, dim3(gx, B), dim3(SP_BLOCK), 0, s, a, workspace);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finalise_single, dim3(B), dim3(SP_BLOCK), 0, s, workspace, seg_tile_off, n_tiles, N, P,
                       aff_src != nullptr, residual, g_kld, g_pose, g_aff);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_photo_stats(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L, int N, int P,
                   int H, int W, const float* K_src, const float* kld, const float* trg4, int Hl, int Wl,
                   const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                   float zmin, float* src_pts, float* trg_pts, float* src_rgb, float* trg_rgb, float* raw,
                   uint8_t* src_valid, uint8_t* trg_valid, int64_t* seg_ids, void* stream) {
    if (!pix || !src4 || !seg_off || !kp_L || !K_src || !kld) return SP_EINVAL;
    if (trg4 && (!K_trg || !pose)) return SP_EINVAL;      /* trg4 == NULL: source-side outputs only */
    if (N <= 0 || P <= 0 || B <= 0 || H < 2 || W < 2) return SP_EINVAL;
    if ((aff_src == nullptr) != (aff_trg == nullptr)) return SP_EINVAL;
    StatsArgs a{pix, reinterpret_cast<const float4*>(src4), seg_off, kp_L, K_src, kld,
                reinterpret_cast<const float4*>(trg4), K_trg, pose, aff_src, aff_trg, N, P, H, W, Hl, Wl, zmin,
                src_pts, trg_pts, src_rgb, trg_rgb, raw, src_valid, trg_valid, seg_ids};
    hipLaunchKernelGGL(k_stats, dim3((P + SP_BLOCK - 1) / SP_BLOCK, B), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), a);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pairs_cost(const SpPair* pairs, const int32_t* tiles, int n_tiles_total, int mode, float irls_eps,
                  float* partials, void* stream) {
    if (!pairs || !tiles || !partials || n_tiles_total <= 0 || (mode != 0 && mode != 1)) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = ((n_tiles_total + 7) / 8) * 8;
    const int4* t4 = reinterpret_cast<const int4*>(tiles);
    if (mode == 0)
        hipLaunchKernelGGL(k_cost_pairs<0>, dim3(gx), dim3(SP_BLOCK), 0, s, pairs, t4, n_tiles_total, irls_eps, partials);
    else
        hipLaunchKernelGGL(k_cost_pairs<1>, dim3(gx), dim3(SP_BLOCK), 0, s, pairs, t4, n_tiles_total, irls_eps, partials);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
