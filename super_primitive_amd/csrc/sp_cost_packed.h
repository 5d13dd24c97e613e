// EXPERIMENT (developer modes 20/21 of sp_pairs_cost): the tile loop with TWO adjacent points per lane, every
// per-point quantity a float2 so that fp32 arithmetic issues as v_pk_* (half the VALU instructions for the
// arithmetic part).  Kept beside the scalar product kernel for A/B measurements; see DESIGN.md §6.
#pragma once
namespace pk {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 vfma(float a, f2 b, f2 c) { return __builtin_elementwise_fma(splat(a), b, c); }
__device__ __forceinline__ f2 vabs(f2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ f2 vsel(i2 m, f2 a, f2 b) { return m ? a : b; }
__device__ __forceinline__ f2 vrcp(f2 a) { return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }

struct Geo {               // what the accumulation of a point pair needs from its geometry
    f2 qx, qy, qz;         // points in the target camera
    f2 zinv, zi;           // m * guarded 1/qz ; m * (1/qz or 0 on the guarded branch, where d(1/z) = 0)
    f2 px, py, pz;         // points in the source camera (gradient mode: dq/dR)
};

struct Pending {           // a point pair whose taps are about to be / have just been issued
    Geo g;
    f2 wx, wy, m;
    f2 sr, sg, sb;         // source colours
    u2 off0, off1;         // byte offsets of texels (x0,y0) and (x0,y1)
};

struct Taps2 {             // raw rgb taps of both points: [point][tap a,b,c,d]
    f32x3 t[2][4];
};


// core/dense_optim.py:19-35,38-86 (depth + unproject), :117-122 (rigid), core/ops.py:19-40 (guarded projection),
// tool/point_utils.py:31-35 (normalise with geometry dims), core/dense_optim.py:128-130,146,160 (validity),
// grid_sample's align_corners un-normalisation -- same operation order as the scalar helpers of sp_device.h.
__device__ __forceinline__ void prepare(const TileCtx& c, float ifx, float ify, u2 pw, const f32x4 sA, const f32x4 sB,
                                        Pending& p) {
    const f2 col = __builtin_convertvector(pw & 0xffffu, f2);
    const f2 row = __builtin_convertvector((pw >> 16) & 0x7fffu, f2);
    const i2 src_ok = __builtin_convertvector(pw >> 31, i2) != 0;
    const f2 d = f2{fast_exp(sA.w + c.shift), fast_exp(sB.w + c.shift)};
    const f2 x = ((col - c.Ks.cx) * d) * ifx;
    const f2 y = ((row - c.Ks.cy) * d) * ify;
    const Warp& w = c.w;
    const f2 qx = vfma(w.R[0], x, vfma(w.R[1], y, w.R[2] * d)) + w.t[0];
    const f2 qy = vfma(w.R[3], x, vfma(w.R[4], y, w.R[5] * d)) + w.t[1];
    const f2 qz = vfma(w.R[6], x, vfma(w.R[7], y, w.R[8] * d)) + w.t[2];
    const i2 zguard = vabs(qz) > 1e-6f;
    const f2 zinv = vsel(zguard, vrcp(qz), splat(1e-6f));
    const f2 u = qx * w.Kt.fx * zinv + w.Kt.cx;
    const f2 v = qy * w.Kt.fy * zinv + w.Kt.cy;
    const f2 xn = 2.f * u * w.invWm1 - 1.f;
    const f2 yn = 2.f * v * w.invHm1 - 1.f;
    const i2 ok = (vabs(xn) <= 0.99f) & (vabs(yn) <= 0.99f) & (qz > w.zmin) & src_ok & (d > 1e-7f);
    const f2 zero = splat(0.f);
    p.m = vsel(ok, splat(1.f), zero);
    p.g.px = x; p.g.py = y; p.g.pz = d;
    p.g.qx = qx; p.g.qy = qy; p.g.qz = qz;
    p.g.zinv = vsel(ok, zinv, zero);
    p.g.zi = vsel(ok & zguard, zinv, zero);
    p.sr = f2{sA.x, sB.x}; p.sg = f2{sA.y, sB.y}; p.sb = f2{sA.z, sB.z};
    const f2 ix = vsel(ok, (xn + 1.f) * w.sx, zero), iy = vsel(ok, (yn + 1.f) * w.sy, zero);
    const f2 fx0 = __builtin_elementwise_floor(ix), fy0 = __builtin_elementwise_floor(iy);
    p.wx = ix - fx0;
    p.wy = iy - fy0;
    // valid => 0 <= x0 <= Wl-2, 0 <= y0 <= Hl-2 (0.99 band, Wl,Hl >= 2 checked on the host)
    const u2 x0 = __builtin_convertvector(__builtin_convertvector(fx0, i2), u2);
    const u2 y0 = __builtin_convertvector(__builtin_convertvector(fy0, i2), u2);
    uint32_t ta, tb, oa, ob;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ta) : "v"(y0.x), "s"((uint32_t)c.Wl), "v"(x0.x));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(tb) : "v"(y0.y), "s"((uint32_t)c.Wl), "v"(x0.y));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(oa) : "v"(ta), "s"(4u * SP_TEXEL_FLOATS));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(ob) : "v"(tb), "s"(4u * SP_TEXEL_FLOATS));
    p.off0 = u2{oa, ob};
    p.off1 = p.off0 + c.row_bytes;
}

// bilinear value and both slopes of one channel from its four taps (both points at once)
__device__ __forceinline__ void tap_mix(f2 a, f2 b, f2 cc, f2 d, f2 wx, f2 wy, f2& it, f2& Ix, f2& Iy) {
    const f2 e1 = b - a, e2 = cc - a, e3 = (d - cc) - e1;
    Iy = vfma(wx, e3, e2);
    Ix = vfma(wy, e3, e1);
    it = vfma(wy, Iy, vfma(wx, e1, a));
}

#define SP_TAP_CH(T, k, ch) f2{(T).t[0][k].ch, (T).t[1][k].ch}

// -----------------------------------------------------------------------------------------------------
// mode 0: gradient accumulators
//   [0] sum |r|      [1..3] g_t      [4..12] g_R (row-major)     [13] g_kld(segment of the tile)
//   [14] d/da_t      [15] d/db_t          (all still to be scaled by 1/(3P))
// carried per point: Mix0 = {sum_ch s_ch Ix_ch, sum_ch s_ch Iy_ch, sum_ch s_ch I_ch, sum_ch s_ch}, s_ch = sign(r_ch)
// -----------------------------------------------------------------------------------------------------
struct Mix0 { f2 gx, gy, s1, s0; };

__device__ __forceinline__ f2 vsgn(f2 v) {
    const f2 one = splat(1.f), zero = splat(0.f);
    return vsel(v > 0.f, one, vsel(v < 0.f, -one, zero));
}

__device__ __forceinline__ void finish_grad(const TileCtx& c, const Pending& p, const Taps2& T, Mix0& o, f2& cost_acc) {
    const f2 ta[3] = {SP_TAP_CH(T, 0, x), SP_TAP_CH(T, 0, y), SP_TAP_CH(T, 0, z)};
    const f2 tb[3] = {SP_TAP_CH(T, 1, x), SP_TAP_CH(T, 1, y), SP_TAP_CH(T, 1, z)};
    const f2 tc[3] = {SP_TAP_CH(T, 2, x), SP_TAP_CH(T, 2, y), SP_TAP_CH(T, 2, z)};
    const f2 td[3] = {SP_TAP_CH(T, 3, x), SP_TAP_CH(T, 3, y), SP_TAP_CH(T, 3, z)};
    const f2 sv[3] = {p.sr, p.sg, p.sb};
    f2 gx = splat(0.f), gy = gx, s1 = gx, s0 = gx, cost = gx;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        f2 it, Ix, Iy;
        tap_mix(ta[ch], tb[ch], tc[ch], td[ch], p.wx, p.wy, it, Ix, Iy);
        const f2 r = sv[ch] - vfma(c.gain, it, splat(c.bias));
        cost += vabs(r);
        const f2 sg = vsgn(r);
        gx = vfma(sg, Ix, gx);
        gy = vfma(sg, Iy, gy);
        s1 = vfma(sg, it, s1);
        s0 += sg;
    }
    cost_acc = vfma(p.m, cost, cost_acc);
    o.gx = gx; o.gy = gy; o.s1 = p.m * s1; o.s0 = p.m * s0;
}

__device__ __forceinline__ void fold_grad(const TileCtx& c, const Geo& g, const Mix0& w, f2 (&acc)[SP_GRAD_PARTIAL_FLOATS]) {
    // d|r|/dq through  I_trg' = gain * I(ix,iy) + bias,  ix = (u ...) * ax,  u = fx qx / qz + cx
    const f2 a = -(c.gain * c.ax * c.w.Kt.fx) * w.gx * g.zinv;
    const f2 b = -(c.gain * c.ay * c.w.Kt.fy) * w.gy * g.zinv;
    const f2 gqz = -(a * g.qx + b * g.qy) * g.zi;
    acc[1] += a; acc[2] += b; acc[3] += gqz;
    acc[4] = vfma(a, g.px, acc[4]);    acc[5] = vfma(a, g.py, acc[5]);    acc[6] = vfma(a, g.pz, acc[6]);
    acc[7] = vfma(b, g.px, acc[7]);    acc[8] = vfma(b, g.py, acc[8]);    acc[9] = vfma(b, g.pz, acc[9]);
    acc[10] = vfma(gqz, g.px, acc[10]); acc[11] = vfma(gqz, g.py, acc[11]); acc[12] = vfma(gqz, g.pz, acc[12]);
    // d p / d kld_n = p  =>  d q / d kld_n = R p = q - t
    acc[13] += a * (g.qx - c.w.t[0]) + b * (g.qy - c.w.t[1]) + gqz * (g.qz - c.w.t[2]);
    acc[14] = vfma(c.gain, w.s1, acc[14]);
    acc[15] -= w.s0;
}

// -----------------------------------------------------------------------------------------------------
// mode 1: Gauss-Newton accumulators over x = [tau(3), phi(3), kld_n] (left perturbation Exp(xi)*T)
//   [0] sum |r|   [1..21] H_pp upper triangle (row-major)   [22..27] b_p   [28..33] h_pd   [34] D   [35] b_d
//   [36] number of valid points   [37..39] unused
// r_ch = I_src - I_trg';  J_ch = c_ch * A,  A (2x7) shared by the channels,  c_ch = -gain*[dI/dix, dI/diy];
// IRLS weight of the L1 cost w_ch = 1/max(|r_ch|, eps)  =>  b = J^T sign(r) for |r| > eps.
// carried per point: Mix1 = {sum w Ix Ix, sum w Ix Iy, sum w Iy Iy, sum w r Ix, sum w r Iy}
// A = diag(ga, gb) * Ahat with Ahat0 = [1, 0, -ux, -ux qy, qz + ux qx, -qy, ex - ux ez],
//                              Ahat1 = [0, 1, -vy, -(qz + vy qy), vy qx, qx, ey - vy ez]   (ux = qx/qz, vy = qy/qz):
// the two unit columns make rows 0 and 1 of H plain sums of B = W' Ahat.
// -----------------------------------------------------------------------------------------------------
struct Mix1 { f2 w00, w01, w11, v0, v1; };

__device__ __forceinline__ void finish_gn(const TileCtx& c, const Pending& p, const Taps2& T, float eps, Mix1& o,
                                          f2& cost_acc, f2& n_acc) {
    const f2 ta[3] = {SP_TAP_CH(T, 0, x), SP_TAP_CH(T, 0, y), SP_TAP_CH(T, 0, z)};
    const f2 tb[3] = {SP_TAP_CH(T, 1, x), SP_TAP_CH(T, 1, y), SP_TAP_CH(T, 1, z)};
    const f2 tc[3] = {SP_TAP_CH(T, 2, x), SP_TAP_CH(T, 2, y), SP_TAP_CH(T, 2, z)};
    const f2 td[3] = {SP_TAP_CH(T, 3, x), SP_TAP_CH(T, 3, y), SP_TAP_CH(T, 3, z)};
    const f2 sv[3] = {p.sr, p.sg, p.sb};
    f2 w00 = splat(0.f), w01 = w00, w11 = w00, v0 = w00, v1 = w00, cost = w00;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        f2 it, Ix, Iy;
        tap_mix(ta[ch], tb[ch], tc[ch], td[ch], p.wx, p.wy, it, Ix, Iy);
        const f2 r = sv[ch] - vfma(c.gain, it, splat(c.bias));
        const f2 ar = vabs(r);
        cost += ar;
        const f2 wgt = vrcp(__builtin_elementwise_max(ar, splat(eps)));
        const f2 wx_ = wgt * Ix, wy_ = wgt * Iy;
        w00 = vfma(wx_, Ix, w00);
        w01 = vfma(wx_, Iy, w01);
        w11 = vfma(wy_, Iy, w11);
        v0 = vfma(wx_, r, v0);
        v1 = vfma(wy_, r, v1);
    }
    cost_acc = vfma(p.m, cost, cost_acc);
    n_acc += p.m;
    o.w00 = w00; o.w01 = w01; o.w11 = w11; o.v0 = v0; o.v1 = v1;
}

__device__ __forceinline__ void fold_gn(const TileCtx& c, const Geo& g, const Mix1& w, f2 (&acc)[SP_GN_PARTIAL_FLOATS]) {
    const f2 ga = (c.gain * c.ax * c.w.Kt.fx) * g.zinv;     // zinv carries the validity mask
    const f2 gb = (c.gain * c.ay * c.w.Kt.fy) * g.zinv;
    const f2 W00 = w.w00 * (ga * ga), W01 = w.w01 * (ga * gb), W11 = w.w11 * (gb * gb);
    const f2 V0 = -w.v0 * ga, V1 = -w.v1 * gb;
    const f2 ux = g.qx * g.zi, vy = g.qy * g.zi;
    const f2 ex = g.qx - c.w.t[0], ey = g.qy - c.w.t[1], ez = g.qz - c.w.t[2];
    // columns 2..6 of Ahat
    const f2 A0[5] = {-ux, -ux * g.qy, vfma(ux, g.qx, g.qz), -g.qy, vfma(-ux, ez, ex)};
    const f2 A1[5] = {-vy, -vfma(vy, g.qy, g.qz), vy * g.qx, g.qx, vfma(-vy, ez, ey)};
    f2 B0[5], B1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        B0[j] = vfma(W00, A0[j], W01 * A1[j]);
        B1[j] = vfma(W01, A0[j], W11 * A1[j]);
    }
    // H_pp upper triangle, row-major: (0,0..5) = acc[1..6], (1,1..5) = acc[7..11], (2,2..5) = acc[12..15],
    // (3,3..5) = acc[16..18], (4,4..5) = acc[19..20], (5,5) = acc[21]
    acc[1] += W00; acc[2] += W01;
    acc[3] += B0[0]; acc[4] += B0[1]; acc[5] += B0[2]; acc[6] += B0[3];
    acc[7] += W11;
    acc[8] += B1[0]; acc[9] += B1[1]; acc[10] += B1[2]; acc[11] += B1[3];
    int k = 12;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) { acc[k] = vfma(A0[i], B0[j], vfma(A1[i], B1[j], acc[k])); ++k; }
    // b_p
    acc[22] += V0; acc[23] += V1;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[24 + i] = vfma(A0[i], V0, vfma(A1[i], V1, acc[24 + i]));
    // coupling with the segment's log-depth (column 6), its diagonal and right-hand side
    acc[28] += B0[4]; acc[29] += B1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[30 + i] = vfma(A0[i], B0[4], vfma(A1[i], B1[4], acc[30 + i]));
    acc[34] = vfma(A0[4], B0[4], vfma(A1[4], B1[4], acc[34]));
    acc[35] = vfma(A0[4], V0, vfma(A1[4], V1, acc[35]));
}

// ABL (developer ablation, only reachable through mode >= 10 of sp_pairs_cost): 0 = product kernel,
// 1 = no target gathers (taps replaced by the source colour), 2 = loads + geometry only (no accumulation)
template <int MODE, int ABL = 0>
__device__ __forceinline__ void run_tile_pk(const TileCtx& c, float irls_eps, float* __restrict__ out, float* lds) {
    constexpr int NV = MODE == 0 ? SP_GRAD_PARTIAL_FLOATS : SP_GN_PARTIAL_FLOATS;
    constexpr uint32_t STEP = 2u * SP_BLOCK;            // points per trip of the workgroup
    f2 acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = splat(0.f);
    const float ifx = 1.f / c.Ks.fx, ify = 1.f / c.Ks.fy;
    const rsrc_t r_pix = make_rsrc(c.pix + c.start, (uint32_t)c.count * 4u);
    const rsrc_t r_src = make_rsrc(c.src4 + c.start, (uint32_t)c.count * 16u);
    const rsrc_t r_trg = make_rsrc(c.trg, (uint32_t)c.Wl * (uint32_t)c.Hl * (4u * SP_TEXEL_FLOATS));
    const int n_iter = (c.count + (int)STEP - 1) / (int)STEP;
    uint32_t i = 2u * threadIdx.x;                      // this lane's first point of the pair
    constexpr int NT = 2;                               // aux: non-temporal
    // prologue: geometry of pair 0
    Pending nx;
    {
        const u2 pw = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(r_pix, (int)(i * 4u), 0, NT));
        const f32x4 sA = buf_load4<NT>(r_src, i * 16u), sB = buf_load4<NT>(r_src, i * 16u + 16u);
        prepare(c, ifx, ify, pw, sA, sB, nx);
    }
    Geo cur = nx.g;
    cur.zinv = splat(0.f); cur.zi = splat(0.f);      // "pair -1": contributes exact zeros
    Mix0 m0{splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    Mix1 m1{splat(0.f), splat(0.f), splat(0.f), splat(0.f), splat(0.f)};
    for (int j = 0; j < n_iter; ++j) {
        // ---- top: issue everything this trip will need -------------------------------------------
        i += STEP;
        u2 pw = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(r_pix, (int)(i * 4u), 0, NT));
        f32x4 sA = buf_load4<NT>(r_src, i * 16u), sB = buf_load4<NT>(r_src, i * 16u + 16u);
        Taps2 T;
        if (ABL == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T.t[0][k] = f32x3{nx.sr.x, nx.sg.x, nx.sb.x};
                T.t[1][k] = f32x3{nx.sr.y, nx.sg.y, nx.sb.y};
            }
        } else {
            T.t[0][0] = buf_load3(r_trg, nx.off0.x);
            T.t[0][1] = buf_load3(r_trg, nx.off0.x + 4u * SP_TEXEL_FLOATS);
            T.t[0][2] = buf_load3(r_trg, nx.off1.x);
            T.t[0][3] = buf_load3(r_trg, nx.off1.x + 4u * SP_TEXEL_FLOATS);
            T.t[1][0] = buf_load3(r_trg, nx.off0.y);
            T.t[1][1] = buf_load3(r_trg, nx.off0.y + 4u * SP_TEXEL_FLOATS);
            T.t[1][2] = buf_load3(r_trg, nx.off1.y);
            T.t[1][3] = buf_load3(r_trg, nx.off1.y + 4u * SP_TEXEL_FLOATS);
        }
        // The machine scheduler otherwise sinks the gathers below the arithmetic to shorten their live range
        // (register pressure heuristics): pin the three sections in source order.
        __builtin_amdgcn_sched_barrier(0);
        // ---- middle: fold the previous pair (arithmetic only) -----------------------------------
        if (ABL != 2) {
            if (MODE == 0) fold_grad(c, cur, m0, reinterpret_cast<f2(&)[SP_GRAD_PARTIAL_FLOATS]>(acc));
            else fold_gn(c, cur, m1, reinterpret_cast<f2(&)[SP_GN_PARTIAL_FLOATS]>(acc));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- bottom: finish this pair's channel mixing, geometry of the next --------------------
        // An empty asm that "rewrites" the tap registers: the only place their wait (s_waitcnt) may land, and a
        // data dependency that keeps instruction selection from starting the channel mixing before the fold;
        // some of the accumulators the fold just updated are operands too, so the fold cannot drift below it.
        if (ABL != 1) {
            if (MODE == 0)
                asm volatile("" : "+v"(T.t[0][0]), "+v"(T.t[0][1]), "+v"(T.t[0][2]), "+v"(T.t[0][3]), "+v"(T.t[1][0]),
                             "+v"(T.t[1][1]), "+v"(T.t[1][2]), "+v"(T.t[1][3]), "+v"(acc[3]), "+v"(acc[6]), "+v"(acc[9]),
                             "+v"(acc[12]), "+v"(acc[13]));
            else
                asm volatile("" : "+v"(T.t[0][0]), "+v"(T.t[0][1]), "+v"(T.t[0][2]), "+v"(T.t[0][3]), "+v"(T.t[1][0]),
                             "+v"(T.t[1][1]), "+v"(T.t[1][2]), "+v"(T.t[1][3]), "+v"(acc[12]), "+v"(acc[15]), "+v"(acc[18]),
                             "+v"(acc[21]), "+v"(acc[24]), "+v"(acc[27]), "+v"(acc[30 % NV]), "+v"(acc[33 % NV]),
                             "+v"(acc[34 % NV]), "+v"(acc[35 % NV]));
        }
        if (ABL == 2) acc[0] += f2{T.t[0][0].x + T.t[0][1].y + T.t[0][2].z + T.t[0][3].x, T.t[1][0].x + T.t[1][1].y + T.t[1][2].z + T.t[1][3].x} + nx.wx + nx.wy + nx.m;
        else if (MODE == 0) finish_grad(c, nx, T, m0, acc[0]);
        else finish_gn(c, nx, T, irls_eps, m1, acc[0], acc[NV - 4]);
        cur = nx.g;
        asm volatile("" : "+v"(pw), "+v"(sA), "+v"(sB));
        prepare(c, ifx, ify, pw, sA, sB, nx);
    }
    if (ABL != 2) {
        if (MODE == 0) fold_grad(c, cur, m0, reinterpret_cast<f2(&)[SP_GRAD_PARTIAL_FLOATS]>(acc));
        else fold_gn(c, cur, m1, reinterpret_cast<f2(&)[SP_GN_PARTIAL_FLOATS]>(acc));
    }
    float accs[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) accs[k] = acc[k].x + acc[k].y;
    const float total = block_sum_to_thread<NV>(accs, lds);
    if (threadIdx.x < NV) out[threadIdx.x] = total;
}


}  // namespace pk
