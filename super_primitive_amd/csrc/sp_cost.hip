// The fused photometric cost pass: per table point, depth seed -> exp -> unproject -> SE(3) -> project ->
// bilinear sample -> brightness -> masked L1, and in the same pass either the analytic gradient (mode 0) or
// the IRLS-weighted Gauss-Newton normal equations (mode 1).  One workgroup = one tile (a run of points of a
// single segment); per-tile partial sums go to a workspace and are combined in fixed order by a finalise /
// solver kernel, so results are bitwise reproducible.
//
// Replaces core/dense_optim.py:265-403 and core/dense_optim_batch.py:50-147 of the reference (about 150 ATen
// launches forward + the autograd backward) -- see include/sp_hip.h for the ABI and DESIGN.md for the layout.
#include "sp_solve_device.h"

namespace {

// Everything a tile needs; filled from kernel arguments (single source, B targets) or from an SpPair record.
struct TileCtx {
    gptr_u32 pix;
    gptr_f4 src4;
    gptr_f32 trg;         // packed HWC3 target level
    Cam Ks;
    Warp w;
    float shift;      // kld[n] - kp_L[n]
    float gain, bias; // exp(-(a_t-a_s)), b_t-b_s
    float ax, ay;     // level pixels per geometry pixel: (Wl-1)/(W-1), (Hl-1)/(H-1)
    uint32_t row_bytes;   // Wl * 12: byte distance between vertically adjacent target texels
    int Wl, Hl;
    int start, count;
};

// -----------------------------------------------------------------------------------------------------
// The tile loop is a three-stage software pipeline; one trip handles three different points:
//
//     top     issue the (pix, src4) stream of point j+1           (HBM, read once, non-temporal)
//             issue the four bilinear taps of point j             (target image; L1/L2, first touch from HBM)
//     middle  fold point j-1 into the accumulators                (pure arithmetic, covers the loads above)
//     bottom  finish point j: channel mixing of its taps  ->  a handful of scalars carried to the next trip
//             geometry of point j+1: exp, unproject, SE(3), project, validity, tap offsets
//
// Every load is consumed in the trip that issued it, *after* the arithmetic of the previous point, so a wave
// does not sit in s_waitcnt right behind its own gathers and the compiler has no reason to sink the loads.
// All loads are buffer loads (32-bit offsets, SGPR descriptors): out-of-range lanes -- the tail of a tile --
// read zeros, i.e. a pix word whose validity bit is clear, so the loop needs neither clamps nor branches.
// Invalid points (outside the 0.99 band, behind the camera, masked source pixel) sample texel (0,0) and carry
// zinv = 0, m = 0: they add exact zeros.
// -----------------------------------------------------------------------------------------------------
struct Geo {               // what the accumulation of a point needs from its geometry
    float qx, qy, qz;      // point in the target camera
    float zinv, zi;        // m * guarded 1/qz ; m * (1/qz or 0 on the guarded branch, where d(1/z) = 0)
    float px, py, pz;      // point in the source camera (gradient mode: dq/dR)
};

struct Pending {           // a point whose taps are about to be / have just been issued
    Geo g;
    float wx, wy, m;
    float sr, sg, sb;      // source colour
    uint32_t off0;         // byte offset of texel (x0,y0); (x0,y1) is one row further (scalar offset of the load)
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void __attribute__((address_space(1)))* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX));
}
typedef float f32x3 __attribute__((ext_vector_type(3)));
// rgb of one packed HWC3 texel (12 bytes, dword aligned)
template <int AUX = 0>
__device__ __forceinline__ f32x3 buf_load3(rsrc_t r, uint32_t off, uint32_t soff = 0) {
    return __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, (int)soff, AUX));
}

__device__ __forceinline__ void prepare_xyz(const TileCtx& c, float x, float y, float d, bool src_ok, float sr, float sg,
                                            float sb, Pending& p);

template <bool DT = false>
__device__ __forceinline__ void prepare(const TileCtx& c, float shift, float ifx, float ify, uint32_t pw, const f32x4 s, Pending& p) {
    float col, row; bool src_ok;
    decode_pix(pw, col, row, src_ok);
    const float d = DT ? s.w * shift : fast_exp(s.w + shift);
    float x, y;
    backproject(col, row, d, c.Ks, ifx, ify, x, y);
    prepare_xyz(c, x, y, d, src_ok, s.x, s.y, s.z, p);
}

// from the source-camera point on: SE(3), projection, validity, tap offsets (shared with the explicit-point tiles of
// sp_points_cost_grad, whose points come straight from a precomputed dict)
__device__ __forceinline__ void prepare_xyz(const TileCtx& c, float x, float y, float d, bool src_ok, float sr, float sg,
                                            float sb, Pending& p) {
    PointGeom g;
    warp_point(c.w, x, y, d, g);
    const bool ok = g.valid && src_ok && (d > 1e-7f);
    p.m = ok ? 1.f : 0.f;
    p.g.px = x; p.g.py = y; p.g.pz = d;
    p.g.qx = g.qx; p.g.qy = g.qy; p.g.qz = g.qz;
    p.g.zinv = ok ? g.zinv : 0.f;
    p.g.zi = (ok && g.zguard) ? g.zinv : 0.f;
    p.sr = sr; p.sg = sg; p.sb = sb;
    const float ix = ok ? g.ix : 0.f, iy = ok ? g.iy : 0.f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    p.wx = ix - fx0;
    p.wy = iy - fy0;
    // valid => 0 <= x0 <= Wl-2, 0 <= y0 <= Hl-2 (0.99 band, Wl,Hl >= 2 checked on the host)
    // 24-bit integer multiplies, forced (hipcc otherwise re-associates this into v_mad_u64_u32 + v_mul_lo_u32, both
    // quarter rate): texel index < 2^24 for any level up to 4096 x 4096, byte offset < 2^32.
    uint32_t texel, off0;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(texel) : "v"((uint32_t)(int)fy0), "s"((uint32_t)c.Wl), "v"((uint32_t)(int)fx0));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(off0) : "v"(texel), "s"(4u * SP_TEXEL_FLOATS));
    p.off0 = off0;
}

// bilinear value and both slopes of one channel from its four taps
__device__ __forceinline__ void tap_mix(float a, float b, float cc, float d, float wx, float wy, float& it, float& Ix,
                                        float& Iy) {
    const float e1 = b - a, e2 = cc - a, e3 = (d - cc) - e1;
    Iy = fmaf(wx, e3, e2);
    Ix = fmaf(wy, e3, e1);
    it = fmaf(wy, Iy, fmaf(wx, e1, a));
}

// -----------------------------------------------------------------------------------------------------
// mode 0: gradient accumulators
//   [0] sum |r|      [1..3] g_t      [4..12] g_R (row-major)     [13] g_kld(segment of the tile)
//   [14] d/da_t      [15] d/db_t          (all still to be scaled by 1/(3P))
// carried per point: Mix0 = {sum_ch s_ch Ix_ch, sum_ch s_ch Iy_ch, sum_ch s_ch I_ch, sum_ch s_ch}, s_ch = sign(r_ch)
// -----------------------------------------------------------------------------------------------------
struct Mix0 { float gx, gy, s1, s0; };

__device__ __forceinline__ void finish_grad(const TileCtx& c, const Pending& p, const f32x3 a, const f32x3 b,
                                            const f32x3 cc, const f32x3 d, Mix0& o, float& cost_acc) {
    const float ta[3] = {a.x, a.y, a.z}, tb[3] = {b.x, b.y, b.z}, tc[3] = {cc.x, cc.y, cc.z}, td[3] = {d.x, d.y, d.z};
    const float sv[3] = {p.sr, p.sg, p.sb};
    float gx = 0.f, gy = 0.f, s1 = 0.f, s0 = 0.f, cost = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float it, Ix, Iy;
        tap_mix(ta[ch], tb[ch], tc[ch], td[ch], p.wx, p.wy, it, Ix, Iy);
        const float r = sv[ch] - fmaf(c.gain, it, c.bias);
        cost += fabsf(r);
        const float sg = sgn(r);
        gx = fmaf(sg, Ix, gx);
        gy = fmaf(sg, Iy, gy);
        s1 = fmaf(sg, it, s1);
        s0 += sg;
    }
    cost_acc = fmaf(p.m, cost, cost_acc);
    o.gx = gx; o.gy = gy; o.s1 = p.m * s1; o.s0 = p.m * s0;
}

__device__ __forceinline__ void fold_grad(const TileCtx& c, const Geo& g, const Mix0& w,
                                          float (&acc)[SP_GRAD_PARTIAL_FLOATS]) {
    // d|r|/dq through  I_trg' = gain * I(ix,iy) + bias,  ix = (u ...) * ax,  u = fx qx / qz + cx
    const float a = -(c.gain * c.ax * c.w.Kt.fx) * w.gx * g.zinv;
    const float b = -(c.gain * c.ay * c.w.Kt.fy) * w.gy * g.zinv;
    const float gqz = -(a * g.qx + b * g.qy) * g.zi;
    acc[1] += a; acc[2] += b; acc[3] += gqz;
    acc[4] = fmaf(a, g.px, acc[4]);    acc[5] = fmaf(a, g.py, acc[5]);    acc[6] = fmaf(a, g.pz, acc[6]);
    acc[7] = fmaf(b, g.px, acc[7]);    acc[8] = fmaf(b, g.py, acc[8]);    acc[9] = fmaf(b, g.pz, acc[9]);
    acc[10] = fmaf(gqz, g.px, acc[10]); acc[11] = fmaf(gqz, g.py, acc[11]); acc[12] = fmaf(gqz, g.pz, acc[12]);
    // d p / d kld_n = p  =>  d q / d kld_n = R p = q - t
    acc[13] += a * (g.qx - c.w.t[0]) + b * (g.qy - c.w.t[1]) + gqz * (g.qz - c.w.t[2]);
    acc[14] = fmaf(c.gain, w.s1, acc[14]);
    acc[15] -= w.s0;
}

// One tile = one run of points of ONE segment (the single-pair path: sp_photo_cost_grad over the keyframe's own,
// unpadded segment table).  Gradient mode only; the many-pairs path uses the span loops further down.
// PTS = true: the run is a slice of an explicit point list (xyz 12 B + rgb 12 B per point, sp_points_cost_grad) and
// c.pix / c.src4 carry the xyz / rgb base pointers; a lane past the end reads zeros, i.e. depth 0 = invalid.
template <bool PTS>
__device__ __forceinline__ void run_tile(const TileCtx& c, float* __restrict__ out, float* lds) {
    constexpr int NV = SP_GRAD_PARTIAL_FLOATS;
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    const float ifx = 1.f / c.Ks.fx, ify = 1.f / c.Ks.fy;
    const rsrc_t r_pix = PTS ? make_rsrc((gptr_f32)c.pix + 3 * (size_t)c.start, (uint32_t)c.count * 12u)
                             : make_rsrc(c.pix + c.start, (uint32_t)c.count * 4u);
    const rsrc_t r_src = PTS ? make_rsrc((gptr_f32)c.src4 + 3 * (size_t)c.start, (uint32_t)c.count * 12u)
                             : make_rsrc(c.src4 + c.start, (uint32_t)c.count * 16u);
    const rsrc_t r_trg = make_rsrc(c.trg, (uint32_t)c.Wl * (uint32_t)c.Hl * (4u * SP_TEXEL_FLOATS));
    const int n_iter = (c.count + SP_BLOCK - 1) / SP_BLOCK;
    uint32_t i = threadIdx.x;
    constexpr int NT = 2;   // aux: non-temporal
    // the stream of one point: {pix word, src4} or {xyz, rgb}
    auto fetch = [&](uint32_t idx, uint32_t& pw, f32x4& s, f32x3& xyz) {
        if (PTS) {
            xyz = buf_load3<NT>(r_pix, idx * 12u);
            const f32x3 rgb = buf_load3<NT>(r_src, idx * 12u);
            s = f32x4{rgb.x, rgb.y, rgb.z, 0.f};
            pw = 0u;
        } else {
            pw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_pix, (int)(idx * 4u), 0, NT);
            s = buf_load4<NT>(r_src, idx * 16u);
            xyz = f32x3{0.f, 0.f, 0.f};
        }
    };
    auto prep = [&](uint32_t pw, const f32x4& s, const f32x3& xyz, Pending& p) {
        if (PTS) prepare_xyz(c, xyz.x, xyz.y, xyz.z, true, s.x, s.y, s.z, p);
        else prepare(c, c.shift, ifx, ify, pw, s, p);
    };
    // prologue: geometry of point 0
    Pending nx;
    {
        uint32_t pw; f32x4 s; f32x3 xyz;
        fetch(i, pw, s, xyz);
        prep(pw, s, xyz, nx);
    }
    Geo cur = nx.g;
    cur.zinv = 0.f; cur.zi = 0.f;      // "point -1": contributes exact zeros
    Mix0 m0{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < n_iter; ++j) {
        // ---- top: issue everything this trip will need -------------------------------------------
        i += SP_BLOCK;
        uint32_t pw; f32x4 s; f32x3 xyz;
        fetch(i, pw, s, xyz);
        f32x3 ta = buf_load3(r_trg, nx.off0);
        f32x3 tb = buf_load3(r_trg, nx.off0 + 4u * SP_TEXEL_FLOATS);
        f32x3 tc = buf_load3(r_trg, nx.off0, c.row_bytes);          // second row: the scalar offset of the buffer instruction
        f32x3 td = buf_load3(r_trg, nx.off0 + 4u * SP_TEXEL_FLOATS, c.row_bytes);
        // The machine scheduler otherwise sinks the gathers below the arithmetic to shorten their live range
        // (register pressure heuristics): pin the three sections in source order.
        __builtin_amdgcn_sched_barrier(0);
        // ---- middle: fold the previous point (arithmetic only) ----------------------------------
        fold_grad(c, cur, m0, acc);
        __builtin_amdgcn_sched_barrier(0);
        // ---- bottom: finish this point's channel mixing, geometry of the next -------------------
        // An empty asm that "rewrites" the tap registers: the only place their wait (s_waitcnt) may land, and
        // a data dependency that keeps instruction selection from starting the channel mixing before the fold.
        // The accumulators the fold just updated are operands too, so the fold cannot drift below this point.
        asm volatile("" : "+v"(ta), "+v"(tb), "+v"(tc), "+v"(td), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                     "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]));
        finish_grad(c, nx, ta, tb, tc, td, m0, acc[0]);
        cur = nx.g;
        uint32_t pw_ = pw;
        f32x4 s_ = s;
        f32x3 xyz_ = xyz;
        asm volatile("" : "+v"(pw_), "+v"(s_), "+v"(xyz_));
        prep(pw_, s_, xyz_, nx);
    }
    fold_grad(c, cur, m0, acc);
    const float total = block_sum_to_thread<NV>(acc, lds);
    if (threadIdx.x < NV) out[threadIdx.x] = total;
}

// -----------------------------------------------------------------------------------------------------
// mode 1: Gauss-Newton accumulators over x = [tau(3), phi(3), kld_n] (left perturbation Exp(xi)*T)
//   [0] sum |r|   [1..21] H_pp upper triangle (row-major)   [22..27] b_p   [28..33] h_pd   [34] D   [35] b_d
//   [36] number of valid points      (the span loops keep [28..35] per segment and write [0..27] + the count as
//   column 28 of a 32-float pair record, include/sp_hip.h)
// r_ch = I_src - I_trg';  J_ch = c_ch * A,  A (2x7) shared by the channels,  c_ch = -gain*[dI/dix, dI/diy];
// IRLS weight of the L1 cost w_ch = 1/max(|r_ch|, eps)  =>  b = J^T sign(r) for |r| > eps.
// carried per point: {sum w Ix Ix, sum w Ix Iy, sum w Iy Iy, sum w r Ix, sum w r Iy} (Mix2 below)
// A = diag(ga, gb) * Ahat with Ahat0 = [1, 0, -ux, -ux qy, qz + ux qx, -qy, ex - ux ez],
//                              Ahat1 = [0, 1, -vy, -(qz + vy qy), vy qx, qx, ey - vy ez]   (ux = qx/qz, vy = qy/qz):
// the two unit columns make rows 0 and 1 of H plain sums of B = W' Ahat.
// -----------------------------------------------------------------------------------------------------
// -----------------------------------------------------------------------------------------------------
// mode 1, register-pair formulation.  The same sums as fold_gn / finish_gn, arranged so that almost every
// multiply-add is a v_pk_fma_f32 on an aligned register pair (wave64 issues a packed fp32 op in the same four
// cycles as a scalar one):
//   * the red and green channels of a tap are the two halves of one pair, blue is scalar;
//   * the columns of Ahat are kept as pairs along the column index (A0p[k] = {Ahat0[2+2k], Ahat0[3+2k]}), so
//     B = W' Ahat, the 4x4 block of H_pp, b_p and h_pd are all "pair += scalar * pair".
// Two entries of the 4x4 block are accumulated twice ((1,0) next to (1,1), (3,2) next to (3,3)); they are dropped
// when the pairs are written back to the SP_GN_PARTIAL_FLOATS layout.
// -----------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Wave-uniform values that come out of float arithmetic live in VGPRs (the scalar unit has no float ALU) and the
// compiler never moves them back; these put them -- and pairs of them, as one 64-bit scalar -- into SGPRs, where a
// VALU / packed-VALU instruction can read them directly.  25 loop-invariant VGPRs of the GN loop go away this way.
__device__ __forceinline__ float sgpr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x2 sgpr2(float a, float b) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b));
    return __builtin_bit_cast(f32x2, ((uint64_t)hi << 32) | (uint64_t)lo);
}
struct PairConsts {            // the uniform operand pairs of prepare2 / fold_gn2 / finish_gn2
    f32x2 nKc, ifxy, R03, R14, R25, t01, Kf, Ktc, invWH, sxy, gab, bias2, eps2;
    float R6, R7, R8, t2, gain, bias, zmin, eps, row_bytes_f;
};
__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// a.y * b (+ c): the HIGH half of an aligned pair broadcast to both lanes through op_sel.  hipcc broadcasts a pair's low half this way by
// itself but copies a high half into a fresh even register first (v_mov_b32; 22 such copies per point in the Gauss-Newton pass, 13 % of its
// vector instructions) -- here the modifier is spelled out.
__device__ __forceinline__ f32x2 pfma_hi(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 padd2(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 psub2(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pmul_hi(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a.x * b + c with the low half spelled out as well (keeps a and its high-half uses in ONE register pair)
__device__ __forceinline__ f32x2 pfma_lo(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// ... and with the product negated (-a.x * b + c, -a.y * b + c): a sign carried as an operand modifier instead of a v_xor_b32
__device__ __forceinline__ f32x2 pfnma_lo(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pfnma_hi(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// {a.y * b.x + c.x, a.x * b.x + c.y} and {a.y * b.y + c.x, a.x * b.y + c.y}: the halves of a swapped, one half of b for both
__device__ __forceinline__ f32x2 pfma_swz_lo(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pfma_swz_hi(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pmul_lo(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pfma(float a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(f32x2{a, a}, b, c); }

struct GnAcc {
    f32x2 h00;        // acc[1], acc[2]
    f32x2 h0[2];      // acc[3..6]
    f32x2 h1[2];      // acc[8..11]
    f32x2 blk[6];     // rows (0,k0) (0,k1) (1,k0) (1,k1) (2,k1) (3,k1) of the 4x4 block
    f32x2 bp01;       // acc[22], acc[23]
    f32x2 bp[2];      // acc[24..27]
    f32x2 hd01;       // acc[28], acc[29]
    f32x2 hd[2];      // acc[30..33]
    float h11;        // acc[7]
    f32x2 D, bd;      // acc[34], acc[35]: x + y (the two rows' products accumulate side by side: one v_pk_fma_f32 each; added at the flush)
    float cost, n;    // acc[0], acc[36]
    float cost_mark, n_mark;   // cost / n at the last segment flush (flush_segment_gn)
};

struct Pending2 {          // Pending, with the x/y quantities as register pairs
    f32x2 qxy; float qz, zinv, zi;
    f32x2 wxy; float m;
    f32x2 srg; float sb;
    uint32_t off0;
};

// prepare() on pairs: same operations in the same order per component (sp_device.h backproject / warp_point)
template <bool DT = false>
__device__ __forceinline__ void prepare2(const TileCtx& c, const PairConsts& k, float shift, uint32_t pw, const f32x4 s, Pending2& p) {
    const f32x2 colrow{(float)(pw & 0xffffu), (float)((pw >> 16) & 0x7fffu)};
    const bool src_ok = (int32_t)pw < 0;
    const float d = DT ? s.w * shift : fast_exp(s.w + shift);
    const f32x2 xy = pfma(colrow, k.ifxy, k.nKc) * d;          // ((col, row) - (cx, cy)) / (fx, fy) * d, the subtraction folded into the multiply
    // (round 6: the translation rides in the innermost multiply-add -- R d + t as one fma -- instead of a trailing add: one packed and one
    //  scalar instruction less per point; the sum is rounded in another order, a last-bit difference of q)
    const f32x2 qxy = pfma(k.R03, f32x2{xy.x, xy.x}, pfma(k.R14, f32x2{xy.y, xy.y}, pfma(k.R25, f32x2{d, d}, k.t01)));
    const float qz = fmaf(k.R6, xy.x, fmaf(k.R7, xy.y, fmaf(k.R8, d, k.t2)));
    const bool zguard = fabsf(qz) > 1e-6f;
    const float zinv = zguard ? __builtin_amdgcn_rcpf(qz) : 1e-6f;
    const f32x2 uv = qxy * k.Kf * zinv + k.Ktc;
    const f32x2 n = pfma(uv, k.invWH, f32x2{-1.f, -1.f});        // k.invWH = 2 / (W - 1, H - 1): the doubling is exact, folded into the constant
    const bool ok = (fabsf(n.x) <= 0.99f) && (fabsf(n.y) <= 0.99f) && (qz > k.zmin) && src_ok && (d > 1e-7f);
    p.m = ok ? 1.f : 0.f;
    p.qxy = qxy; p.qz = qz;
    p.zinv = ok ? zinv : 0.f;
    p.zi = (ok && zguard) ? zinv : 0.f;
    p.srg = f32x2{s.x, s.y}; p.sb = s.z;
    // (round 6: an invalid point's sampling position is no longer forced to texel (0, 0) -- two selects per point.  It is finite whatever the
    //  point (|q K / z| < 1e12 under the 1e-6 guard of z), its taps go through the buffer descriptor (out of range: zeros), its weights lie
    //  in [0, 1), and everything it contributes is multiplied by the mask (p.m, p.zinv, p.zi = 0) downstream as before)
    const f32x2 ixy = pfma(n, k.sxy, k.sxy);
    const f32x2 fl{floorf(ixy.x), floorf(ixy.y)};
    p.wxy = ixy - fl;
    // byte offset of the upper-left tap, in float arithmetic (exact: below 2^24 for any image the packed pixel word can address at 12 bytes
    // per texel... checked by the host, sp_pairs_cost) and ONE conversion: multiply, fma, convert instead of two conversions, mad, multiply
    p.off0 = (uint32_t)fmaf(fl.y, k.row_bytes_f, fl.x * (float)(4u * SP_TEXEL_FLOATS));
}

struct Mix2 { f32x2 wa, wb, v; };      // {w00, w01}, {w01, w11}, {v0, v1}
// mode 2 (Gauss-Newton WITH the affine brightness pair of the target as unknowns, the window optimiser's flavour): two more
// residual columns per channel, j_a = d r / d a_t = gain * I  and  j_b = d r / d b_t = -1  (r = I_src - (gain I + bias),
// gain = exp(-(a_t - a_s)), bias = b_t - b_s; the source frame's pair enters with the opposite sign).
//   carried per point: ua = {sum w j_a Ix, sum w j_a Iy}, ub = {sum w Ix, sum w Iy}   (cross terms with the geometric columns)
//   accumulated:       H_aa = [sum w j_a^2, -sum w j_a, sum w], b_a = [sum w r j_a, -sum w r]                (pair level)
//                      pa / pb (6 each) = cross terms with the pose columns (pair level), hda / hdb with the depth column (segment)
struct MixA { f32x2 ua, ub; };
struct AffAcc {
    float haa, hab, hbb, ba, bb;
    f32x2 pa01, pa[2], pb01, pb[2];
    float hda, hdb;
};

template <bool AFF = false>
__device__ __forceinline__ void finish_gn2(const PairConsts& k, const Pending2& p, const f32x3 a, const f32x3 b,
                                           const f32x3 cc, const f32x3 d, Mix2& o, float& cost_acc, float& n_acc,
                                           MixA* oa = nullptr, AffAcc* aa = nullptr) {
    const float eps = k.eps;
    // Per channel the pair G = {Ix, Iy} = {e1 + wy e3, e2 + wx e3} (e1 = b - a, e2 = c - a, e3 = (d - c) - e1): the sums over the
    // channels of w G G^T and of w r G are then pair operations whose results ARE the pairs fold_gn2 consumes (a pairing across the
    // channels needs horizontal sums and copies to build them).  e3 of red and green is one pair, used half by half.
    const f32x2 a2{a.x, a.y}, b2{b.x, b.y}, c2{cc.x, cc.y}, d2{d.x, d.y};
    // (spelled out: the compiler splits these three pair operations into six scalar ones)
    const f32x2 e3 = padd2(psub2(psub2(d2, c2), b2), a2);
    const f32x2 Er{b.x - a.x, cc.x - a.x}, Eg{b.y - a.y, cc.y - a.y}, Eb{b.z - a.z, cc.z - a.z};
    f32x2 e3b;                                   // only the low half is read
    e3b.x = (d.z - cc.z) - Eb.x;
    const f32x2 Gr = pfma_swz_lo(p.wxy, e3, Er), Gg = pfma_swz_hi(p.wxy, e3, Eg), Gb = pfma_swz_lo(p.wxy, e3b, Eb);
    const f32x2 it2{fmaf(p.wxy.y, Gr.y, fmaf(p.wxy.x, Er.x, a.x)), fmaf(p.wxy.y, Gg.y, fmaf(p.wxy.x, Eg.x, a.y))};
    const float itb = fmaf(p.wxy.y, Gb.y, fmaf(p.wxy.x, Eb.x, a.z));
    const f32x2 r2 = p.srg - pfma(k.gain, it2, k.bias2);
    const float rb = p.sb - fmaf(k.gain, itb, k.bias2.x);      // (the bias from its vector-register pair: gain is the instruction's one scalar operand)
    const float ar0 = fabsf(r2.x), ar1 = fabsf(r2.y), arb = fabsf(rb);
    // max(|r|, eps) with eps as the instruction's scalar operand (fmaxf() canonicalises the uniform eps into a vector register first: one
    // v_max_f32 per point for nothing)
    auto absmax = [&](float r) { float m; asm("v_max_f32 %0, |%1|, %2" : "=v"(m) : "v"(r), "s"(eps)); return m; };
    const f32x2 wg2{__builtin_amdgcn_rcpf(absmax(r2.x)), __builtin_amdgcn_rcpf(absmax(r2.y))};
    const float wgb = __builtin_amdgcn_rcpf(absmax(rb));
    const f32x2 wGr = pmul_lo(wg2, Gr), wGg = pmul_hi(wg2, Gg), wGb = Gb * wgb;
    o.wa = pfma_lo(wGb, Gb, pfma_lo(wGg, Gg, pmul_lo(wGr, Gr)));       // sum w Ix {Ix, Iy}
    o.wb = pfma_hi(wGb, Gb, pfma_hi(wGg, Gg, pmul_hi(wGr, Gr)));       // sum w Iy {Ix, Iy}
    o.v = pfma(wGb, rb, pfma_hi(r2, wGg, pmul_lo(r2, wGr)));           // sum w r {Ix, Iy}
    cost_acc = fmaf(p.m, (ar0 + ar1) + arb, cost_acc);
    n_acc += p.m;
    if constexpr (AFF) {
        const f32x2 ja2 = k.gain * it2;
        const float jab = k.gain * itb;
        const f32x2 wj2 = wg2 * ja2;
        const float wjb = wgb * jab;
        aa->haa = fmaf(p.m, fmaf(wjb, jab, wj2.x * ja2.x) + wj2.y * ja2.y, aa->haa);
        aa->hab = fmaf(-p.m, (wj2.x + wj2.y) + wjb, aa->hab);
        aa->hbb = fmaf(p.m, (wg2.x + wg2.y) + wgb, aa->hbb);
        aa->ba = fmaf(p.m, fmaf(wjb, rb, wj2.x * r2.x) + wj2.y * r2.y, aa->ba);
        aa->bb = fmaf(-p.m, fmaf(wgb, rb, wg2.x * r2.x) + wg2.y * r2.y, aa->bb);
        oa->ua = pfma(Gb, wjb, pfma_hi(wj2, Gg, pmul_lo(wj2, Gr)));    // sum w j_a {Ix, Iy}
        oa->ub = (wGr + wGg) + wGb;                                    // sum w {Ix, Iy}
    }
}

struct GeoGn { f32x2 qxy; float qz, zinv, zi; };
template <bool AFF = false>
__device__ __forceinline__ void fold_gn2(const PairConsts& k, const GeoGn g, const Mix2 w, GnAcc& A, const MixA* wa = nullptr,
                                         AffAcc* aa = nullptr) {
    const f32x2 g2 = k.gab * g.zinv;   // {ga, gb}; zinv carries the mask
    // (every product with ONE half of a pair as the multiplier goes through pfma_lo / pfma_hi / pmul_lo / pmul_hi: the broadcast is an
    //  operand modifier, no copy)
    const f32x2 Wa = w.wa * pmul_lo(g2, g2);             // {W00, W01} = {w00, w01} * ga * {ga, gb}
    const f32x2 Wb = w.wb * pmul_hi(g2, g2);             // {W01, W11} = {w01, w11} * gb * {ga, gb}
    const f32x2 V = -(w.v * g2);
    const f32x2 qxy = g.qxy;
    const f32x2 u = qxy * g.zi;                          // {ux, vy}
    const float ez = g.qz - k.t2;
    const f32x2 Q = pfma(-ez, u, qxy - k.t01);   // column 6 of Ahat: {A0[4], A1[4]}
    // columns 2..5 of Ahat as pairs along the column index.  Columns 2 and 3 of row 0 and of row 1 are negative throughout: they are
    // kept with the sign flipped (nA00 = -Ahat0[2..3], nA10 = -Ahat1[2..3], hence nB00 = -B00, nB10 = -B10) and the sign goes into the
    // operand modifiers of the products below -- four v_xor_b32 per point otherwise
    float nqy;                                  // (-qxy.y: as `-qxy.y` the compiler negates the whole pair and copies the half it wants)
    asm("v_xor_b32 %0, 0x80000000, %1" : "=v"(nqy) : "v"(qxy.y));
    const f32x2 nA00{u.x, u.x * qxy.y}, A01{fmaf(u.x, qxy.x, g.qz), nqy};
    const f32x2 nA10{u.y, fmaf(u.y, qxy.y, g.qz)}, A11{u.y * qxy.x, qxy.x};
    const f32x2 nB00 = pfma_lo(Wa, nA00, pmul_hi(Wa, nA10)), B01 = pfma_lo(Wa, A01, pmul_hi(Wa, A11));
    const f32x2 nB10 = pfma_lo(Wb, nA00, pmul_hi(Wb, nA10)), B11 = pfma_lo(Wb, A01, pmul_hi(Wb, A11));
    const f32x2 P = pfma_lo(Q, Wa, pmul_hi(Q, Wb));          // {B0[4], B1[4]}
    A.h00 += Wa; A.h11 += Wb.y;
    A.h0[0] -= nB00; A.h0[1] += B01;
    A.h1[0] -= nB10; A.h1[1] += B11;
    A.blk[0] = pfma_lo(nA00, nB00, pfma_lo(nA10, nB10, A.blk[0]));
    A.blk[1] = pfnma_lo(nA00, B01, pfnma_lo(nA10, B11, A.blk[1]));
    A.blk[2] = pfma_hi(nA00, nB00, pfma_hi(nA10, nB10, A.blk[2]));
    A.blk[3] = pfnma_hi(nA00, B01, pfnma_hi(nA10, B11, A.blk[3]));
    A.blk[4] = pfma_lo(A01, B01, pfma_lo(A11, B11, A.blk[4]));
    A.blk[5] = pfma_hi(A01, B01, pfma_hi(A11, B11, A.blk[5]));
    A.bp01 += V;
    A.hd01 += P;
    A.bp[0] = pfnma_lo(V, nA00, pfnma_hi(V, nA10, A.bp[0]));
    A.bp[1] = pfma_lo(V, A01, pfma_hi(V, A11, A.bp[1]));
    A.hd[0] = pfnma_lo(P, nA00, pfnma_hi(P, nA10, A.hd[0]));
    A.hd[1] = pfma_lo(P, A01, pfma_hi(P, A11, A.hd[1]));
    A.D = pfma(Q, P, A.D);
    A.bd = pfma(Q, V, A.bd);
    if constexpr (AFF) {
        // sum_ch w j_a J_geo = -(g2 o ua) Ahat ;  sum_ch w j_b J_geo = +(g2 o ub) Ahat      (g2 carries the mask)
        const f32x2 Ua = -(wa->ua * g2), Ub = wa->ub * g2;
        aa->pa01 += Ua;
        aa->pa[0] = pfnma_lo(Ua, nA00, pfnma_hi(Ua, nA10, aa->pa[0]));
        aa->pa[1] = pfma_lo(Ua, A01, pfma_hi(Ua, A11, aa->pa[1]));
        aa->hda = fmaf(Q.x, Ua.x, fmaf(Q.y, Ua.y, aa->hda));
        aa->pb01 += Ub;
        aa->pb[0] = pfnma_lo(Ub, nA00, pfnma_hi(Ub, nA10, aa->pb[0]));
        aa->pb[1] = pfma_lo(Ub, A01, pfma_hi(Ub, A11, aa->pb[1]));
        aa->hdb = fmaf(Q.x, Ub.x, fmaf(Q.y, Ub.y, aa->hdb));
    }
}

// =====================================================================================================
// The many-pairs path: span loops.
//
// A workgroup processes a SPAN: a run of consecutive chunks of one frame pair; a chunk is a run of points of one
// segment whose length is a multiple of SP_BLOCK (PairBatch pads every segment of its tables with invalid points, which
// contribute exact zeros).  Chunks are contiguous in memory, so the three-stage pipeline of run_tile() simply keeps
// streaming across chunk boundaries: only the segment's depth shift changes (a scalar, fetched one chunk ahead), and
// the few accumulators that belong to the SEGMENT (mode 1: h_pd, D, b_d; mode 0: d/dkld) are flushed -- per wave, no
// barrier -- right after the fold of the chunk's last point.  The accumulators that belong to the PAIR run through
// the whole span and go through the block reduction once.  The pipeline fill/drain, the tile set-up and the 40-value
// block reduction (about 530 instructions per wave, 12 % of a 23-trip tile) are thus paid once per span, not once
// per segment.
//
// Partials: pair sums -> one record per span (span_partials); segment sums -> one record per (chunk, wave)
// (seg_partials, record 4*chunk + wave).  The solvers sum records in index order, so they need not know about spans.
// =====================================================================================================
typedef const float __attribute__((address_space(4)))* cptr_f32;       // constant address space: scalar (s_load) reads
struct ChunkRec { int pair, seg, start, count; };                       // one int32x4 entry of the chunk list
typedef const ChunkRec __attribute__((address_space(4)))* cptr_i4;

struct SpanCursor {            // all wave-uniform
    cptr_i4 chunks;            // {pair, seg, start, count}
    cptr_f32 kld, kp_L;
    int q, q_end;              // chunk being PREPARED, end of the span
    int left;                  // trips of chunk q not yet prepared
    float shift;               // kld[seg] - kp_L[seg] of chunk q
    int nq_left;               // chunk q + 1, fetched one chunk ahead so that no wait lands in the trip loop
    float nq_kld, nq_kpl;
};

template <int TRIP = SP_BLOCK>
__device__ __forceinline__ void cursor_fetch_next(SpanCursor& k) {
    if (k.q + 1 < k.q_end) {
        const int seg = k.chunks[k.q + 1].seg;
        k.nq_left = k.chunks[k.q + 1].count / TRIP;
        k.nq_kld = k.kld[seg];
        k.nq_kpl = k.kp_L[seg];
    }
}

// DT ("depth table"): the tables hold exp(L) in src4.w instead of L, and the cursor's shift is exp(kld - kp_L): a point's depth is then
// one multiply (d = src4.w * shift) instead of add + multiply + v_exp_f32 (a quarter-rate transcendental) -- the exponential is taken
// once per CHUNK here, not once per point
template <bool DT>
__device__ __forceinline__ float cursor_shift(float kld, float kpl) { return DT ? fast_exp(kld - kpl) : kld - kpl; }

template <int TRIP = SP_BLOCK, bool DT = false>
__device__ __forceinline__ void cursor_init(SpanCursor& k, const SpPair& pr, const int4* chunks, int q0, int n) {
    k.chunks = (cptr_i4)chunks;
    k.kld = (cptr_f32)pr.kld;
    k.kp_L = (cptr_f32)pr.kp_L;
    k.q = q0; k.q_end = q0 + n;
    const int seg0 = k.chunks[q0].seg;
    k.left = k.chunks[q0].count / TRIP;
    k.shift = cursor_shift<DT>(k.kld[seg0], k.kp_L[seg0]);
    k.nq_left = 0; k.nq_kld = 0.f; k.nq_kpl = 0.f;
    cursor_fetch_next<TRIP>(k);
}

// Account one prepared trip.  Returns true if that trip was the last of its chunk (whose index is written to `done_q`).
template <int TRIP = SP_BLOCK, bool DT = false>
__device__ __forceinline__ bool cursor_advance(SpanCursor& k, int& done_q) {
    done_q = k.q;
    if (--k.left > 0) return false;
    ++k.q;
    k.left = k.nq_left;
    k.shift = cursor_shift<DT>(k.nq_kld, k.nq_kpl);
    cursor_fetch_next<TRIP>(k);
    return true;
}

template <bool WT>
__device__ __forceinline__ void store_partial(float* p, float v) {
    // fused cost+solve: the record is read by ANOTHER workgroup of this launch -> write-through (sc1) store
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// mode 1: flush {h_pd(6), D, b_d} of the chunk this wave just finished into its segment record (mode 2: + {h_da, h_db})
template <bool WT, bool AFF = false>
__device__ __forceinline__ void flush_segment_gn(GnAcc& A, float* __restrict__ rec, AffAcc* aa = nullptr) {
    const int lane = lane_id();
    int pos; bool ok;
    if constexpr (AFF) {
        float v[10] = {A.hd01.x, A.hd01.y, A.hd[0].x, A.hd[0].y, A.hd[1].x, A.hd[1].y, A.D.x + A.D.y, A.bd.x + A.bd.y, aa->hda, aa->hdb};
        wave_sum_to_lanes<10>(v, lane, pos, ok);
        if (ok) store_partial<WT>(rec + pos, v[0]);
        aa->hda = 0.f; aa->hdb = 0.f;
    } else {
        // [8], [9]: the chunk's own sum |r| and valid points (round 6: the verdict's WITHIN-PAIR test looks at the cost per segment).
        // The pair-level accumulators run on through the span; what they held at the previous flush is kept in two registers (the
        // fold of a chunk's last point comes after that point's finish_gn2, so at this moment they cover exactly the chunks so far)
        float v[10] = {A.hd01.x, A.hd01.y, A.hd[0].x, A.hd[0].y, A.hd[1].x, A.hd[1].y, A.D.x + A.D.y, A.bd.x + A.bd.y,
                       A.cost - A.cost_mark, A.n - A.n_mark};
        wave_sum_to_lanes<10>(v, lane, pos, ok);
        if (ok) store_partial<WT>(rec + pos, v[0]);
        A.cost_mark = A.cost; A.n_mark = A.n;
    }
    const f32x2 z{0.f, 0.f};
    A.hd01 = z; A.hd[0] = z; A.hd[1] = z; A.D = z; A.bd = z;
}

// ABL (developer ablation, only reachable through mode >= 10 of sp_pairs_cost): 0 = product kernel,
// 1 = no target gathers (taps replaced by the source colour), 2 = loads + geometry only (no accumulation),
// 3 (mode 1 only) = the four tap loads made lane-consecutive (coalesced) instead of gathered,
// 5 (mode 16) = NO pix stream: the pixel word of a point comes out of the low mantissa bits of its own src4 colours (tools/kbench.py
//   packs it there for the run): the cheapest conceivable way of "deriving" the pixel -- 6 integer instructions, no load, 16 B per
//   point instead of 20 -- i.e. the upper bound of what replacing pix[P] by run descriptors + a validity mask could gain
__device__ __forceinline__ uint32_t pix_from_colour_bits(const f32x4 s) {
    const uint32_t a = __builtin_bit_cast(uint32_t, s.x), b = __builtin_bit_cast(uint32_t, s.y), c = __builtin_bit_cast(uint32_t, s.z);
    const uint32_t v = (a & 0x7fu) | ((b & 0x7fu) << 7) | ((c & 0x3fu) << 14);            // col (10) | row (9) << 10 | valid << 19
    return (v & 0x3ffu) | (((v >> 10) & 0x1ffu) << 16) | ((v >> 19) << 31);
}
// 6 = DEPTH TABLES (a product form, not an ablation: SP_COST_DEPTH_TABLE / SP_PHASE_DEPTH_TABLE) -- src4.w holds exp(L), see cursor_shift
// W64 ("wave spans"): the span belongs to ONE WAVE -- trips of 64 points, segment records one per chunk, pair-level sums reduced
// over the wave only -- so that the padding granule of the tables is 64 points instead of 256 (small ragged segments: 1200
// SAM-like masks of ~280 pixels pad 40 % at 256 and 11 % at 64); the four waves of a workgroup work on four consecutive spans.
template <int ABL, bool WT, bool AFF = false, bool W64 = false>
__device__ __forceinline__ void run_span_gn(const TileCtx& c, const SpPair& pr, const int4* __restrict__ chunks, int q0,
                                            int n_chunks, int total, float irls_eps, float* __restrict__ span_rec,
                                            float* __restrict__ seg_partials, float* lds) {
    constexpr int TRIP = W64 ? 64 : SP_BLOCK;
    constexpr bool DT = ABL == 6;            // depth tables (cursor_shift)
    constexpr int NV = AFF ? SP_GNA_PARTIAL_FLOATS : SP_GN_PARTIAL_FLOATS, NS = AFF ? SP_GNA_SEG_FLOATS : SP_GN_SEG_FLOATS;
    GnAcc A;
    AffAcc AA;
    MixA ma{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    {
        const f32x2 z{0.f, 0.f};
        AA.haa = AA.hab = AA.hbb = AA.ba = AA.bb = AA.hda = AA.hdb = 0.f;
        AA.pa01 = z; AA.pb01 = z; AA.pa[0] = z; AA.pa[1] = z; AA.pb[0] = z; AA.pb[1] = z;
    }
    {
        const f32x2 z{0.f, 0.f};
        A.h00 = z; A.bp01 = z; A.hd01 = z;
#pragma unroll
        for (int k = 0; k < 2; ++k) { A.h0[k] = z; A.h1[k] = z; A.bp[k] = z; A.hd[k] = z; }
#pragma unroll
        for (int k = 0; k < 6; ++k) A.blk[k] = z;
        A.D = z; A.bd = z;
        A.h11 = A.cost = A.n = A.cost_mark = A.n_mark = 0.f;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SpanCursor k;
    cursor_init<TRIP, DT>(k, pr, chunks, q0, n_chunks);
    const int start = k.chunks[q0].start;
    PairConsts kc;
    {
        const Warp& w = c.w;
        kc.ifxy = sgpr2(1.f / c.Ks.fx, 1.f / c.Ks.fy);
        // (the two pairs that are the ADDEND of a packed fma whose multiplier is an SGPR pair already -- one scalar operand per instruction --
        //  stay in vector registers: from SGPRs they were copied into a register pair again for every point)
        kc.nKc = f32x2{-c.Ks.cx * (1.f / c.Ks.fx), -c.Ks.cy * (1.f / c.Ks.fy)};
        kc.R03 = sgpr2(w.R[0], w.R[3]);        kc.R14 = sgpr2(w.R[1], w.R[4]);      kc.R25 = sgpr2(w.R[2], w.R[5]);
        kc.t01 = f32x2{w.t[0], w.t[1]};        kc.Kf = sgpr2(w.Kt.fx, w.Kt.fy);     kc.Ktc = sgpr2(w.Kt.cx, w.Kt.cy);
        kc.invWH = sgpr2(2.f * w.invWm1, 2.f * w.invHm1);  kc.sxy = sgpr2(w.sx, w.sy);
        kc.gab = sgpr2(c.gain * c.ax * w.Kt.fx, c.gain * c.ay * w.Kt.fy);
        kc.bias2 = f32x2{c.bias, c.bias};      kc.eps2 = sgpr2(irls_eps, irls_eps);
        kc.R6 = sgpr(w.R[6]); kc.R7 = sgpr(w.R[7]); kc.R8 = sgpr(w.R[8]); kc.t2 = w.t[2];
        // (the translation, too, is the addend of a multiply-add whose multiplier -- a row of R -- is the instruction's scalar operand)
        asm volatile("" : "+v"(kc.nKc), "+v"(kc.bias2), "+v"(kc.t01), "+v"(kc.t2));      // (opaque: or the compiler puts the uniform pairs back into SGPRs)
        kc.gain = sgpr(c.gain); kc.bias = sgpr(c.bias); kc.zmin = sgpr(w.zmin); kc.eps = sgpr(irls_eps);
        kc.row_bytes_f = sgpr((float)(c.Wl * (int)(4u * SP_TEXEL_FLOATS)));
    }
    const rsrc_t r_pix = make_rsrc(c.pix + start, (uint32_t)total * 4u);
    const rsrc_t r_src = make_rsrc(c.src4 + start, (uint32_t)total * 16u);
    const rsrc_t r_trg = make_rsrc(c.trg, (uint32_t)c.Wl * (uint32_t)c.Hl * (4u * SP_TEXEL_FLOATS));
    const int n_iter = total / TRIP;
    uint32_t op = (W64 ? (threadIdx.x & 63u) : threadIdx.x) * 4u;
    constexpr int NT = 2;
    // Two point slots used alternately (the loop is unrolled by two with the roles swapped), so that nothing has to be
    // copied at the back edge: point j lives in slot j % 2 from its geometry (bottom of trip j - 1) through its taps and
    // channel mixing (trip j) to its fold (middle of trip j + 1), and the slot is overwritten by point j + 2 only after
    // that fold.  q / last: chunk of the point and whether it closes that chunk (wave-uniform).
    struct Slot { Pending2 p; int q; bool last; };
    Slot S0, S1;
    {
        const f32x4 s = buf_load4<NT>(r_src, op * 4u);
        const uint32_t pw = ABL == 5 ? pix_from_colour_bits(s) : (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_pix, (int)op, 0, NT);
        prepare2<DT>(c, kc, k.shift, pw, s, S0.p);
        S0.last = cursor_advance<TRIP, DT>(k, S0.q);
    }
    S1.p = S0.p;                         // "point -1": finite values, zinv = zi = 0 and a zero Mix2 -> contributes exact zeros
    S1.p.zinv = 0.f; S1.p.zi = 0.f;
    S1.q = q0; S1.last = false;
    Mix2 m{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    // one trip: taps + channel mixing of the point in `a`, fold of the point in `b`, geometry of the next point into `b`
    auto trip = [&](Slot& a, Slot& b) {
        op += 4u * TRIP;
        asm volatile("" : "+v"(op));        // one induction register; the src4 offset is a shift of it
        const uint32_t pw = ABL == 5 ? 0u : (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_pix, (int)op, 0, NT);
        const f32x4 s = buf_load4<NT>(r_src, op * 4u);
        f32x3 ta, tb, tc, td;
        if (ABL == 1) { ta = tb = tc = td = f32x3{a.p.srg.x, a.p.srg.y, a.p.sb}; }
        else if (ABL == 3) {
            // same four 12-byte loads per point, same data volume and real image values, but lane-consecutive texels
            // (fully coalesced): what the bilinear footprint would cost if it were not a gather
            const uint32_t lin = (op >> 2) * (4u * SP_TEXEL_FLOATS);
            const uint32_t cap = (uint32_t)(c.Wl * (c.Hl - 2)) * (4u * SP_TEXEL_FLOATS);
            const uint32_t o0 = lin < cap ? lin : lin - cap * (lin / cap);
            ta = buf_load3(r_trg, o0);
            tb = buf_load3(r_trg, o0 + 4u * SP_TEXEL_FLOATS);
            tc = buf_load3(r_trg, o0, c.row_bytes);
            td = buf_load3(r_trg, o0 + 4u * SP_TEXEL_FLOATS, c.row_bytes);
        } else if (ABL == 4) {
            // upper bound of sharing the right-hand texels with the neighbouring lane: only the left column is loaded
            ta = buf_load3(r_trg, a.p.off0);
            tc = buf_load3(r_trg, a.p.off0, c.row_bytes);
            tb = ta; td = tc;
        } else {
            ta = buf_load3(r_trg, a.p.off0);
            tb = buf_load3(r_trg, a.p.off0 + 4u * SP_TEXEL_FLOATS);
            tc = buf_load3(r_trg, a.p.off0, c.row_bytes);
            td = buf_load3(r_trg, a.p.off0 + 4u * SP_TEXEL_FLOATS, c.row_bytes);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 2) fold_gn2<AFF>(kc, GeoGn{b.p.qxy, b.p.qz, b.p.zinv, b.p.zi}, m, A, &ma, &AA);
        if (b.last) flush_segment_gn<WT, AFF>(A, seg_partials + (size_t)(W64 ? b.q : 4 * b.q + wave) * NS, &AA);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 1)
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(tc), "+v"(td), "+v"(A.blk[0]), "+v"(A.blk[1]), "+v"(A.blk[2]),
                         "+v"(A.blk[3]), "+v"(A.blk[4]), "+v"(A.blk[5]), "+v"(A.bp[0]), "+v"(A.bp[1]), "+v"(A.hd[0]),
                         "+v"(A.hd[1]), "+v"(A.D), "+v"(A.bd), "+v"(A.h00), "+v"(A.h0[0]), "+v"(A.h0[1]), "+v"(A.h1[0]), "+v"(A.h1[1]), "+v"(A.h11), "+v"(A.bp01), "+v"(A.hd01));
        if (ABL == 2) A.cost += ta.x + tb.y + tc.z + td.x + a.p.wxy.x + a.p.wxy.y + a.p.m;
        else finish_gn2<AFF>(kc, a.p, ta, tb, tc, td, m, A.cost, A.n, &ma, &AA);
        uint32_t pw_ = pw;
        f32x4 s_ = s;
        asm volatile("" : "+v"(pw_), "+v"(s_));
        if (ABL == 5) pw_ = pix_from_colour_bits(s_);
        prepare2<DT>(c, kc, k.shift, pw_, s_, b.p);       // (the trip past the end reads zeros and is never used)
        b.last = cursor_advance<TRIP, DT>(k, b.q);
    };
    int j = 0;
    for (; j + 1 < n_iter; j += 2) { trip(S0, S1); trip(S1, S0); }
    if (j < n_iter) {                    // odd trip count: the last point sits in S0
        trip(S0, S1);
        if (ABL != 2) fold_gn2<AFF>(kc, GeoGn{S0.p.qxy, S0.p.qz, S0.p.zinv, S0.p.zi}, m, A, &ma, &AA);
        flush_segment_gn<WT, AFF>(A, seg_partials + (size_t)(W64 ? S0.q : 4 * S0.q + wave) * NS, &AA);
    } else {
        if (ABL != 2) fold_gn2<AFF>(kc, GeoGn{S1.p.qxy, S1.p.qz, S1.p.zinv, S1.p.zi}, m, A, &ma, &AA);
        flush_segment_gn<WT, AFF>(A, seg_partials + (size_t)(W64 ? S1.q : 4 * S1.q + wave) * NS, &AA);
    }
    float acc[NV];
    acc[0] = A.cost;
    acc[1] = A.h00.x; acc[2] = A.h00.y;
    acc[3] = A.h0[0].x; acc[4] = A.h0[0].y; acc[5] = A.h0[1].x; acc[6] = A.h0[1].y;
    acc[7] = A.h11;
    acc[8] = A.h1[0].x; acc[9] = A.h1[0].y; acc[10] = A.h1[1].x; acc[11] = A.h1[1].y;
    acc[12] = A.blk[0].x; acc[13] = A.blk[0].y; acc[14] = A.blk[1].x; acc[15] = A.blk[1].y;
    acc[16] = A.blk[2].y; acc[17] = A.blk[3].x; acc[18] = A.blk[3].y;
    acc[19] = A.blk[4].x; acc[20] = A.blk[4].y; acc[21] = A.blk[5].y;
    acc[22] = A.bp01.x; acc[23] = A.bp01.y;
    acc[24] = A.bp[0].x; acc[25] = A.bp[0].y; acc[26] = A.bp[1].x; acc[27] = A.bp[1].y;
    acc[28] = A.n;
    if constexpr (AFF) {
        acc[29] = AA.haa; acc[30] = AA.hab; acc[31] = AA.hbb; acc[32] = AA.ba; acc[33] = AA.bb;
        acc[34] = AA.pa01.x; acc[35] = AA.pa01.y; acc[36] = AA.pa[0].x; acc[37] = AA.pa[0].y; acc[38] = AA.pa[1].x; acc[39] = AA.pa[1].y;
        acc[40] = AA.pb01.x; acc[41] = AA.pb01.y; acc[42] = AA.pb[0].x; acc[43] = AA.pb[0].y; acc[44] = AA.pb[1].x; acc[45] = AA.pb[1].y;
        acc[46] = acc[47] = 0.f;
    } else {
        acc[29] = acc[30] = acc[31] = 0.f;
    }
    // op still knows the thread index (op = 4 * (threadIdx.x + trips * SP_BLOCK)): nothing derived from threadIdx.x
    // has to stay live, or be spilled, across the loop for the sake of this epilogue
    if (W64) {
        int pos; bool ok;
        wave_sum_to_lanes<NV>(acc, (int)((op >> 2) & 63u), pos, ok);
        if (ok) store_partial<WT>(span_rec + pos, acc[0]);
    } else {
        const int tid = (int)((op >> 2) & (uint32_t)(SP_BLOCK - 1));
        const float tot = block_sum_to_thread<NV>(acc, lds, tid);
        if (tid < NV) store_partial<WT>(span_rec + tid, tot);
    }
}

// mode 0: the segment column is 13 (d/dkld)
template <int ABL, bool WT, bool W64 = false>
__device__ __forceinline__ void run_span_grad(const TileCtx& c, const SpPair& pr, const int4* __restrict__ chunks, int q0,
                                              int n_chunks, int total, float* __restrict__ span_rec,
                                              float* __restrict__ seg_partials, float* lds) {
    constexpr int TRIP = W64 ? 64 : SP_BLOCK;
    constexpr bool DT = ABL == 6;            // depth tables (cursor_shift)
    constexpr int NV = SP_GRAD_PARTIAL_FLOATS;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SpanCursor k;
    cursor_init<TRIP, DT>(k, pr, chunks, q0, n_chunks);
    const int start = k.chunks[q0].start;
    const float ifx = 1.f / c.Ks.fx, ify = 1.f / c.Ks.fy;
    const rsrc_t r_pix = make_rsrc(c.pix + start, (uint32_t)total * 4u);
    const rsrc_t r_src = make_rsrc(c.src4 + start, (uint32_t)total * 16u);
    const rsrc_t r_trg = make_rsrc(c.trg, (uint32_t)c.Wl * (uint32_t)c.Hl * (4u * SP_TEXEL_FLOATS));
    const int n_iter = total / TRIP;
    uint32_t op = (W64 ? (threadIdx.x & 63u) : threadIdx.x) * 4u;
    constexpr int NT = 2;
    // two point slots used alternately, the loop unrolled by two (see run_span_gn): nothing is copied at the back edge
    struct Slot { Pending p; int q; bool last; };
    Slot S0, S1;
    {
        const uint32_t pw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_pix, (int)op, 0, NT);
        const f32x4 s = buf_load4<NT>(r_src, op * 4u);
        prepare<DT>(c, k.shift, ifx, ify, pw, s, S0.p);
        S0.last = cursor_advance<TRIP, DT>(k, S0.q);
    }
    S1.p = S0.p;                         // "point -1": contributes exact zeros
    S1.p.g.zinv = 0.f; S1.p.g.zi = 0.f;
    S1.q = q0; S1.last = false;
    Mix0 m0{0.f, 0.f, 0.f, 0.f};
    auto flush = [&](int q) {
        const float tot = wave_sum(acc[13]);
        if (lane_id() == 0) store_partial<WT>(seg_partials + (size_t)(W64 ? q : 4 * q + wave) * SP_GRAD_SEG_FLOATS, tot);
        acc[13] = 0.f;
    };
    auto trip = [&](Slot& a, Slot& b) {
        op += 4u * TRIP;
        asm volatile("" : "+v"(op));
        const uint32_t pw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_pix, (int)op, 0, NT);
        const f32x4 s = buf_load4<NT>(r_src, op * 4u);
        f32x3 ta, tb, tc, td;
        if (ABL == 1) { ta = tb = tc = td = f32x3{a.p.sr, a.p.sg, a.p.sb}; }
        else {
            ta = buf_load3(r_trg, a.p.off0);
            tb = buf_load3(r_trg, a.p.off0 + 4u * SP_TEXEL_FLOATS);
            tc = buf_load3(r_trg, a.p.off0, c.row_bytes);
            td = buf_load3(r_trg, a.p.off0 + 4u * SP_TEXEL_FLOATS, c.row_bytes);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 2) fold_grad(c, b.p.g, m0, acc);
        if (b.last) flush(b.q);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 1)
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(tc), "+v"(td), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                         "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
                         "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        if (ABL == 2) acc[0] += ta.x + tb.y + tc.z + td.x + a.p.wx + a.p.wy + a.p.m;
        else finish_grad(c, a.p, ta, tb, tc, td, m0, acc[0]);
        uint32_t pw_ = pw;
        f32x4 s_ = s;
        asm volatile("" : "+v"(pw_), "+v"(s_));
        prepare<DT>(c, k.shift, ifx, ify, pw_, s_, b.p);
        b.last = cursor_advance<TRIP, DT>(k, b.q);
    };
    int j = 0;
    for (; j + 1 < n_iter; j += 2) { trip(S0, S1); trip(S1, S0); }
    if (j < n_iter) {                    // odd trip count: the last point sits in S0
        trip(S0, S1);
        if (ABL != 2) fold_grad(c, S0.p.g, m0, acc);
        flush(S0.q);
    } else {
        if (ABL != 2) fold_grad(c, S1.p.g, m0, acc);
        flush(S1.q);
    }
    if (W64) {
        int pos; bool ok;
        wave_sum_to_lanes<NV>(acc, (int)((op >> 2) & 63u), pos, ok);
        if (ok) store_partial<WT>(span_rec + pos, acc[0]);
    } else {
        const int tid = (int)((op >> 2) & (uint32_t)(SP_BLOCK - 1));
        const float tot = block_sum_to_thread<NV>(acc, lds, tid);      // (column 13 is zero here: flushed per chunk)
        if (tid < NV) store_partial<WT>(span_rec + tid, tot);
    }
}

__device__ __forceinline__ void fill_warp(TileCtx& c, const float* pose, const Cam& Kt, int H, int W, int Hl, int Wl,
                                          float zmin) {
    load_pose(pose, c.w.R, c.w.t);
    c.w.Kt = Kt;
    // (results of float arithmetic on uniform values sit in VGPRs unless moved back: sgpr())
    c.w.invWm1 = sgpr(1.f / (float)(W - 1));
    c.w.invHm1 = sgpr(1.f / (float)(H - 1));
    c.w.sx = sgpr(0.5f * (float)(Wl - 1));
    c.w.sy = sgpr(0.5f * (float)(Hl - 1));
    c.w.zmin = zmin;
    c.Wl = Wl; c.Hl = Hl;
    c.row_bytes = (uint32_t)Wl * (4u * SP_TEXEL_FLOATS);
    c.ax = sgpr(2.f * c.w.sx * c.w.invWm1);
    c.ay = sgpr(2.f * c.w.sy * c.w.invHm1);
}


// ------------------------------------------------------------------------------------------------
// single source keyframe, B targets: grid = (n_tiles, B)
// ------------------------------------------------------------------------------------------------
struct SingleArgs {
    const uint32_t* pix; const float4* src4; const float* kp_L; const int4* tiles;
    const float* K_src; const float* kld; const float* trg; const float* K_trg; const float* pose;
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
    c.pix = (gptr_u32)a.pix; c.src4 = (gptr_f4)a.src4;
    c.trg = (gptr_f32)(a.trg + (size_t)b * a.Hl * a.Wl * SP_TEXEL_FLOATS);
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
    run_tile<false>(c, partials + ((size_t)b * a.n_tiles + t) * SP_GRAD_PARTIAL_FLOATS, lds);
}

// ------------------------------------------------------------------------------------------------
// explicit source points (a precomputed dict that did not come from this library's table), B targets:
// grid = (ceil(n / SP_POINT_TILE), B).  24 B per point: xyz + rgb; invalid source points were dropped when the list
// was built, P_total (the reference's denominator, core/dense_optim.py:249-253) is passed separately.
// ------------------------------------------------------------------------------------------------
#define SP_POINT_TILE 2048
struct PointsArgs {
    const float* xyz; const float* rgb; const float* trg; const float* K_trg; const float* pose;
    const float* aff_src; const float* aff_trg;
    int n_points, n_tiles, H, W, Hl, Wl;
    float zmin;
};

__global__ __launch_bounds__(SP_BLOCK) void k_cost_points_grad(PointsArgs a, float* __restrict__ partials) {
    __shared__ float lds[SP_WAVES * SP_GRAD_PARTIAL_FLOATS];
    const int t = xcd_chunked_tile(blockIdx.x, a.n_tiles);
    if (t >= a.n_tiles) return;
    const int b = blockIdx.y;
    TileCtx c;
    c.pix = (gptr_u32)a.xyz; c.src4 = (gptr_f4)a.rgb;
    c.trg = (gptr_f32)(a.trg + (size_t)b * a.Hl * a.Wl * SP_TEXEL_FLOATS);
    c.Ks = Cam{1.f, 1.f, 0.f, 0.f};
    Cam Kt; load_cam(a.K_trg + 9 * b, Kt);
    fill_warp(c, a.pose + 16 * b, Kt, a.H, a.W, a.Hl, a.Wl, a.zmin);
    c.shift = 0.f;
    c.gain = 1.f; c.bias = 0.f;
    if (a.aff_src) {
        c.gain = expf(-(a.aff_trg[2 * b] - a.aff_src[0]));
        c.bias = a.aff_trg[2 * b + 1] - a.aff_src[1];
    }
    c.start = t * SP_POINT_TILE;
    c.count = min(SP_POINT_TILE, a.n_points - c.start);
    run_tile<true>(c, partials + ((size_t)b * a.n_tiles + t) * SP_GRAD_PARTIAL_FLOATS, lds);
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
    for (int n = threadIdx.x; n < N; n += SP_BLOCK) {      // (N = 0 for explicit point lists: no log-depth unknowns)
        double s = 0.0;
        for (int t = seg_tile_off[n]; t < seg_tile_off[n + 1]; ++t) s += (double)p[(size_t)t * SP_GRAD_PARTIAL_FLOATS + 13];
        g_kld[(size_t)b * N + n] = (float)(s * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// many independent pairs: grid = n_tiles_total
// ------------------------------------------------------------------------------------------------
// FUSED = 0: cost pass only.  FUSED = 1 / 2: the workgroup that finishes the LAST tile of a frame pair also runs
// that pair's Adam / Gauss-Newton update (solve_adam / solve_gn), so one optimiser iteration of the whole batch is a
// single launch and solver work overlaps with other pairs' tiles.  Hand-off protocol (cdna_hip_programming.md
// Guideline 16, write-through form): partial stores are sc1, every storing lane drains them (s_waitcnt vmcnt(0)),
// workgroup barrier, one relaxed agent-scope fetch-add on the pair's arrival counter; the last arriver performs one
// agent-scope acquire (invalidates its CU's L1), barrier, then reads the partials with plain loads.  The counter is
// reset by the last arriver, so the buffer stays all-zero between launches.
#define SP_SCHED_LISTS 8
struct SchedList {            // one work list of a launch over SEVERAL (k_cost_pairs with FuseArgs.sched.n_lists > 0)
    const int4* chunks;
    const int4* spans;
    float* partials;
    float* seg_partials;
    int32_t n_spans;
    int32_t first_block;      // its workgroups are blockIdx.x - first_block (a multiple of 8: the XCD of a block stays blockIdx.x % 8)
    uint32_t mask;            // bit p: phase p runs on this list
    int32_t vspans;           // > 0: a queue run's VIRTUAL spans (SpQueue.max_spans): n_spans = slots * vspans, see k_cost_pairs
};
struct SchedCost {            // what the cost kernel needs of an SpSchedule
    const SpPair* pairs[SP_MAX_PHASES];
    float irls_eps[SP_MAX_PHASES];
    uint32_t mask;            // bit p: phase p runs on the work list of this launch (launch over ONE list)
    int32_t vspans;           // ... and that list's virtual spans per slot (queue runs), 0 otherwise
    int32_t n_lists;          // > 0: the launch covers `list[0 .. n_lists)`, one after the other in block order
    SchedList list[SP_SCHED_LISTS];
};
struct FuseArgs {
    int32_t* arrivals;
    AdamArgs adam;
    GnArgs gn;
    const int32_t* done;      // [n_pairs] or NULL: spans of pairs marked done by the solver return at once
    const int32_t* phase;     // per-pair schedules (sp_pairs_schedule_cost): the phase of a pair selects its level descriptors and
    SchedCost sched;          // IRLS epsilon; spans of pairs that are finished, or in a phase of another work list, return at once
    const int32_t* active;    // queue runs, the tail: the launch covers only these slots (SpQueue.active), virtual span v belongs to active[v / vspans]
    const MultiList* multi;   // the launch covers n_multi work lists of DIFFERENT batches (the windows of sp_window_gn_run_multi), one after the other
    int32_t n_multi;
};

template <int MODE, int ABL = 0, int FUSED = 0, bool W64 = false>
__global__ __launch_bounds__(SP_BLOCK, FUSED != 0 ? 1 : (MODE == 2 ? 2 : 4)) void k_cost_pairs(
        const SpPair* __restrict__ pairs, const int4* __restrict__ chunks, const int4* __restrict__ spans, int n_spans,
        float irls_eps, float* __restrict__ partials, float* __restrict__ seg_partials, FuseArgs f) {
    constexpr int NV = MODE == 0 ? SP_GRAD_PARTIAL_FLOATS : (MODE == 2 ? SP_GNA_PARTIAL_FLOATS : SP_GN_PARTIAL_FLOATS);
    __shared__ float lds[SP_WAVES * NV];
    int block = blockIdx.x;
    uint32_t phase_mask = f.sched.mask;
    int vspans = f.sched.vspans;
    if (f.multi) {
        int l = 0;
        for (int i = 1; i < f.n_multi && block >= f.multi[i].first_block; ++i) l = i;      // (wave-uniform: scalar loads)
        const MultiList& ml = f.multi[l];
        block -= ml.first_block;
        pairs = ml.pairs; chunks = reinterpret_cast<const int4*>(ml.chunks); spans = reinterpret_cast<const int4*>(ml.spans);
        n_spans = ml.n_spans; partials = ml.partials; seg_partials = ml.seg_partials;
    }
    if (f.phase && f.sched.n_lists > 0) {
        // One launch over the work lists of a schedule's iteration (the coarse levels' decimated tables and the full one), which
        // took a launch each: the lists of an iteration are independent, mostly small, and one after the other each drained
        // the chip before the next could start.
        int l = 0;
#pragma unroll
        for (int i = 1; i < SP_SCHED_LISTS; ++i)
            if (i < f.sched.n_lists && block >= f.sched.list[i].first_block) l = i;
        const SchedList& sl = f.sched.list[l];
        block -= sl.first_block;
        chunks = sl.chunks; spans = sl.spans; n_spans = sl.n_spans; partials = sl.partials; seg_partials = sl.seg_partials;
        phase_mask = sl.mask;
        vspans = sl.vspans;
    }
    // (wave spans: the four waves of the workgroup take four consecutive spans)
    int w = W64 ? 4 * xcd_chunked_tile(block, (n_spans + 3) >> 2) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
                : xcd_chunked_tile(block, n_spans);
    if (w >= n_spans) return;
    int owner;                              // index of the span's pair in `pairs` / `phase` (a slot, in a queue run)
    int4 span;                              // {first chunk, number of chunks, points, pair}
    if (MODE == 1 && FUSED == 0 && f.phase && vspans > 0) {
        // QUEUE RUN (SpQueue): the launch is over virtual spans, `vspans` per slot; virtual span j of a slot is span tile0 + j of the
        // pair the slot works on right now (its descriptor carries the pair's own span range of the batch's list), or nothing
        owner = w / vspans;
        const int j = w - owner * vspans;
        if (f.active) owner = f.active[owner];
        const int ph = f.phase[owner];
        if (ph >= SP_MAX_PHASES || ph < 0 || !((phase_mask >> ph) & 1u)) return;
        pairs = f.sched.pairs[ph];
        irls_eps = f.sched.irls_eps[ph];
        if (j >= pairs[owner].n_tiles) return;
        w = pairs[owner].tile0 + j;
        span = spans[w];
    } else {
        span = spans[w];
        owner = span.w;
        if (f.done && f.done[owner]) return;    // converged pair (sp_pairs_cost_active): nothing to evaluate
        if (f.phase) {
            const int ph = f.phase[owner];
            if (ph >= SP_MAX_PHASES || ph < 0 || !((phase_mask >> ph) & 1u)) return;
            pairs = f.sched.pairs[ph];
            irls_eps = f.sched.irls_eps[ph];
        }
    }
    const SpPair& pr = pairs[owner];
    TileCtx c;
    c.pix = (gptr_u32)pr.pix;
    c.src4 = (gptr_f4)pr.src4;
    c.trg = (gptr_f32)pr.trg3;
    c.Ks = Cam{pr.K_src[0], pr.K_src[1], pr.K_src[2], pr.K_src[3]};
    const Cam Kt{pr.K_trg[0], pr.K_trg[1], pr.K_trg[2], pr.K_trg[3]};
    fill_warp(c, pr.pose, Kt, pr.H, pr.W, pr.Hl, pr.Wl, pr.zmin);
    c.shift = 0.f;
    c.gain = 1.f; c.bias = 0.f;
    if (pr.aff) {
        c.gain = sgpr(expf(-(pr.aff[2] - pr.aff[0])));
        c.bias = sgpr(pr.aff[3] - pr.aff[1]);
    }
    c.start = 0; c.count = 0;
    if (MODE == 2) run_span_gn<ABL, FUSED != 0, true>(c, pr, chunks, span.x, span.y, span.z, irls_eps, partials + (size_t)w * NV, seg_partials, lds);
    else if (MODE == 1) run_span_gn<ABL, FUSED != 0, false, W64>(c, pr, chunks, span.x, span.y, span.z, irls_eps, partials + (size_t)w * NV, seg_partials, lds);
    else run_span_grad<ABL, FUSED != 0, W64>(c, pr, chunks, span.x, span.y, span.z, partials + (size_t)w * NV, seg_partials, lds);
    if (FUSED != 0) {
        __shared__ int is_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(f.arrivals + span.w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == pr.n_tiles - 1;
            if (last) {
                __hip_atomic_store(f.arrivals + span.w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            is_last = last;
        }
        __syncthreads();
        if (is_last) {
            if (FUSED == 1) solve_adam(pairs, span.w, partials, seg_partials, f.adam);
            else solve_gn(pairs, span.w, partials, seg_partials, f.gn);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-point diagnostics (collect_stats > 0): grid = (ceil(P/256), B)
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
    const uint32_t* pix; const float4* src4; const int32_t* seg_off; const float* kp_L;
    const float* K_src; const float* kld; const float* trg; const float* K_trg; const float* pose;
    const float* aff_src; const float* aff_trg;
    int N, P, H, W, Hl, Wl;
    int stride, Pout;   // every stride-th table point is reported; Pout = ceil(P / stride) rows per output
    float zmin;
    float *src_pts, *trg_pts, *src_rgb, *trg_rgb, *raw;
    uint8_t *src_valid, *trg_valid;
    int64_t* seg_ids;
};

__global__ __launch_bounds__(SP_BLOCK) void k_stats(StatsArgs a) {
    const int o = blockIdx.x * SP_BLOCK + threadIdx.x;       // output row
    if (o >= a.Pout) return;
    const int i = o * a.stride;                              // table point
    const int b = blockIdx.y;
    // segment of point i: binary search in seg_off
    int lo = 0, hi = a.N;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.seg_off[mid] <= i) lo = mid; else hi = mid; }
    // (empty segments share an offset with their successor: skip forward to the one that owns i)
    while (lo + 1 < a.N && a.seg_off[lo + 1] <= i) ++lo;
    const int n = lo;
    TileCtx c;
    load_cam(a.K_src, c.Ks);
    if (a.trg) {
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
    const size_t P = a.Pout;
    if (b == 0) {
        if (a.src_pts) { a.src_pts[3 * o] = x; a.src_pts[3 * o + 1] = y; a.src_pts[3 * o + 2] = d; }
        if (a.src_rgb) { a.src_rgb[o] = s.x; a.src_rgb[P + o] = s.y; a.src_rgb[2 * P + o] = s.z; }
        if (a.src_valid) a.src_valid[o] = src_ok;
        if (a.seg_ids) a.seg_ids[o] = n;
    }
    if (!a.trg) return;   // source-only query (unproject_kf)
    PointGeom g;
    warp_point(c.w, x, y, d, g);
    Taps tp;
    fetch_taps((gptr_f32)(a.trg + (size_t)b * a.Hl * a.Wl * SP_TEXEL_FLOATS), a.Wl, a.Hl, g.ix, g.iy, tp);
    const float it[3] = {fmaf(gain, bilerp(tp.t00.x, tp.t10.x, tp.t01.x, tp.t11.x, tp.wx, tp.wy), bias),
                         fmaf(gain, bilerp(tp.t00.y, tp.t10.y, tp.t01.y, tp.t11.y, tp.wx, tp.wy), bias),
                         fmaf(gain, bilerp(tp.t00.z, tp.t10.z, tp.t01.z, tp.t11.z, tp.wx, tp.wy), bias)};
    const float sv[3] = {s.x, s.y, s.z};
    const float m = (g.valid && src_ok) ? 1.f : 0.f;
    if (a.trg_pts) { float* q = a.trg_pts + ((size_t)b * P + o) * 3; q[0] = g.qx; q[1] = g.qy; q[2] = g.qz; }
    if (a.trg_valid) a.trg_valid[(size_t)b * P + o] = g.valid;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (a.trg_rgb) a.trg_rgb[((size_t)b * 3 + ch) * P + o] = it[ch];
        if (a.raw) a.raw[((size_t)b * 3 + ch) * P + o] = (sv[ch] - it[ch]) * m;
    }
}

}  // namespace

extern "C" {

int sp_abi_version(void) { return SP_ABI_VERSION; }

int sp_photo_cost_grad(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L,
                       const int32_t* tiles, const int32_t* seg_tile_off, int n_tiles, int N, int P, int H, int W,
                       const float* K_src, const float* kld, const float* trg3, int Hl, int Wl,
                       const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                       float zmin, float* workspace, float* residual, float* g_kld, float* g_pose, float* g_aff,
                       void* stream) {
    (void)seg_off;
    if (!pix || !src4 || !kp_L || !tiles || !seg_tile_off || !K_src || !kld || !trg3 || !K_trg || !pose ||
        !workspace || !residual || !g_kld || !g_pose || !g_aff)
        return SP_EINVAL;
    if (n_tiles <= 0 || N <= 0 || P <= 0 || B <= 0 || H < 2 || W < 2 || Hl < 1 || Wl < 1) return SP_EINVAL;
    if ((aff_src == nullptr) != (aff_trg == nullptr)) return SP_EINVAL;
    if (H > 32767 || W > 65535 || B > 65535) return SP_ELIMIT;
    SingleArgs a{pix, reinterpret_cast<const float4*>(src4), kp_L, reinterpret_cast<const int4*>(tiles),
                 K_src, kld, trg3, K_trg, pose, aff_src, aff_trg,
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

int sp_points_workspace_floats(int n_points, int B) {
    if (n_points <= 0 || B <= 0) return 0;
    return ((n_points + SP_POINT_TILE - 1) / SP_POINT_TILE) * B * SP_GRAD_PARTIAL_FLOATS;
}

int sp_points_cost_grad(const float* xyz, const float* rgb, int n_points, int P_total, int H, int W, const float* trg3,
                        int Hl, int Wl, const float* K_trg, const float* pose, int B, const float* aff_src,
                        const float* aff_trg, float zmin, float* workspace, float* residual, float* g_pose, float* g_aff,
                        void* stream) {
    if (!xyz || !rgb || !trg3 || !K_trg || !pose || !workspace || !residual || !g_pose || !g_aff) return SP_EINVAL;
    if (n_points <= 0 || P_total < n_points || B <= 0 || H < 2 || W < 2 || Hl < 1 || Wl < 1) return SP_EINVAL;
    if ((aff_src == nullptr) != (aff_trg == nullptr)) return SP_EINVAL;
    if (B > 65535 || (size_t)n_points * 12u > 0xffffffffu) return SP_ELIMIT;
    const int n_tiles = (n_points + SP_POINT_TILE - 1) / SP_POINT_TILE;
    PointsArgs a{xyz, rgb, trg3, K_trg, pose, aff_src, aff_trg, n_points, n_tiles, H, W, Hl, Wl, zmin};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_cost_points_grad, dim3(gx, B), dim3(SP_BLOCK), 0, s, a, workspace);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finalise_single, dim3(B), dim3(SP_BLOCK), 0, s, workspace, (const int32_t*)nullptr, n_tiles, 0,
                       P_total, aff_src != nullptr, residual, (float*)nullptr, g_pose, g_aff);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_photo_stats(const uint32_t* pix, const float* src4, const int32_t* seg_off, const float* kp_L, int N, int P,
                   int H, int W, const float* K_src, const float* kld, const float* trg3, int Hl, int Wl,
                   const float* K_trg, const float* pose, int B, const float* aff_src, const float* aff_trg,
                   float zmin, float* src_pts, float* trg_pts, float* src_rgb, float* trg_rgb, float* raw,
                   uint8_t* src_valid, uint8_t* trg_valid, int64_t* seg_ids, int stride, void* stream) {
    if (!pix || !src4 || !seg_off || !kp_L || !K_src || !kld || stride < 1) return SP_EINVAL;
    if (trg3 && (!K_trg || !pose)) return SP_EINVAL;      /* trg3 == NULL: source-side outputs only */
    if (N <= 0 || P <= 0 || B <= 0 || H < 2 || W < 2) return SP_EINVAL;
    if ((aff_src == nullptr) != (aff_trg == nullptr)) return SP_EINVAL;
    StatsArgs a{pix, reinterpret_cast<const float4*>(src4), seg_off, kp_L, K_src, kld,
                trg3, K_trg, pose, aff_src, aff_trg, N, P, H, W, Hl, Wl, stride, (P + stride - 1) / stride, zmin,
                src_pts, trg_pts, src_rgb, trg_rgb, raw, src_valid, trg_valid, seg_ids};
    hipLaunchKernelGGL(k_stats, dim3((a.Pout + SP_BLOCK - 1) / SP_BLOCK, B), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), a);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pairs_cost_active(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int mode, float irls_eps,
                         float* partials, float* seg_partials, const int32_t* done, void* stream);

int sp_pairs_cost(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int mode, float irls_eps,
                  float* partials, float* seg_partials, void* stream) {
    return sp_pairs_cost_active(pairs, chunks, spans, n_spans, mode, irls_eps, partials, seg_partials, nullptr, stream);
}

int sp_pairs_cost_active(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int mode, float irls_eps,
                         float* partials, float* seg_partials, const int32_t* done, void* stream) {
    if (!pairs || !chunks || !spans || !partials || !seg_partials || n_spans <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode & SP_COST_WAVE_SPANS) {        // wave-granular work list (granule 64): modes 0 and 1
        const int base = mode & ~(SP_COST_WAVE_SPANS | SP_COST_DEPTH_TABLE);
        const bool dt = (mode & SP_COST_DEPTH_TABLE) != 0 || base == 17;
        if (base != 0 && base != 1 && base != 17) return SP_EINVAL;
        const int gw = ((((n_spans + 3) / 4) + 7) / 8) * 8;
        FuseArgs nf{};
        nf.done = done;
        const int4* c4 = reinterpret_cast<const int4*>(chunks);
        const int4* s4 = reinterpret_cast<const int4*>(spans);
        if (base == 0 && dt)
            hipLaunchKernelGGL((k_cost_pairs<0, 6, 0, true>), dim3(gw), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nf);
        else if (base == 0)
            hipLaunchKernelGGL((k_cost_pairs<0, 0, 0, true>), dim3(gw), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nf);
        else if (dt)
            hipLaunchKernelGGL((k_cost_pairs<1, 6, 0, true>), dim3(gw), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nf);
        else
            hipLaunchKernelGGL((k_cost_pairs<1, 0, 0, true>), dim3(gw), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nf);
        SP_CHECK_LAUNCH();
        return 0;
    }
    if (mode & SP_COST_DEPTH_TABLE) {
        const int base = mode & ~SP_COST_DEPTH_TABLE;
        if (base != 0 && base != 1) return SP_EINVAL;
        mode = base == 0 ? 18 : 17;
    }
    if (mode != 0 && mode != 1 && mode != 2 && !(mode >= 10 && mode <= 18)) return SP_EINVAL;
    const int gx = ((n_spans + 7) / 8) * 8;
    const int4* c4 = reinterpret_cast<const int4*>(chunks);
    const int4* s4 = reinterpret_cast<const int4*>(spans);
    FuseArgs nofuse{};
    nofuse.done = done;
    if (mode == 0)
        hipLaunchKernelGGL(k_cost_pairs<0>, dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 1)
        hipLaunchKernelGGL(k_cost_pairs<1>, dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 2)
        hipLaunchKernelGGL(k_cost_pairs<2>, dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 10)   /* developer ablations, see run_span_gn */
        hipLaunchKernelGGL((k_cost_pairs<0, 1>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 11)
        hipLaunchKernelGGL((k_cost_pairs<1, 1>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 14)
        hipLaunchKernelGGL((k_cost_pairs<1, 3>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 17)
        hipLaunchKernelGGL((k_cost_pairs<1, 6>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 18)
        hipLaunchKernelGGL((k_cost_pairs<0, 6>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 16)
        hipLaunchKernelGGL((k_cost_pairs<1, 5>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 15)
        hipLaunchKernelGGL((k_cost_pairs<1, 4>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else if (mode == 12)
        hipLaunchKernelGGL((k_cost_pairs<0, 2>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    else
        hipLaunchKernelGGL((k_cost_pairs<1, 2>), dim3(gx), dim3(SP_BLOCK), 0, s, pairs, c4, s4, n_spans, irls_eps, partials, seg_partials, nofuse);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

// the cost pass (mode 2: the window optimiser's, or 0 / 1) over n_lists work lists of different batches in ONE launch
int cost_pairs_multi(const MultiList* lists_dev, int n_lists, int total_blocks, int mode, float irls_eps, void* stream) {
    if (!lists_dev || n_lists <= 0 || total_blocks <= 0 || (mode != 0 && mode != 1 && mode != 2)) return SP_EINVAL;
    FuseArgs f{};
    f.multi = lists_dev; f.n_multi = n_lists;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SpPair* np = nullptr; const int4* n4 = nullptr; float* nf = nullptr;
    if (mode == 2) hipLaunchKernelGGL(k_cost_pairs<2>, dim3(total_blocks), dim3(SP_BLOCK), 0, s, np, n4, n4, 0, irls_eps, nf, nf, f);
    else if (mode == 1) hipLaunchKernelGGL(k_cost_pairs<1>, dim3(total_blocks), dim3(SP_BLOCK), 0, s, np, n4, n4, 0, irls_eps, nf, nf, f);
    else hipLaunchKernelGGL(k_cost_pairs<0>, dim3(total_blocks), dim3(SP_BLOCK), 0, s, np, n4, n4, 0, irls_eps, nf, nf, f);
    SP_CHECK_LAUNCH();
    return 0;
}

extern "C" {

int sp_pairs_schedule_cost(const SpSchedule* sched, const int32_t* phase, void* stream) { return schedule_cost_from(sched, phase, stream, 0, nullptr, 0, nullptr, 0, 0u); }

}  // extern "C"

// first_phase: a phase every pair is known to have reached (pairs only move forward): work lists none of whose phases is at or
// beyond it have no pair left and are not launched -- two of the three launches of a frame-pair schedule's iteration through
// its long tail (sp_pairs_schedule_run).
// queue / n_slots: a queue run (sp_pairs_schedule_run_queue) -- every list is launched over n_slots * max_spans virtual spans.
// active / n_active: the tail of a queue run -- only the n_active slots listed in `active` are launched over (sp_pairs_schedule_run_queue)
// idle_mask: phases whose pairs sit this round out (the solver skips them too): work lists ALL of whose phases idle are not launched -- the
//   fine-grained lists of the third attempt's Adam phases while no pair is in one
int schedule_cost_from(const SpSchedule* sched, const int32_t* phase, void* stream, int first_phase, const SpQueue* queue, int n_slots,
                       const int32_t* active, int n_active, uint32_t idle_mask) {
    if (!sched || !phase || sched->n_phases <= 0 || sched->n_phases > SP_MAX_PHASES) return SP_EINVAL;
    if (queue && n_slots <= 0) return SP_EINVAL;
    for (int p = 0; p < sched->n_phases; ++p) {
        const SpPhase& ph = sched->phase[p];
        if (!ph.pairs || !ph.span_partials || !ph.seg_partials || ph.n_spans < 0) return SP_EINVAL;
        if (ph.n_spans > 0 && (!ph.chunks || !ph.spans)) return SP_EINVAL;
    }
    // the distinct work lists that still have pairs (phases sharing `spans` share the list)
    struct Lead { int p; uint32_t mask; int n_spans; int vspans; };
    Lead leads[SP_MAX_PHASES];
    int n_leads = 0;
    uint32_t seen = 0;
    FuseArgs f{};
    f.phase = phase;
    if (queue && active && n_active > 0) { f.active = active; n_slots = n_active; }
    for (int p = 0; p < sched->n_phases; ++p) {
        f.sched.pairs[p] = sched->phase[p].pairs;
        f.sched.irls_eps[p] = sched->phase[p].irls_eps;
    }
    for (int p = 0; p < sched->n_phases; ++p) {
        if ((seen >> p) & 1u) continue;
        const SpPhase& lead = sched->phase[p];          // first phase of a work list
        if (lead.n_spans == 0) continue;                // an empty point set (every segment smaller than the lattice stride)
        uint32_t mask = 0;
        for (int q = p; q < sched->n_phases; ++q) {
            const SpPhase& ph = sched->phase[q];
            if (ph.spans != lead.spans || ph.n_spans != lead.n_spans) continue;
            if (ph.chunks != lead.chunks || ph.span_partials != lead.span_partials || ph.seg_partials != lead.seg_partials ||
                ((ph.flags ^ lead.flags) & (SP_PHASE_WAVE_SPANS | SP_PHASE_DEPTH_TABLE))) return SP_EINVAL;
            if (queue && queue->max_spans[q] != queue->max_spans[p]) return SP_EINVAL;
            mask |= 1u << q;
        }
        seen |= mask;
        if ((mask >> first_phase) == 0u) continue;      // (every phase of this work list lies behind all pairs)
        if ((mask & ~idle_mask) == 0u) continue;        // (every phase of this work list idles this round)
        const int vspans = queue ? queue->max_spans[p] : 0;
        if (queue && vspans == 0) continue;
        leads[n_leads++] = Lead{p, mask, queue ? n_slots * vspans : lead.n_spans, vspans};
    }
    if (n_leads == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const auto blocks_of = [&](const SpPhase& ph, int n_spans) { return (ph.flags & SP_PHASE_WAVE_SPANS) ? ((((n_spans + 3) / 4) + 7) / 8) * 8 : ((n_spans + 7) / 8) * 8; };
    // the four kinds of a scheduled Gauss-Newton pass: wave spans or workgroup spans, depth tables or log-depth tables
    const auto launch_sched = [&](const SpPhase& lead, int n_spans, int blocks) {
        const int4* c4 = reinterpret_cast<const int4*>(lead.chunks);
        const int4* s4 = reinterpret_cast<const int4*>(lead.spans);
        const bool w64 = (lead.flags & SP_PHASE_WAVE_SPANS) != 0, dt = (lead.flags & SP_PHASE_DEPTH_TABLE) != 0;
        if (w64 && dt) hipLaunchKernelGGL((k_cost_pairs<1, 6, 0, true>), dim3(blocks), dim3(SP_BLOCK), 0, s, lead.pairs, c4, s4, n_spans, lead.irls_eps, lead.span_partials, lead.seg_partials, f);
        else if (w64) hipLaunchKernelGGL((k_cost_pairs<1, 0, 0, true>), dim3(blocks), dim3(SP_BLOCK), 0, s, lead.pairs, c4, s4, n_spans, lead.irls_eps, lead.span_partials, lead.seg_partials, f);
        else if (dt) hipLaunchKernelGGL((k_cost_pairs<1, 6>), dim3(blocks), dim3(SP_BLOCK), 0, s, lead.pairs, c4, s4, n_spans, lead.irls_eps, lead.span_partials, lead.seg_partials, f);
        else hipLaunchKernelGGL(k_cost_pairs<1>, dim3(blocks), dim3(SP_BLOCK), 0, s, lead.pairs, c4, s4, n_spans, lead.irls_eps, lead.span_partials, lead.seg_partials, f);
    };
    bool same_kind = n_leads <= SP_SCHED_LISTS;
    for (int i = 1; i < n_leads; ++i) same_kind = same_kind && !((sched->phase[leads[i].p].flags ^ sched->phase[leads[0].p].flags) & (SP_PHASE_WAVE_SPANS | SP_PHASE_DEPTH_TABLE));
    if (n_leads > 1 && same_kind) {
        // ONE launch over all of them: 30.5 k -> 35.3 k frame pairs/s on 384 pairs (a launch per list: each drains the chip before the next starts)
        int total = 0;
        for (int i = 0; i < n_leads; ++i) {
            const SpPhase& ph = sched->phase[leads[i].p];
            f.sched.list[i] = SchedList{reinterpret_cast<const int4*>(ph.chunks), reinterpret_cast<const int4*>(ph.spans), ph.span_partials, ph.seg_partials,
                                        leads[i].n_spans, total, leads[i].mask, leads[i].vspans};
            total += blocks_of(ph, leads[i].n_spans);
        }
        f.sched.n_lists = n_leads;
        const SpPhase& lead = sched->phase[leads[0].p];
        launch_sched(lead, leads[0].n_spans, total);
        SP_CHECK_LAUNCH();
        return 0;
    }
    for (int i = 0; i < n_leads; ++i) {
        const SpPhase& lead = sched->phase[leads[i].p];
        f.sched.mask = leads[i].mask;
        f.sched.vspans = leads[i].vspans;
        launch_sched(lead, leads[i].n_spans, blocks_of(lead, leads[i].n_spans));
        SP_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" {

int sp_pairs_adam_iterate(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int n_pairs, int max_N,
                          float* partials, float* seg_partials, int32_t* arrivals, float lr_kld, float lr_pose, float lr_aff, float* state,
                          float* losses, void* stream) {
    if (!pairs || !chunks || !spans || !partials || !seg_partials || !arrivals || !state || !losses || n_spans <= 0 || n_pairs <= 0 ||
        max_N <= 0)
        return SP_EINVAL;
    FuseArgs f{};
    f.arrivals = arrivals;
    f.adam = AdamArgs{max_N, lr_kld, lr_pose, lr_aff, state, losses};
    const int gx = ((n_spans + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cost_pairs<0, 0, 1>), dim3(gx), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), pairs,
                       reinterpret_cast<const int4*>(chunks), reinterpret_cast<const int4*>(spans), n_spans, 0.f, partials, seg_partials, f);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pairs_gn_iterate(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, int n_pairs, int max_N,
                        float irls_eps, float* partials, float* seg_partials, int32_t* arrivals, float lm_up, float lm_down,
                        float lm_min, float* lm_state, float* backup, float* costs, void* stream) {
    if (!pairs || !chunks || !spans || !partials || !seg_partials || !arrivals || !lm_state || !backup || !costs || n_spans <= 0 || n_pairs <= 0 ||
        max_N <= 0)
        return SP_EINVAL;
    FuseArgs f{};
    f.arrivals = arrivals;
    f.gn = GnArgs{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs, 0.f, nullptr, nullptr, nullptr, 0, 0, 0};
    const int gx = ((n_spans + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cost_pairs<1, 0, 2>), dim3(gx), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), pairs,
                       reinterpret_cast<const int4*>(chunks), reinterpret_cast<const int4*>(spans), n_spans, irls_eps, partials, seg_partials, f);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
