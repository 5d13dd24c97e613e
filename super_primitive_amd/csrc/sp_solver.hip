// Per-pair parameter updates that follow the cost pass: tile-partial reduction (fixed order, fp64), then either
// an Adam step with SE(3) retraction (the reference's optimiser: torch.optim.Adam over log-depths, a lietorch
// pose tangent and the affine pair -- odometery/two_frame_sfm.py:116-123,201-206, odometery/odometery.py:300-312,
// 394-403) or a Gauss-Newton / Levenberg-Marquardt step (the solver BASELINE.json's north_star adds).
// One workgroup per frame pair; no host synchronisation, no atomics.
#include "sp_solve_device.h"

namespace {

__global__ __launch_bounds__(SP_BLOCK) void k_pairs_adam(const SpPair* __restrict__ pairs, const float* __restrict__ partials,
                                                         const float* __restrict__ seg_partials, AdamArgs h) {
    solve_adam(pairs, blockIdx.x, partials, seg_partials, h);
}

__global__ __launch_bounds__(SP_BLOCK) void k_pairs_gn(const SpPair* __restrict__ pairs, const float* __restrict__ partials,
                                                       const float* __restrict__ seg_partials, GnArgs h) {
    solve_gn(pairs, blockIdx.x, partials, seg_partials, h);
}


// lie/lie_algebra.py:41-119: R -> best-conditioned quaternion -> R, in place, one thread per matrix
__global__ void k_renormalise(float* __restrict__ T, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* M = T + 16 * (size_t)i;
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    float qa[4] = {1.f + m00 + m11 + m22, 1.f + m00 - m11 - m22, 1.f - m00 + m11 - m22, 1.f - m00 - m11 + m22};
    for (int k = 0; k < 4; ++k) qa[k] = qa[k] > 0.f ? sqrtf(qa[k]) : 0.f;
    int best = 0;
    for (int k = 1; k < 4; ++k) if (qa[k] > qa[best]) best = k;   // argmax, first maximum like torch
    const float cand[4][4] = {{qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
                              {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
                              {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
                              {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    const float den = 2.f * fmaxf(qa[best], 0.1f);
    const float r = cand[best][0] / den, x = cand[best][1] / den, y = cand[best][2] / den, z = cand[best][3] / den;
    const float s = 2.f / (r * r + x * x + y * y + z * z);
    M[0] = 1.f - s * (y * y + z * z); M[1] = s * (x * y - z * r);       M[2] = s * (x * z + y * r);
    M[4] = s * (x * y + z * r);       M[5] = 1.f - s * (x * x + z * z); M[6] = s * (y * z - x * r);
    M[8] = s * (x * z - y * r);       M[9] = s * (y * z + x * r);       M[10] = 1.f - s * (x * x + y * y);
}

// ---------------------------------------------------------------------------------------------------
// Pose parameter of the reference's drivers: T = Exp(a) * X with a = [tau, phi] (lietorch LieGroupParameter.retr()
// followed by .matrix(), odometery/two_frame_sfm.py:83-84, odometery/odometery.py:224-228).  As eager torch code
// the exponential and its autograd graph are ~100 tiny launches (1.5 ms per iteration measured); here forward
// and backward are one launch each.  The backward pass differentiates the very same code with forward-mode dual
// numbers (6 tangents at once), so value and derivative cannot drift apart, also at a = 0.
// ---------------------------------------------------------------------------------------------------
template <int ND>
struct Dual {
    float v;
    float d[ND];
};
template <int ND> __device__ __forceinline__ Dual<ND> dconst(float c) { Dual<ND> r; r.v = c; for (int i = 0; i < ND; ++i) r.d[i] = 0.f; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v + b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v - b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a) { Dual<ND> r; r.v = -a.v; for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i]; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> operator*(float a, const Dual<ND>& b) { Dual<ND> r; r.v = a * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a * b.d[i]; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> chain(const Dual<ND>& a, float f, float df) { Dual<ND> r; r.v = f; for (int i = 0; i < ND; ++i) r.d[i] = df * a.d[i]; return r; }

// T(3x4) = Exp(a) * X(3x4) on dual numbers; the three coefficient functions are evaluated in theta^2
template <int ND>
__device__ void se3_exp_times(const Dual<ND> (&a)[6], const float* __restrict__ X16, Dual<ND> (&T)[12]) {
    const Dual<ND> th2 = a[3] * a[3] + a[4] * a[4] + a[5] * a[5];
    const float t2 = th2.v;
    float A, B, C, dA, dB, dC;     // values and derivatives with respect to theta^2
    {   // evaluated in fp64: the closed forms cancel badly in fp32 for theta < ~0.5 (1 - cos, theta - sin)
        const double t = (double)t2;
        double a_, b_, c_, da_, db_, dc_;
        if (t < 1e-6) {
            a_ = 1.0 - t / 6.0 * (1.0 - t / 20.0);           da_ = -1.0 / 6.0 + t / 60.0;
            b_ = 0.5 - t / 24.0 * (1.0 - t / 30.0);           db_ = -1.0 / 24.0 + t / 360.0;
            c_ = 1.0 / 6.0 - t / 120.0 * (1.0 - t / 42.0);    dc_ = -1.0 / 120.0 + t / 2520.0;
        } else {
            const double th = sqrt(t), sn = sin(th), cs = cos(th);
            a_ = sn / th; b_ = (1.0 - cs) / t; c_ = (th - sn) / (t * th);
            // d/d(theta^2) = (1 / (2 theta)) d/dtheta
            da_ = (cs * th - sn) / (2.0 * t * th);
            db_ = (sn * th - 2.0 * (1.0 - cs)) / (2.0 * t * t);
            dc_ = ((1.0 - cs) * th - 3.0 * (th - sn)) / (2.0 * t * t * th);
        }
        A = (float)a_; B = (float)b_; C = (float)c_; dA = (float)da_; dB = (float)db_; dC = (float)dc_;
    }
    const Dual<ND> dA_ = chain(th2, A, dA), dB_ = chain(th2, B, dB), dC_ = chain(th2, C, dC);
    const Dual<ND> z = dconst<ND>(0.f), one = dconst<ND>(1.f);
    const Dual<ND> W[9] = {z, -a[5], a[4], a[5], z, -a[3], -a[4], a[3], z};
    Dual<ND> W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    Dual<ND> E[9], V[9];
    for (int i = 0; i < 9; ++i) {
        const Dual<ND> I = (i % 4 == 0) ? one : z;
        E[i] = I + dA_ * W[i] + dB_ * W2[i];
        V[i] = I + dB_ * W[i] + dC_ * W2[i];
    }
    for (int i = 0; i < 3; ++i) {
        const Dual<ND> dt = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
        for (int j = 0; j < 3; ++j)
            T[4 * i + j] = X16[j] * E[3 * i] + X16[4 + j] * E[3 * i + 1] + X16[8 + j] * E[3 * i + 2];
        T[4 * i + 3] = X16[3] * E[3 * i] + X16[7] * E[3 * i + 1] + X16[11] * E[3 * i + 2] + dt;
    }
}

// one thread per pose; G = nullptr: forward (writes T), else backward (writes ga = (dT/da)^T G)
__global__ void k_se3_retract(const float* __restrict__ a6, const float* __restrict__ X, int n, float* __restrict__ T,
                              const float* __restrict__ G, float* __restrict__ ga) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const float* a = a6 + 6 * (size_t)b;
    const float* Xb = X + 16 * (size_t)b;
    if (!G) {
        Dual<1> ad[6], out[12];
        for (int i = 0; i < 6; ++i) { ad[i].v = a[i]; ad[i].d[0] = 0.f; }
        se3_exp_times<1>(ad, Xb, out);
        float* Tb = T + 16 * (size_t)b;
        for (int i = 0; i < 12; ++i) Tb[i] = out[i].v;
        Tb[12] = 0.f; Tb[13] = 0.f; Tb[14] = 0.f; Tb[15] = 1.f;
    } else {
        Dual<6> ad[6], out[12];
        for (int i = 0; i < 6; ++i) { ad[i].v = a[i]; for (int k = 0; k < 6; ++k) ad[i].d[k] = (i == k) ? 1.f : 0.f; }
        se3_exp_times<6>(ad, Xb, out);
        const float* Gb = G + 16 * (size_t)b;
        for (int k = 0; k < 6; ++k) {
            float s = 0.f;
            for (int i = 0; i < 12; ++i) s = fmaf(out[i].d[k], Gb[i], s);
            ga[6 * (size_t)b + k] = s;
        }
    }
}

}  // namespace

extern "C" {

int sp_pairs_adam_step(const SpPair* pairs, int n_pairs, int max_N, const float* partials, const float* seg_partials,
                       float lr_kld, float lr_pose, float lr_aff, float* state, float* losses, void* stream) {
    if (!pairs || !partials || !seg_partials || !state || !losses || n_pairs <= 0 || max_N <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_pairs_adam, dim3(n_pairs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), pairs, partials,
                       seg_partials, AdamArgs{max_N, lr_kld, lr_pose, lr_aff, state, losses});
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pairs_gn_step(const SpPair* pairs, int n_pairs, int max_N, const float* partials, const float* seg_partials,
                     float lm_up, float lm_down, float lm_min, float* lm_state, float* backup, float* costs, void* stream) {
    if (!pairs || !partials || !seg_partials || !lm_state || !backup || !costs || n_pairs <= 0 || max_N <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_pairs_gn, dim3(n_pairs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), pairs, partials,
                       seg_partials, GnArgs{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs});
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_renormalise_se3(float* T, int n, void* stream) {
    if (!T || n <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_renormalise, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), T, n);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_se3_retract(const float* a, const float* X, int n, float* T, const float* grad_T, float* grad_a, void* stream) {
    if (!a || !X || n <= 0) return SP_EINVAL;
    if (grad_T ? !grad_a : !T) return SP_EINVAL;
    hipLaunchKernelGGL(k_se3_retract, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a, X, n, T, grad_T,
                       grad_a);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
