// Device-side solver bodies shared by the stand-alone solver kernels (sp_solver.hip) and the fused
// cost+solve kernels (sp_cost.hip): per-pair tile-partial reduction (fixed order, fp64), Adam step with SE(3)
// retraction, Gauss-Newton / Levenberg-Marquardt step.  See sp_solver.hip for the reference lines they replace.
#pragma once
#include "sp_device.h"

#define SP_LM_STRIDE SP_LM_STATE_FLOATS

// All NV columns of the pair's tile partials summed over its tiles, into out[NV] (LDS).  Thread (g, c) = (tid / NV,
// tid % NV) walks tiles g, g+G, ... of column c: neighbouring threads read neighbouring floats of one 4*NV-byte
// tile record (coalesced), every thread's loads are independent (pipelined), and the G group partials are added
// in fixed order -- one barrier, bitwise reproducible.
template <int NV>
__device__ __forceinline__ void reduce_columns(const float* __restrict__ p, int n_tiles, double* out, double* scratch) {
    constexpr int G = SP_BLOCK / NV;
    const int g = threadIdx.x / NV, c = threadIdx.x - g * NV;
    if (g < G) {
        double s = 0.0;
        // 8 independent loads in flight per trip (the additions keep their fixed order)
        for (int t = g; t < n_tiles; t += 8 * G) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = t + u * G;
                const float x = p[(size_t)min(tt, n_tiles - 1) * NV + c];      // unconditional load (clamped address):
                v[u] = tt < n_tiles ? x : 0.f;                                 // a branch here would serialise the loads
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        scratch[g * NV + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < G; ++k) s += scratch[k * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// Exp of a twist [tau, phi] (left-multiplied onto T in place): T <- Exp(xi) * T, all in fp64.
__device__ void se3_retract_left(const double xi[6], float* T16) {
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double A, B, C;
    if (th2 < 1e-12) {
        A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0;
    } else if (th2 < 1e-4) {
        // 3-term series: exact to < 1e-12 for theta < 0.01 and immune to the cancellation in 1-cos, th-sin
        A = 1.0 - th2 / 6.0 * (1.0 - th2 / 20.0);
        B = 0.5 - th2 / 24.0 * (1.0 - th2 / 30.0);
        C = 1.0 / 6.0 - th2 / 120.0 * (1.0 - th2 / 42.0);
    } else {
        const double th = sqrt(th2);
        const double sn = sin(th), cs = cos(th);
        A = sn / th; B = (1.0 - cs) / th2; C = (th - sn) / (th2 * th);
    }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    double E[9], V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        E[i] = I + A * W[i] + B * W2[i];
        V[i] = I + B * W[i] + C * W2[i];
    }
    double dt[3];
    for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * xi[0] + V[3 * i + 1] * xi[1] + V[3 * i + 2] * xi[2];
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[3 * i + j] = T16[4 * i + j];
        t[i] = T16[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            T16[4 * i + j] = (float)(E[3 * i] * R[j] + E[3 * i + 1] * R[3 + j] + E[3 * i + 2] * R[6 + j]);
        T16[4 * i + 3] = (float)(E[3 * i] * t[0] + E[3 * i + 1] * t[1] + E[3 * i + 2] * t[2] + dt[i]);
    }
    T16[12] = 0.f; T16[13] = 0.f; T16[14] = 0.f; T16[15] = 1.f;
}

// torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad) on one fp32 scalar, in torch's operation
// order: exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2); denom = sqrt(v)/sqrt(bc2) + eps;
// p.addcdiv_(m, denom, value=-lr/bc1).  The bias corrections are Python doubles there, cast to fp32 at the op:
// neg_step = (float)(-lr / bc1), bc2s = (float)sqrt(bc2).  Returns the step to ADD.
__device__ __forceinline__ float adam_torch(float g, float& m, float& v, float neg_step, float bc2s) {
    m = m + 0.1f * (g - m);
    v = v * 0.999f;
    v = v + 0.001f * g * g;
    const float denom = sqrtf(v) / bc2s + 1e-8f;
    return neg_step * (m / denom);
}

struct AdamArgs {
    int max_N;
    float lr_kld, lr_pose, lr_aff;
    float* state;
    float* losses;
};

// One workgroup: tile-partial reduction + Adam update of pair `pi` (see include/sp_hip.h sp_pairs_adam_step).
__device__ __forceinline__ void solve_adam(const SpPair* __restrict__ pairs, int pi, const float* __restrict__ partials,
                                           const float* __restrict__ seg_partials, const AdamArgs& h) {
    const int max_N = h.max_N;
    const float lr_kld = h.lr_kld, lr_pose = h.lr_pose, lr_aff = h.lr_aff;
    float* __restrict__ state = h.state;
    float* __restrict__ losses = h.losses;
    constexpr int NV = SP_GRAD_PARTIAL_FLOATS;
    __shared__ double sums[NV];
    __shared__ double scratch[(SP_BLOCK / NV) * NV];
    const SpPair& pr = pairs[pi];
    const float* p = partials + (size_t)pr.tile0 * NV;
    reduce_columns<NV>(p, pr.n_tiles, sums, scratch);      // span records (column 13 is unused: d/dkld is per segment)
    const double scale = 1.0 / (3.0 * (double)pr.P);
    const float residual = (float)(sums[0] * scale);
    // loss = |residual| (two_frame_sfm.py:201): d loss / d residual = sign(residual)
    const double up = residual > 0.f ? scale : (residual < 0.f ? -scale : 0.0);
    float* st = state + (size_t)pi * (2 + 2 * (max_N + 8));
    float* m_kld = st + 2; float* v_kld = m_kld + max_N;
    float* m_xi = v_kld + max_N; float* v_xi = m_xi + 6;
    float* m_af = v_xi + 6; float* v_af = m_af + 2;
    const float step = st[0] + 1.f;
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const float bc2s = (float)sqrt(1.0 - pow(0.999, (double)step));
    const float ns_kld = (float)(-(double)lr_kld / bc1), ns_pose = (float)(-(double)lr_pose / bc1), ns_aff = (float)(-(double)lr_aff / bc1);
    __syncthreads();            // every thread has read st[0] before thread 0 stores the new step count below
    // per-segment log-depths
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
        double s = 0.0;
        const float* sp = seg_partials + (size_t)pr.rec0 * SP_GRAD_SEG_FLOATS;
        const int t0 = pr.seg_tile_off[n], t1 = pr.seg_tile_off[n + 1];
        for (int t = t0; t < t1; t += 8) {       // eight records in flight per trip, added in record order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)min(t + u, t1 - 1) * SP_GRAD_SEG_FLOATS];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (t + u < t1) ? (double)v[u] : 0.0;
        }
        const float g = (float)(s * up);
        pr.kld[n] += adam_torch(g, m_kld[n], v_kld[n], ns_kld, bc2s);
    }
    if (threadIdx.x == 0) {
        // gradient wrt the left tangent at identity: d/dtau = g_t ; d/dphi = vee(A - A^T), A = R g_R^T + t g_t^T
        double gR[9], gt[3], R[9], t[3];
        for (int i = 0; i < 3; ++i) {
            gt[i] = sums[1 + i] * up;
            for (int j = 0; j < 3; ++j) { gR[3 * i + j] = sums[4 + 3 * i + j] * up; R[3 * i + j] = pr.pose[4 * i + j]; }
            t[i] = pr.pose[4 * i + 3];
        }
        double A[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                A[3 * i + j] = R[3 * i] * gR[3 * j] + R[3 * i + 1] * gR[3 * j + 1] + R[3 * i + 2] * gR[3 * j + 2] + t[i] * gt[j];
        const float g6[6] = {(float)gt[0], (float)gt[1], (float)gt[2],
                             (float)(A[5] - A[7]), (float)(A[6] - A[2]), (float)(A[1] - A[3])};
        double xi[6];
        for (int i = 0; i < 6; ++i) xi[i] = adam_torch(g6[i], m_xi[i], v_xi[i], ns_pose, bc2s);
        se3_retract_left(xi, pr.pose);
        if (pr.aff) {
            const float ga = (float)(sums[14] * up), gb = (float)(sums[15] * up);
            pr.aff[2] += adam_torch(ga, m_af[0], v_af[0], ns_aff, bc2s);
            pr.aff[3] += adam_torch(gb, m_af[1], v_af[1], ns_aff, bc2s);
        }
        st[0] = step;
        losses[pi] = fabsf(residual);
    }
}

// ---------------------------------------------------------------------------------------------------
// Gauss-Newton / LM
// ---------------------------------------------------------------------------------------------------
// lm_state per pair (8 floats): [0] lambda  [1] cost of the last accepted point (<0: none yet)
//                               [2] #accepted  [3] #rejected  [4] 1 if the previous call rejected  [5] last cost seen

// Solve the symmetric positive definite 6x6 system S x = rhs by LDL^T (no square roots, six reciprocals);
// x is returned in rhs.  false if a pivot is not positive.
__device__ bool ldlt6_solve(double S[36], double rhs[6]) {
    double dinv[6];
    for (int j = 0; j < 6; ++j) {
        double d = S[7 * j];
        for (int k = 0; k < j; ++k) d -= S[6 * j + k] * S[6 * j + k] * S[7 * k];
        if (!(d > 0.0)) return false;
        S[7 * j] = d;
        dinv[j] = 1.0 / d;
        for (int i = j + 1; i < 6; ++i) {
            double v = S[6 * i + j];
            for (int k = 0; k < j; ++k) v -= S[6 * i + k] * S[6 * j + k] * S[7 * k];
            S[6 * i + j] = v * dinv[j];
        }
    }
    for (int i = 0; i < 6; ++i) {            // L y = rhs
        double v = rhs[i];
        for (int k = 0; k < i; ++k) v -= S[6 * i + k] * rhs[k];
        rhs[i] = v;
    }
    for (int i = 0; i < 6; ++i) rhs[i] *= dinv[i];
    for (int i = 5; i >= 0; --i) {           // L^T x = y
        double v = rhs[i];
        for (int k = i + 1; k < 6; ++k) v -= S[6 * k + i] * rhs[k];
        rhs[i] = v;
    }
    return true;
}

#define SP_SEG_CACHE 256     // segments whose reduced {h(6), 1/D', b_d} live in LDS; beyond that they are recomputed

// {h_pd(6), D, b_d} of segment n summed over its (chunk, wave) records; lam -> {h, 1/(D(1+lam)) or 0, b_d}
__device__ __forceinline__ void segment_system(const float* __restrict__ p, const int32_t* __restrict__ seg_tile_off, int n,
                                               double lam, double (&o)[8], bool pose_only = false) {
    constexpr int NV = SP_GN_SEG_FLOATS;
    if (pose_only) {             // the segment's depth is not an unknown of this step: nothing to eliminate, nothing to update
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.0;
        return;
    }
    double h[6] = {0, 0, 0, 0, 0, 0}, D = 0.0, bd = 0.0;
    const int t0 = seg_tile_off[n], t1 = seg_tile_off[n + 1];
    for (int t = t0; t < t1; t += 8) {          // eight records in flight per trip, summed in record order
        float v[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4* q = reinterpret_cast<const float4*>(p + (size_t)min(t + u, t1 - 1) * NV);     // 32-byte records
            const float4 a = q[0], b = q[1];
            v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w;
            v[u][4] = b.x; v[u][5] = b.y; v[u][6] = b.z; v[u][7] = b.w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t + u < t1) {
#pragma unroll
                for (int i = 0; i < 6; ++i) h[i] += (double)v[u][i];
                D += (double)v[u][6];
                bd += (double)v[u][7];
            }
        }
    }
    const double Dd = D * (1.0 + lam);
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = h[i];
    o[6] = Dd > 1e-12 ? 1.0 / Dd : 0.0;
    o[7] = bd;
}

struct GnArgs {
    int max_N;
    float lm_up, lm_down, lm_min;
    float* lm_state;
    float* backup;
    float* costs;
    float conv_tol;          // > 0: a pair whose last accepted step lowered its cost by less than conv_tol * cost is marked done
    int32_t* done;           // [n_pairs] or NULL; done pairs are skipped by the cost pass and by this solver
    int32_t* phase;          // per-pair schedules (sp_pairs_schedule_*): [n_pairs] current phase, advanced here instead of `done`
    int32_t* iters;          // ... iterations spent in the current phase
    int max_iters;           // ... of at most this many
    int pose_only;           // SP_PHASE_POSE_ONLY: no Schur complement, no depth update
    int next_phase;          // ... the phase that follows the current one (SpPhase.next)
    float depth_damp;        // ... extra LM damping of the log-depth block (SP_PHASE_DEPTH_DAMP): D (1 + lambda + depth_damp)
    float adam_lr_pose, adam_lr_kld;   // SP_PHASE_ADAM phases (solve_adam_sched): the rates and the moments (SpSchedule.adam_state)
    float* adam_state;
    int predicted_exit;      // SP_PHASE_PREDICTED_EXIT: leave the phase when the step just taken is PREDICTED to buy less than conv_tol of the cost
    uint32_t idle_mask;      // phases whose pairs sit this round out: their work list was not launched (schedule_cost_from)
};

// a pair leaves its current phase (thread 0): the next one starts afresh; lm_state[7] records how this one ended
// (+iterations = on its cap, -iterations = by its convergence test)
__device__ __forceinline__ void leave_phase(const GnArgs& h, int pi, float* __restrict__ ls, int spent, bool on_cap) {
    h.phase[pi] = h.next_phase;
    h.iters[pi] = 0;
    ls[1] = -1.f;
    ls[7] = on_cap ? (float)spent : -(float)spent;
}

// One workgroup: tile-partial reduction + Schur-complement LM step of pair `pi` (include/sp_hip.h sp_pairs_gn_step).
__device__ __forceinline__ void solve_gn(const SpPair* __restrict__ pairs, int pi, const float* __restrict__ partials,
                                         const float* __restrict__ seg_partials, const GnArgs& h) {
    const int max_N = h.max_N;
    const float lm_up = h.lm_up, lm_down = h.lm_down, lm_min = h.lm_min;
    float* __restrict__ lm_state = h.lm_state;
    float* __restrict__ backup = h.backup;
    float* __restrict__ costs = h.costs;
    constexpr int NV = SP_GN_PARTIAL_FLOATS;
    __shared__ double sums[NV];      // [0] cost, [1..21] Hpp upper, [22..27] b_p, [28] valid points
    __shared__ double scratch[(SP_BLOCK / NV) * NV];
    __shared__ double seg[SP_SEG_CACHE][8];
    __shared__ double schur_part[8][27];
    __shared__ double schur[27];     // 21 + 6
    __shared__ double dxi[6];
    __shared__ double gain_part[SP_WAVES];
    __shared__ int decision;         // 0 = step, 1 = rejected (restore)
    const SpPair& pr = pairs[pi];
    if (h.done && h.done[pi]) return;          // converged earlier (its spans were not evaluated either)
    const float* p = partials + (size_t)pr.tile0 * NV;
    const float* sp = seg_partials + (size_t)pr.rec0 * SP_GN_SEG_FLOATS;
    float* ls = lm_state + (size_t)pi * SP_LM_STRIDE;
    float* bk = backup + (size_t)pi * (16 + max_N);
    reduce_columns<NV>(p, pr.n_tiles, sums, scratch);
    const double cost = sums[0] / (3.0 * (double)pr.P);
    if (threadIdx.x == 0) {
        const float last = ls[1];
        int rej = 0;
        if (last >= 0.f && (float)cost > last * (1.f + 1e-6f) && ls[4] == 0.f) rej = 1;
        // convergence (per pair, on the device): the step that led here was accepted and bought less than conv_tol of the
        // cost -> keep the current point and stop working on this pair (decision 2)
        if (!rej && (h.done || h.phase) && h.conv_tol > 0.f && last >= 0.f && ls[4] == 0.f && (last - (float)cost) <= h.conv_tol * last) {
            rej = 2;
            if (h.done) h.done[pi] = 1;
            if (h.phase) { leave_phase(h, pi, ls, h.iters[pi], false); ls[4] = 0.f; }
        }
        decision = rej;
        ls[5] = (float)cost;
        ls[6] = (float)(sums[28] / (double)pr.P);
        costs[pi] = (float)cost;
    }
    __syncthreads();
    if (decision == 2) return;
    if (decision == 1) {
        // undo the previous step; the next cost pass re-evaluates at the restored point with a larger lambda
        for (int i = threadIdx.x; i < 16; i += SP_BLOCK) pr.pose[i] = bk[i];
        for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) pr.kld[n] = bk[16 + n];
        if (threadIdx.x == 0) {
            ls[0] *= lm_up; ls[3] += 1.f; ls[4] = 1.f;
            if (h.phase) {                          // a rejected iteration counts towards the phase's budget like an accepted one
                const int n = h.iters[pi] + 1;
                if (n >= h.max_iters) { leave_phase(h, pi, ls, n, true); ls[4] = 0.f; }
                else h.iters[pi] = n;
            }
        }
        return;
    }
    float lambda = ls[0];
    if (ls[4] == 0.f) lambda = fmaxf(lambda * lm_down, lm_min);
    const double lam = (double)lambda;
    // back up the point we are about to leave
    for (int i = threadIdx.x; i < 16; i += SP_BLOCK) bk[i] = pr.pose[i];
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) bk[16 + n] = pr.kld[n];
    // per-segment reduced systems into LDS
    const int n_cached = min(pr.N, SP_SEG_CACHE);
    for (int n = threadIdx.x; n < n_cached; n += SP_BLOCK) {
        double o[8];
        segment_system(sp, pr.seg_tile_off, n, lam + (double)h.depth_damp, o, h.pose_only != 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) seg[n][i] = o[i];
    }
    __syncthreads();
    // Schur complement of the diagonal depth block: 27 sums over the segments, thread (k, j) takes value k over
    // segments j, j+8, ... ; the 8 partials per value are combined in fixed order
    {
        const int k = threadIdx.x >> 3, j = threadIdx.x & 7;
        if (k < 27) {
            // upper-triangle index k -> (a, b); k >= 21 -> (k - 21, "b_d")
            int a = 0, b = 0;
            if (k < 21) { int rem = k, row = 0; while (rem >= 6 - row) { rem -= 6 - row; ++row; } a = row; b = row + rem; }
            else a = k - 21;
            double acc = 0.0;
            for (int n = j; n < pr.N; n += 8) {
                double o[8];
                if (n < SP_SEG_CACHE) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = seg[n][i];
                } else segment_system(sp, pr.seg_tile_off, n, lam + (double)h.depth_damp, o, h.pose_only != 0);
                acc += o[a] * (k < 21 ? o[b] : o[7]) * o[6];
            }
            schur_part[j][k] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += schur_part[j][threadIdx.x];
        schur[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S[36], rhs[6];
        int k = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) {
                double v = sums[1 + k] - schur[k];
                if (i == j) v += lam * sums[1 + k] + 1e-12;
                S[6 * i + j] = v; S[6 * j + i] = v;
                ++k;
            }
        for (int i = 0; i < 6; ++i) rhs[i] = -(sums[22 + i] - schur[21 + i]);
        const bool ok = ldlt6_solve(S, rhs);
        for (int i = 0; i < 6; ++i) dxi[i] = ok ? rhs[i] : 0.0;
        ls[0] = lambda; ls[1] = (float)cost; ls[2] += 1.f; ls[4] = 0.f;
    }
    __syncthreads();
    // the first-order change of sum |r| along the step, b . delta (b = sum sign(r) J where |r| > eps): what the step is PREDICTED to buy
    double gain = 0.0;
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
        double o[8];
        if (n < SP_SEG_CACHE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = seg[n][i];
        } else segment_system(sp, pr.seg_tile_off, n, lam + (double)h.depth_damp, o, h.pose_only != 0);
        if (o[6] > 0.0) {
            double r = -o[7];
#pragma unroll
            for (int i = 0; i < 6; ++i) r -= o[i] * dxi[i];
            double dd = r * o[6];
            dd = fmin(fmax(dd, -0.5), 0.5);   // trust region on one log-depth step (factor e^0.5 in depth)
            pr.kld[n] += (float)dd;
            gain -= o[7] * dd;
        }
    }
    if (threadIdx.x == 64) {                  // a different wave than the one that just solved: no reason, just parallel
        double xi[6];
        for (int i = 0; i < 6; ++i) xi[i] = dxi[i];
        se3_retract_left(xi, pr.pose);
    }
    if (h.phase) {
        // iteration budget of the phase (the step computed here has been applied), and -- SP_PHASE_PREDICTED_EXIT -- its PREDICTED end: the
        // convergence test above needs one more evaluation to SEE that the last step bought less than conv_tol of the cost, a quarter
        // of a frame pair's work when that evaluation is the all-points polish; b . delta says so beforehand (the IRLS quadratic
        // majorises sum |r|: the linearised cost falls by between half of it and all of it).  Not under a heavy damping, where a short
        // step is the damping's doing: lambda <= 1e-2 and no SP_PHASE_DEPTH_DAMP.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gain += __shfl_xor(gain, o, 64);
        if ((threadIdx.x & 63) == 0) gain_part[threadIdx.x >> 6] = gain;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = h.iters[pi] + 1;
            bool done = false;
            if (h.predicted_exit && h.conv_tol > 0.f && h.depth_damp == 0.f && lambda <= 1e-2f) {
                double g = gain_part[0] + gain_part[1] + gain_part[2] + gain_part[3];
                for (int i = 0; i < 6; ++i) g -= sums[22 + i] * dxi[i];
                done = g / (3.0 * (double)pr.P) <= (double)h.conv_tol * cost;
            }
            if (done) leave_phase(h, pi, ls, n, false);
            else if (n >= h.max_iters) leave_phase(h, pi, ls, n, true);
            else h.iters[pi] = n;
        }
    }
}

// One workgroup: an SP_PHASE_ADAM iteration of pair / slot `pi` (include/sp_hip.h): one torch.optim.Adam step on {left pose tangent,
// log-depths} from the sums of the phase's Gauss-Newton cost pass -- b_p and b_d of the IRLS normal equations are the gradient of
// sum |r| (exactly sum sign(r) J where |r| > irls_eps), and the reference's loss is that sum over 3 P (odometery/two_frame_sfm.py:201).
// Rates 1e-2 / 1e-3 as :116-123; no step is rejected; the phase ends on its cap.  Adam arithmetic: adam_torch above, like solve_adam.
__device__ __forceinline__ void solve_adam_sched(const SpPair* __restrict__ pairs, int pi, const float* __restrict__ partials,
                                                 const float* __restrict__ seg_partials, const GnArgs& h) {
    constexpr int NV = SP_GN_PARTIAL_FLOATS;
    __shared__ double sums[NV];
    __shared__ double scratch[(SP_BLOCK / NV) * NV];
    const SpPair& pr = pairs[pi];
    const int max_N = h.max_N;
    const float* p = partials + (size_t)pr.tile0 * NV;
    const float* sp = seg_partials + (size_t)pr.rec0 * SP_GN_SEG_FLOATS;
    float* ls = h.lm_state + (size_t)pi * SP_LM_STRIDE;
    reduce_columns<NV>(p, pr.n_tiles, sums, scratch);
    const double scale = 1.0 / (3.0 * (double)pr.P);
    float* st = h.adam_state + (size_t)pi * (2 + 2 * (max_N + 8));
    float* m_kld = st + 2; float* v_kld = m_kld + max_N;
    float* m_xi = v_kld + max_N; float* v_xi = m_xi + 6;
    const float step = st[0] + 1.f;
    const double bc1 = 1.0 - pow(0.9, (double)step);
    const float bc2s = (float)sqrt(1.0 - pow(0.999, (double)step));
    const float ns_kld = (float)(-(double)h.adam_lr_kld / bc1), ns_pose = (float)(-(double)h.adam_lr_pose / bc1);
    __syncthreads();            // every thread has read st[0] before thread 0 stores the new step count below
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
        double o[8];
        segment_system(sp, pr.seg_tile_off, n, 0.0, o);
        const float g = (float)(o[7] * scale);
        pr.kld[n] += adam_torch(g, m_kld[n], v_kld[n], ns_kld, bc2s);
    }
    if (threadIdx.x == 0) {
        double xi[6];
        for (int i = 0; i < 6; ++i) xi[i] = adam_torch((float)(sums[22 + i] * scale), m_xi[i], v_xi[i], ns_pose, bc2s);
        se3_retract_left(xi, pr.pose);
        st[0] = step;
        const float cost = (float)(sums[0] * scale);
        ls[1] = cost; ls[2] += 1.f; ls[4] = 0.f; ls[5] = cost;
        ls[6] = (float)(sums[28] / (double)pr.P);
        h.costs[pi] = cost;
        const int n = h.iters[pi] + 1;
        if (n >= h.max_iters) leave_phase(h, pi, ls, n, true);
        else h.iters[pi] = n;
    }
}

// ---------------------------------------------------------------------------------------------------
// lie/lie_algebra.py:41-119 renormalise_se3: R -> best-conditioned quaternion -> R, in place (fp32, the reference's
// operation order).  M: row-major 4x4 (or the first 12 floats of one).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void renormalise_rotation(float* __restrict__ M) {
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    float q0 = 1.f + m00 + m11 + m22, q1 = 1.f + m00 - m11 - m22, q2 = 1.f - m00 + m11 - m22, q3 = 1.f - m00 - m11 + m22;
    q0 = q0 > 0.f ? sqrtf(q0) : 0.f; q1 = q1 > 0.f ? sqrtf(q1) : 0.f;
    q2 = q2 > 0.f ? sqrtf(q2) : 0.f; q3 = q3 > 0.f ? sqrtf(q3) : 0.f;
    // argmax, first maximum like torch; the candidate row is picked with selects (a dynamically indexed local array would
    // live in scratch memory, and this runs on the critical path of the one-workgroup optimiser kernel)
    int best = 0; float qb = q0;
    if (q1 > qb) { best = 1; qb = q1; }
    if (q2 > qb) { best = 2; qb = q2; }
    if (q3 > qb) { best = 3; qb = q3; }
    const float a = m21 - m12, b = m02 - m20, c = m10 - m01, d = m10 + m01, e = m02 + m20, f = m12 + m21;
    const float c0 = best == 0 ? q0 * q0 : (best == 1 ? a : (best == 2 ? b : c));
    const float c1 = best == 0 ? a : (best == 1 ? q1 * q1 : (best == 2 ? d : e));
    const float c2 = best == 0 ? b : (best == 1 ? d : (best == 2 ? q2 * q2 : f));
    const float c3 = best == 0 ? c : (best == 1 ? e : (best == 2 ? f : q3 * q3));
    const float den = 2.f * fmaxf(qb, 0.1f);
    const float r = c0 / den, x = c1 / den, y = c2 / den, z = c3 / den;
    const float s = 2.f / (r * r + x * x + y * y + z * z);
    M[0] = 1.f - s * (y * y + z * z); M[1] = s * (x * y - z * r);       M[2] = s * (x * z + y * r);
    M[4] = s * (x * y + z * r);       M[5] = 1.f - s * (x * x + z * z); M[6] = s * (y * z - x * r);
    M[8] = s * (x * z - y * r);       M[9] = s * (y * z + x * r);       M[10] = 1.f - s * (x * x + y * y);
}

// ---------------------------------------------------------------------------------------------------
// Forward-mode dual numbers (ND tangents at once) and T = Exp(a) * X on them: value and derivative of the pose
// parameter of the reference's drivers come from the very same code (sp_se3_retract, sp_window_step).
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
        if (t < 0.5) {
            // Taylor series in theta^2, Horner form, 10 terms: |error| < 1e-17 for theta^2 < 0.5, and no sqrt / sin / cos on
            // the critical path of the one-workgroup optimiser kernels (pose tangents are small)
            //   A = sum (-t)^k / (2k+1)!    B = sum (-t)^k / (2k+2)!    C = sum (-t)^k / (2k+3)!
            // coefficients 1/(2k+1)!, 1/(2k+2)!, 1/(2k+3)! (folded to constants by the compiler)
            double ia[10], ib[10], ic[10];
            {
                double f = 1.0;
                for (int k = 0; k < 10; ++k) {
                    const double n1 = 2.0 * k + 1.0;
                    if (k > 0) f *= (2.0 * k) * n1;            // f = (2k+1)!
                    ia[k] = 1.0 / f; ib[k] = ia[k] / (n1 + 1.0); ic[k] = ib[k] / (n1 + 2.0);
                }
            }
            double va = 0.0, vb = 0.0, vc = 0.0, wa = 0.0, wb = 0.0, wc = 0.0;    // v = value, w = derivative (Horner on both)
            for (int k = 9; k >= 0; --k) {
                const double sg = (k & 1) ? -1.0 : 1.0;
                wa = wa * t + va; va = va * t + sg * ia[k];
                wb = wb * t + vb; vb = vb * t + sg * ib[k];
                wc = wc * t + vc; vc = vc * t + sg * ic[k];
            }
            a_ = va; b_ = vb; c_ = vc; da_ = wa; db_ = wb; dc_ = wc;
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
