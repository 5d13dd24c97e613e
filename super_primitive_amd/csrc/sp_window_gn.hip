// Gauss-Newton / Levenberg-Marquardt flavour of the window optimiser (BASELINE.json north_star: "Gauss-Newton/LM
// solve on SE(3) (+) log-depth").  Same window graph as sp_window.hip -- NODES = poses (+ affine brightness pair), BLOCKS = per-keyframe
// log-depth vectors, EDGES = (source keyframe -> target frame) photometric terms, each edge one SpPair of the many-pairs cost path --
// and the same loop semantics as the reference's drivers:
//
//   tracking           odometery/odometery.py:375-407    6 pose + 2 affine unknowns of the tracked frame, keyframe depths fixed
//   windowed mapping   odometery/odometery.py:756-915    K poses + sum N log-depths + affines; relative pose
//                      D_trg inv(T_trg) T_src inv(D_src) (:793,817); first keyframe fixed (:589-592), oldest depths frozen when the
//                      window is full (:594-603); every pose folded in T <- T inv(Exp(D)) and renormalised after the step (:861-882)
//
// but the step is a damped Newton step on the IRLS-weighted normal equations instead of Adam: ~10 iterations where Adam takes 300-500.
// One iteration = sp_pairs_cost(mode 2) over all edges  ->  sp_window_gn_step = two launches:
//
//   k_window_gn_reduce (grid = edges)   fixed-order fp64 reduction of an edge's span / segment records into its local system over
//                                       z_e = [xi_e (left tangent of the edge's relative pose), a_e, b_e] and its source keyframe's depths
//   k_window_gn_update (one workgroup)  LM accept / undo against the previous point; chain rule z_e = G_t y_t + G_s y_s onto the node
//                                       unknowns y = [d (6), a, b] (G_t = I; G_s = -blockdiag(Ad_P, I_2) since P = Exp(d_t) M Exp(-d_s) =
//                                       Exp(d_t - Ad_M d_s) M); assembly of the reduced camera system in LDS (<= 128 unknowns, fp64);
//                                       Schur complement of the per-segment depth unknowns; Cholesky; back-substitution; fold-in,
//                                       renormalisation; relative poses and affine slots of every edge for the next cost pass.
// No host synchronisation; the loss history and the converged flag live on the device like in sp_window_step.
#include "sp_solve_device.h"

namespace {

#define SP_WGN_REC 46            // doubles per edge record: cost, valid points, H_z upper triangle (36), b_z (8); then 10 per segment
#define SP_WGN_SEG 10            // c (8) = [h_pd (6), h_ad, h_bd], D, b_d
#define SP_WGN_MAX_Y 128         // unknowns of the reduced camera system (6 per free pose + 2 per free affine pair)
#define SP_WGN_MAX_NODES 64
#define SP_WGN_STATE 16

__device__ __forceinline__ int tri8(int i, int j) { return i * 8 - i * (i - 1) / 2 + (j - i); }      // (i <= j) in the upper triangle of an 8x8

__global__ __launch_bounds__(SP_BLOCK) void k_window_gn_reduce(const SpPair* __restrict__ pairs, const SpWindowEdge* __restrict__ edges,
                                                               const float* __restrict__ partials, const float* __restrict__ seg_partials,
                                                               double* __restrict__ scratch, int stride) {
    constexpr int NV = SP_GNA_PARTIAL_FLOATS, NS = SP_GNA_SEG_FLOATS;
    __shared__ double sums[NV];
    __shared__ double red[(SP_BLOCK / NV) * NV];
    const int e = blockIdx.x;
    const SpPair& pr = pairs[e];
    reduce_columns<NV>(partials + (size_t)pr.tile0 * NV, pr.n_tiles, sums, red);
    const double inv3P = 1.0 / (3.0 * (double)pr.P);
    const double scale = (double)edges[e].weight * inv3P;         // loss = sum_e w_e mean|r_e|
    double* rec = scratch + (size_t)e * stride;
    if (threadIdx.x == 0) {
        rec[0] = sums[0] * inv3P;
        rec[1] = sums[28];
    }
    if (threadIdx.x < 36) {
        // upper triangle of H_z, z = [xi(6), a, b]
        int k = threadIdx.x, i = 0;
        while (k >= 8 - i) { k -= 8 - i; ++i; }
        const int j = i + k;
        double v;
        if (j < 6) {                               // pose-pose: sums[1..21] is the upper triangle of the 6x6
            v = sums[1 + (i * 6 - i * (i - 1) / 2 + (j - i))];
        } else if (i < 6) v = sums[(j == 6 ? 34 : 40) + i];
        else v = sums[29 + (i - 6) + (j - 6)];       // (6,6) -> 29, (6,7) -> 30, (7,7) -> 31
        rec[2 + threadIdx.x] = v * scale;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 72) {
        const int i = threadIdx.x - 64;
        rec[38 + i] = (i < 6 ? sums[22 + i] : sums[32 + (i - 6)]) * scale;
    }
    const float* sp = seg_partials + (size_t)pr.rec0 * NS;
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, D = 0.0, bd = 0.0;
        const int t0 = pr.seg_tile_off[n], t1 = pr.seg_tile_off[n + 1];
        for (int t = t0; t < t1; ++t) {              // summed in record order (fixed)
            const float4* q = reinterpret_cast<const float4*>(sp + (size_t)t * NS);
            const float4 a = q[0], b = q[1], cc = q[2];
            c[0] += (double)a.x; c[1] += (double)a.y; c[2] += (double)a.z; c[3] += (double)a.w;
            c[4] += (double)b.x; c[5] += (double)b.y; D += (double)b.z; bd += (double)b.w;
            c[6] += (double)cc.x; c[7] += (double)cc.y;
        }
        double* o = rec + SP_WGN_REC + (size_t)n * SP_WGN_SEG;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = c[i] * scale;
        o[8] = D * scale;
        o[9] = bd * scale;
    }
}

struct WGnArgs {
    const SpPair* pairs; const SpWindowEdge* edges; int n_edges;
    SpWindowNode* nodes; int n_nodes;
    const SpWindowBlock* blocks; int n_blocks; int max_N;
    double* scratch; int stride;          // edge records
    double* Ad;                           // n_edges x 36: Ad of every edge's relative pose
    double* C;                            // sum N x SP_WGN_MAX_Y: coupling of every depth unknown with the camera unknowns
    double* Dinv;                         // sum N: 1 / (D (1 + lambda)) or 0
    double* Bd;                           // sum N
    SpWindowNode* nodes_backup; float* kld_backup;
    int flags; float lm_up, lm_down, lm_min, conv_tol;
    float* state; float* losses; int max_losses;
};

// Exp(xi) as a 3x4 double matrix (same closed form as sp_window.hip)
__device__ void wgn_se3_exp(const double xi[6], double E[12]) {
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double A, B, C;
    if (th2 < 1e-4) {
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
    for (int i = 0; i < 3; ++i) {
        double Vrow[3];
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            E[4 * i + j] = I + A * W[3 * i + j] + B * W2[3 * i + j];
            Vrow[j] = I + B * W[3 * i + j] + C * W2[3 * i + j];
        }
        E[4 * i + 3] = Vrow[0] * xi[0] + Vrow[1] * xi[1] + Vrow[2] * xi[2];
    }
}

// relative pose and affine slot of edge e from the current nodes (the compose step of sp_window.hip, unstaged)
__device__ void wgn_compose_edge(const WGnArgs& w, int e) {
    const SpWindowEdge ed = w.edges[e];
    float* P = w.pairs[e].pose;
    float* af = w.pairs[e].aff;
    const SpWindowNode& nt = w.nodes[ed.trg_node];
    if (nt.kind == 1) {
        Dual<1> ad[6], out[12];
        for (int q = 0; q < 6; ++q) { ad[q].v = nt.a[q]; ad[q].d[0] = 0.f; }
        se3_exp_times<1>(ad, nt.T, out);
        for (int q = 0; q < 12; ++q) P[q] = out[q].v;
    } else {
        double Rs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ts[3] = {0, 0, 0};
        if (ed.src_node >= 0) {
            const SpWindowNode& ns = w.nodes[ed.src_node];
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rs[3 * r + c] = ns.T[4 * r + c]; ts[r] = ns.T[4 * r + 3]; }
        }
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                P[4 * r + c] = (float)((double)nt.T[r] * Rs[c] + (double)nt.T[4 + r] * Rs[3 + c] + (double)nt.T[8 + r] * Rs[6 + c]);
            P[4 * r + 3] = (float)((double)nt.T[r] * (ts[0] - nt.T[3]) + (double)nt.T[4 + r] * (ts[1] - nt.T[7]) +
                                   (double)nt.T[8 + r] * (ts[2] - nt.T[11]));
        }
    }
    P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
    if (af) {
        af[0] = ed.src_node >= 0 ? w.nodes[ed.src_node].aff[0] : 0.f;
        af[1] = ed.src_node >= 0 ? w.nodes[ed.src_node].aff[1] : 0.f;
        af[2] = nt.aff[0];
        af[3] = nt.aff[1];
    }
}

// column `col` (0..15) of the 8 x 16 map z_e = G [y_t ; y_s]: y_t = columns 0..7 (identity), y_s = columns 8..15 (-Ad, -I_2)
__device__ __forceinline__ void wgn_gcol(const double* __restrict__ Ad, int col, double (&g)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) g[p] = 0.0;
    if (col < 8) g[col] = 1.0;
    else if (col < 14) { for (int p = 0; p < 6; ++p) g[p] = -Ad[6 * p + (col - 8)]; }
    else g[col - 8] = -1.0;
}

__global__ __launch_bounds__(SP_BLOCK) void k_window_gn_update(WGnArgs w) {
    __shared__ double H[SP_WGN_MAX_Y * SP_WGN_MAX_Y];
    __shared__ double g[SP_WGN_MAX_Y], dy[SP_WGN_MAX_Y];
    __shared__ int pose_off[SP_WGN_MAX_NODES], aff_off[SP_WGN_MAX_NODES];
    __shared__ int blk_off[SP_WGN_MAX_NODES + 1];
    __shared__ int n_y_s, decision, chol_fail;
    __shared__ double lam_s;
    const int tid = threadIdx.x;
    float* st = w.state;
    const bool pose_only = (w.flags & 1) != 0;
    if (tid == 0) {
        int ny = 0;
        for (int i = 0; i < w.n_nodes; ++i) {
            const SpWindowNode& nd = w.nodes[i];
            pose_off[i] = nd.lr_pose > 0.f ? ny : -1;
            if (nd.lr_pose > 0.f) ny += 6;
            aff_off[i] = nd.lr_aff > 0.f ? ny : -1;
            if (nd.lr_aff > 0.f) ny += 2;
        }
        n_y_s = ny;
        int off = 0;
        for (int b = 0; b < w.n_blocks; ++b) { blk_off[b] = off; off += w.blocks[b].N; }
        blk_off[w.n_blocks] = off;
        // ---- loss, LM decision ----------------------------------------------------------------------------------
        double loss = 0.0;
        for (int e = 0; e < w.n_edges; ++e) loss += (double)w.edges[e].weight * w.scratch[(size_t)e * w.stride];
        int dec = 0;                      // 0 = step, 1 = reject (restore), 2 = converged / frozen
        if (ny > SP_WGN_MAX_Y || ny == 0) { st[9] = 1.f; st[6] = 1.f; }      // more camera unknowns than the LDS system holds: refuse (freeze)
        if (st[6] != 0.f) dec = 2;
        else {
            const int it = (int)st[5];
            if (it < w.max_losses) w.losses[it] = (float)loss;
            st[5] = (float)(it + 1);
            st[7] = (float)loss;
            const float last = st[1];
            if (last >= 0.f && (float)loss > last * (1.f + 1e-6f) && st[4] == 0.f) dec = 1;
            else if (last >= 0.f && st[4] == 0.f && w.conv_tol > 0.f && (last - (float)loss) <= w.conv_tol * last) { dec = 2; st[6] = 1.f; }
            if (dec == 1) { st[0] *= w.lm_up; st[3] += 1.f; st[4] = 1.f; }
            if (dec == 0) {
                float lam = st[0];
                if (st[4] == 0.f) lam = fmaxf(lam * w.lm_down, w.lm_min);
                st[0] = lam; st[1] = (float)loss; st[2] += 1.f; st[4] = 0.f;
                lam_s = (double)lam;
            }
        }
        decision = dec;
        chol_fail = 0;
    }
    __syncthreads();
    const int n_y = n_y_s;
    const int sumN = blk_off[w.n_blocks];
    if (decision == 2) return;
    if (decision == 1) {
        // undo the previous step: nodes (pose, tangent, affine) and log-depths back to the stored point
        for (int i = tid; i < w.n_nodes * 44; i += SP_BLOCK) reinterpret_cast<uint32_t*>(w.nodes)[i] = reinterpret_cast<const uint32_t*>(w.nodes_backup)[i];
        for (int b = 0; b < w.n_blocks; ++b)
            for (int n = tid; n < w.blocks[b].N; n += SP_BLOCK) w.blocks[b].kld[n] = w.kld_backup[blk_off[b] + n];
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < w.n_edges; e += SP_BLOCK) wgn_compose_edge(w, e);
        return;
    }
    const double lam = lam_s;
    // ---- back up the point we are about to leave ----------------------------------------------------------------
    for (int i = tid; i < w.n_nodes * 44; i += SP_BLOCK) reinterpret_cast<uint32_t*>(w.nodes_backup)[i] = reinterpret_cast<const uint32_t*>(w.nodes)[i];
    for (int b = 0; b < w.n_blocks; ++b)
        for (int n = tid; n < w.blocks[b].N; n += SP_BLOCK) w.kld_backup[blk_off[b] + n] = w.blocks[b].kld[n];
    // ---- Ad of every edge's relative pose; clear the system -----------------------------------------------------
    for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
        const float* P = w.pairs[e].pose;
        double R[9], t[3];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = P[4 * r + c]; t[r] = P[4 * r + 3]; }
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double* A = w.Ad + (size_t)e * 36;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                A[6 * r + c] = R[3 * r + c];                                                           // tau' = R tau + [t]x R phi
                A[6 * r + 3 + c] = tx[3 * r] * R[c] + tx[3 * r + 1] * R[3 + c] + tx[3 * r + 2] * R[6 + c];
                A[6 * (r + 3) + c] = 0.0;                                                              // phi' = R phi
                A[6 * (r + 3) + 3 + c] = R[3 * r + c];
            }
    }
    for (int i = tid; i < n_y * n_y; i += SP_BLOCK) H[i] = 0.0;
    for (int i = tid; i < n_y; i += SP_BLOCK) g[i] = 0.0;
    __syncthreads();
    // ---- assembly: thread (i, j) of the 16 x 16 local block of one edge at a time ---------------------------------
    {
        const int li = tid >> 4, lj = tid & 15;
        for (int e = 0; e < w.n_edges; ++e) {
            const SpWindowEdge ed = w.edges[e];
            const double* rec = w.scratch + (size_t)e * w.stride;
            const double* Ad = w.Ad + (size_t)e * 36;
            auto global_index = [&](int l) -> int {
                const int node = l < 8 ? ed.trg_node : ed.src_node;
                if (node < 0) return -1;
                const int k = l & 7;
                const int base = k < 6 ? pose_off[node] : aff_off[node];
                return base < 0 ? -1 : base + (k < 6 ? k : k - 6);
            };
            const int gi = global_index(li), gj = global_index(lj);
            if (gi >= 0 && gj >= 0) {
                double a[8], b[8];
                wgn_gcol(Ad, li, a);
                wgn_gcol(Ad, lj, b);
                double v = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    if (a[p] == 0.0) continue;
                    double row = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) row += rec[2 + (p <= q ? tri8(p, q) : tri8(q, p))] * b[q];
                    v += a[p] * row;
                }
                H[gi * n_y + gj] += v;
                if (lj == 0) {
                    double s = 0.0;
#pragma unroll
                    for (int p = 0; p < 8; ++p) s += a[p] * rec[38 + p];
                    g[gi] += s;
                }
            } else if (gi >= 0 && lj == 0) {
                // (column 0 of the local block is fixed -- the target pose is not optimised -- but row li still owns a rhs entry)
                double a[8];
                wgn_gcol(Ad, li, a);
                double s = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) s += a[p] * rec[38 + p];
                g[gi] += s;
            }
            __syncthreads();
        }
    }
    // ---- depth unknowns: D, b_d and the coupling rows C of every segment (one thread per row) ---------------------
    for (int b = 0; b < w.n_blocks; ++b) {
        const SpWindowBlock bk = w.blocks[b];
        const bool frozen = pose_only || !(bk.lr > 0.f);
        for (int n = tid; n < bk.N; n += SP_BLOCK) {
            const int r = blk_off[b] + n;
            double* Crow = w.C + (size_t)r * SP_WGN_MAX_Y;
            double D = 0.0, bd = 0.0;
            if (!frozen) {
                for (int i = 0; i < n_y; ++i) Crow[i] = 0.0;
                for (int e = 0; e < w.n_edges; ++e) {
                    const SpWindowEdge ed = w.edges[e];
                    if (ed.block != b) continue;
                    const double* o = w.scratch + (size_t)e * w.stride + SP_WGN_REC + (size_t)n * SP_WGN_SEG;
                    D += o[8]; bd += o[9];
                    const int pt = pose_off[ed.trg_node], at = aff_off[ed.trg_node];
                    if (pt >= 0) for (int k = 0; k < 6; ++k) Crow[pt + k] += o[k];
                    if (at >= 0) { Crow[at] += o[6]; Crow[at + 1] += o[7]; }
                    if (ed.src_node >= 0) {
                        const int ps = pose_off[ed.src_node], as = aff_off[ed.src_node];
                        if (ps >= 0) {
                            const double* Ad = w.Ad + (size_t)e * 36;
                            for (int k = 0; k < 6; ++k) {
                                double s = 0.0;
                                for (int p = 0; p < 6; ++p) s += Ad[6 * p + k] * o[p];
                                Crow[ps + k] -= s;
                            }
                        }
                        if (as >= 0) { Crow[as] -= o[6]; Crow[as + 1] -= o[7]; }
                    }
                }
            }
            const double Dd = D * (1.0 + lam);
            w.Dinv[r] = (!frozen && Dd > 1e-12) ? 1.0 / Dd : 0.0;
            w.Bd[r] = bd;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- LM damping of the camera block, then the Schur complement of the depth block ------------------------------
    for (int i = tid; i < n_y; i += SP_BLOCK) H[i * n_y + i] = H[i * n_y + i] * (1.0 + lam) + 1e-12;
    __syncthreads();
    for (int idx = tid; idx < n_y * n_y; idx += SP_BLOCK) {
        const int i = idx / n_y, j = idx - i * n_y;
        if (j > i) continue;
        double s = 0.0;
        for (int r = 0; r < sumN; ++r) {
            const double dinv = w.Dinv[r];
            if (dinv == 0.0) continue;
            const double* Crow = w.C + (size_t)r * SP_WGN_MAX_Y;
            s += Crow[i] * Crow[j] * dinv;
        }
        H[i * n_y + j] -= s;
    }
    for (int i = tid; i < n_y; i += SP_BLOCK) {
        double s = 0.0;
        for (int r = 0; r < sumN; ++r) {
            const double dinv = w.Dinv[r];
            if (dinv != 0.0) s += w.C[(size_t)r * SP_WGN_MAX_Y + i] * w.Bd[r] * dinv;
        }
        dy[i] = -(g[i] - s);                  // right-hand side of S dy = -(g - C^T D^-1 b_d)
    }
    __syncthreads();
    // ---- Cholesky S = L L^T in place (lower triangle), forward / backward substitution -----------------------------
    for (int k = 0; k < n_y; ++k) {
        if (tid == 0) {
            const double d = H[k * n_y + k];
            if (!(d > 0.0)) chol_fail = 1;
            else H[k * n_y + k] = sqrt(d);
        }
        __syncthreads();
        if (chol_fail) break;
        const double piv = H[k * n_y + k];
        for (int i = k + 1 + tid; i < n_y; i += SP_BLOCK) H[i * n_y + k] /= piv;
        __syncthreads();
        const int m = n_y - k - 1;
        for (int idx = tid; idx < m * m; idx += SP_BLOCK) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) H[i * n_y + j] -= H[i * n_y + k] * H[j * n_y + k];
        }
        __syncthreads();
    }
    if (!chol_fail) {
        for (int k = 0; k < n_y; ++k) {                  // L y = rhs (column oriented)
            if (tid == 0) dy[k] /= H[k * n_y + k];
            __syncthreads();
            const double yk = dy[k];
            for (int i = k + 1 + tid; i < n_y; i += SP_BLOCK) dy[i] -= H[i * n_y + k] * yk;
            __syncthreads();
        }
        for (int k = n_y - 1; k >= 0; --k) {             // L^T x = y
            if (tid == 0) dy[k] /= H[k * n_y + k];
            __syncthreads();
            const double xk = dy[k];
            for (int i = tid; i < k; i += SP_BLOCK) dy[i] -= H[k * n_y + i] * xk;
            __syncthreads();
        }
    } else {
        // not positive definite at this damping: no step; raise lambda like after a rejected step (the point is unchanged, so
        // the next evaluation repeats the cost and the solve happens again with the larger lambda)
        if (tid == 0) { st[0] *= w.lm_up; st[8] += 1.f; }
        for (int i = tid; i < n_y; i += SP_BLOCK) dy[i] = 0.0;
        __syncthreads();
    }
    // ---- depth steps (back-substitution), clamped like the pair solver's ------------------------------------------
    if (!chol_fail) {
        for (int b = 0; b < w.n_blocks; ++b) {
            const SpWindowBlock bk = w.blocks[b];
            for (int n = tid; n < bk.N; n += SP_BLOCK) {
                const int r = blk_off[b] + n;
                const double dinv = w.Dinv[r];
                if (dinv == 0.0) continue;
                const double* Crow = w.C + (size_t)r * SP_WGN_MAX_Y;
                double s = -w.Bd[r];
                for (int i = 0; i < n_y; ++i) s -= Crow[i] * dy[i];
                double dd = s * dinv;
                dd = fmin(fmax(dd, -0.5), 0.5);
                bk.kld[n] += (float)dd;
            }
        }
        // ---- poses and affine pairs ---------------------------------------------------------------------------------
        for (int i = tid; i < w.n_nodes; i += SP_BLOCK) {
            SpWindowNode& nd = w.nodes[i];
            if (aff_off[i] >= 0) { nd.aff[0] += (float)dy[aff_off[i]]; nd.aff[1] += (float)dy[aff_off[i] + 1]; }
            if (pose_off[i] < 0) continue;
            double d6[6];
            for (int k = 0; k < 6; ++k) d6[k] = dy[pose_off[i] + k];
            if (nd.kind == 0) {
                // T <- T inv(Exp(d)) = T Exp(-d)   (odometery.py:400-403, 861-882), then renormalise
                double xi[6], E[12], Tn[12];
                for (int k = 0; k < 6; ++k) xi[k] = -d6[k];
                wgn_se3_exp(xi, E);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) {
                        double s = (double)nd.T[4 * r] * E[c] + (double)nd.T[4 * r + 1] * E[4 + c] + (double)nd.T[4 * r + 2] * E[8 + c];
                        if (c == 3) s += (double)nd.T[4 * r + 3];
                        Tn[4 * r + c] = s;
                    }
                for (int q = 0; q < 12; ++q) nd.T[q] = (float)Tn[q];
                if (nd.flags & 1) renormalise_rotation(nd.T);
            } else {
                // persistent-tangent node (pose = Exp(a) X, two-frame SfM): the step is a left perturbation of that pose; re-base
                // X <- Exp(d) Exp(a) X and clear the tangent
                Dual<1> ad[6], out[12];
                for (int q = 0; q < 6; ++q) { ad[q].v = nd.a[q]; ad[q].d[0] = 0.f; }
                se3_exp_times<1>(ad, nd.T, out);
                double E[12], Tn[12];
                wgn_se3_exp(d6, E);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) {
                        double s = E[4 * r] * (double)out[c].v + E[4 * r + 1] * (double)out[4 + c].v + E[4 * r + 2] * (double)out[8 + c].v;
                        if (c == 3) s += E[4 * r + 3];
                        Tn[4 * r + c] = s;
                    }
                for (int q = 0; q < 12; ++q) nd.T[q] = (float)Tn[q];
                for (int k = 0; k < 6; ++k) nd.a[k] = 0.f;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < w.n_edges; e += SP_BLOCK) wgn_compose_edge(w, e);
}

}  // namespace

extern "C" {

int sp_window_gn_scratch_doubles(int n_edges, int sum_N, int max_N) {
    if (n_edges <= 0 || sum_N <= 0 || max_N <= 0) return 0;
    return n_edges * (SP_WGN_REC + SP_WGN_SEG * max_N) + n_edges * 36 + sum_N * (SP_WGN_MAX_Y + 2);
}

int sp_window_gn_step(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                      const SpWindowBlock* blocks, int n_blocks, int sum_N, int max_N, const float* span_partials,
                      const float* seg_partials, double* scratch, SpWindowNode* nodes_backup, float* kld_backup, int flags,
                      float lm_up, float lm_down, float lm_min, float conv_tol, float* state, float* losses, int max_losses,
                      void* stream) {
    if (!pairs || !edges || !nodes || !blocks || !span_partials || !seg_partials || !scratch || !nodes_backup || !kld_backup || !state ||
        !losses)
        return SP_EINVAL;
    if (n_edges <= 0 || n_nodes <= 0 || n_blocks <= 0 || sum_N <= 0 || max_N <= 0 || max_losses < 0) return SP_EINVAL;
    if (n_nodes > SP_WGN_MAX_NODES || n_blocks > SP_WGN_MAX_NODES) return SP_ELIMIT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int stride = SP_WGN_REC + SP_WGN_SEG * max_N;
    hipLaunchKernelGGL(k_window_gn_reduce, dim3(n_edges), dim3(SP_BLOCK), 0, s, pairs, edges, span_partials, seg_partials, scratch, stride);
    SP_CHECK_LAUNCH();
    WGnArgs w;
    w.pairs = pairs; w.edges = edges; w.n_edges = n_edges; w.nodes = nodes; w.n_nodes = n_nodes; w.blocks = blocks; w.n_blocks = n_blocks;
    w.max_N = max_N; w.scratch = scratch; w.stride = stride;
    w.Ad = scratch + (size_t)n_edges * stride;
    w.C = w.Ad + (size_t)n_edges * 36;
    w.Dinv = w.C + (size_t)sum_N * SP_WGN_MAX_Y;
    w.Bd = w.Dinv + sum_N;
    w.nodes_backup = nodes_backup; w.kld_backup = kld_backup; w.flags = flags;
    w.lm_up = lm_up; w.lm_down = lm_down; w.lm_min = lm_min; w.conv_tol = conv_tol;
    w.state = state; w.losses = losses; w.max_losses = max_losses;
    hipLaunchKernelGGL(k_window_gn_update, dim3(1), dim3(SP_BLOCK), 0, s, w);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
