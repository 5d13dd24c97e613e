// Fused optimiser of the reference's three Adam loops ("sp_adam_se3_step"):
//
//   two-frame SfM      odometery/two_frame_sfm.py:116-123,150-207   persistent tangent, pose = Exp(a) X
//   tracking           odometery/odometery.py:300-312,375-407       zero-reset tangent, T_supp <- T_supp inv(Exp(d))
//   windowed mapping   odometery/odometery.py:576-648,756-915       several source keyframes, relative poses
//                      D_trg inv(T_trg) T_src inv(D_src) (:793,817), fold-in + renormalise + tangent reset on every
//                      pose (:861-882), relative-loss early stop (:907-915)
//
// The window is a small graph: NODES are poses (keyframes and supporting frames, with an optional affine brightness
// pair), BLOCKS are the per-keyframe log-depth vectors, EDGES are (source keyframe -> target frame) photometric terms.
// Every edge is one SpPair of the many-pairs cost path, so ONE sp_pairs_cost launch (mode 0) evaluates all of them;
// sp_window_step then is two launches:
//   k_window_reduce (grid = edges)  fixed-order fp64 reduction of an edge's span / segment partials into a small record:
//                                   residual, left-tangent gradient, d/dP (3x4), d/d(a_t, b_t), d/dkld per segment
//   k_window_update (one workgroup) loss, chain rule onto the node tangents (d_trg: left; d_src: -Ad^T; persistent
//                                   tangent: through dExp by dual numbers), torch.optim.Adam semantics per parameter
//                                   group, fold-in, renormalisation, tangent reset, early-stop bookkeeping, and the
//                                   relative pose + affine slot of every edge for the NEXT cost pass.
// No autograd graph, no host synchronisation; an iteration is 3 launches and can be captured in a hipGraph.
#include "sp_solve_device.h"

namespace {

#define SP_WIN_REC 24            // doubles per edge record, followed by max_N per-segment sums
#define SP_WIN_MAX_EDGES 1024

__global__ __launch_bounds__(SP_BLOCK) void k_window_reduce(const SpPair* __restrict__ pairs, const float* __restrict__ partials,
                                                            const float* __restrict__ seg_partials, double* __restrict__ scratch,
                                                            int stride) {
    constexpr int NV = SP_GRAD_PARTIAL_FLOATS;
    __shared__ double sums[NV];
    __shared__ double red[(SP_BLOCK / NV) * NV];
    const int e = blockIdx.x;
    const SpPair& pr = pairs[e];
    reduce_columns<NV>(partials + (size_t)pr.tile0 * NV, pr.n_tiles, sums, red);
    const double scale = 1.0 / (3.0 * (double)pr.P);
    double* rec = scratch + (size_t)e * stride;
    const float* sp = seg_partials + (size_t)pr.rec0 * SP_GRAD_SEG_FLOATS;
    for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
        double s = 0.0;
        const int t0 = pr.seg_tile_off[n], t1 = pr.seg_tile_off[n + 1];
        for (int t = t0; t < t1; t += 8) {       // eight records in flight per trip, added in record order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)min(t + u, t1 - 1) * SP_GRAD_SEG_FLOATS];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (t + u < t1) ? (double)v[u] : 0.0;
        }
        rec[SP_WIN_REC + n] = s * scale;
    }
    if (threadIdx.x == 0) {
        rec[0] = sums[0] * scale;                                     // residual of this edge
        double gR[9], gt[3], R[9], t[3];
        for (int i = 0; i < 3; ++i) {
            gt[i] = sums[1 + i] * scale;
            for (int j = 0; j < 3; ++j) { gR[3 * i + j] = sums[4 + 3 * i + j] * scale; R[3 * i + j] = pr.pose[4 * i + j]; }
            t[i] = pr.pose[4 * i + 3];
        }
        // gradient wrt the left tangent at identity: d/dtau = g_t ; d/dphi = vee(A - A^T), A = R g_R^T + t g_t^T
        double A[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                A[3 * i + j] = R[3 * i] * gR[3 * j] + R[3 * i + 1] * gR[3 * j + 1] + R[3 * i + 2] * gR[3 * j + 2] + t[i] * gt[j];
        rec[1] = gt[0]; rec[2] = gt[1]; rec[3] = gt[2];
        rec[4] = A[5] - A[7]; rec[5] = A[6] - A[2]; rec[6] = A[1] - A[3];
        for (int i = 0; i < 3; ++i) {                                  // d residual / d P (3x4, row-major)
            for (int j = 0; j < 3; ++j) rec[7 + 4 * i + j] = gR[3 * i + j];
            rec[7 + 4 * i + 3] = gt[i];
        }
        rec[19] = sums[14] * scale;                                    // d/da_trg  (d/da_src = -this)
        rec[20] = sums[15] * scale;                                    // d/db_trg  (d/db_src = -this)
    }
}

struct WinArgs {
    const SpPair* pairs; const SpWindowEdge* edges; int n_edges;
    SpWindowNode* nodes; int n_nodes;
    const SpWindowBlock* blocks; int n_blocks;
    const double* scratch; int stride;
    int abs_loss, skip_first; float rel_tol;
    float* state; float* losses; int max_losses;
    int compose_only;
};

// Exp(xi) for xi = [tau, phi] as a 3x4 double matrix
__device__ void se3_exp_d(const double xi[6], double E[12]) {
    // the closed form of se3_retract_left (sp_solve_device.h), kept in double
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

__global__ __launch_bounds__(SP_BLOCK) void k_window_update(WinArgs w) {
    __shared__ float coef[SP_WIN_MAX_EDGES];
    __shared__ int do_update;
    __shared__ float neg_inv_bc1, bc2s_f;
    float* st = w.state;
    const int tid = threadIdx.x;
    if (!w.compose_only) {
        if (st[3] != 0.f) return;                       // converged earlier: the window is frozen
        for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
            const double r = w.scratch[(size_t)e * w.stride];
            const float sg = w.abs_loss ? (r > 0.0 ? 1.f : (r < 0.0 ? -1.f : 0.f)) : 1.f;
            coef[e] = w.edges[e].weight * sg;
        }
        __syncthreads();
        if (tid == 0) {
            double loss = 0.0;
            for (int e = 0; e < w.n_edges; ++e) {
                const double r = w.scratch[(size_t)e * w.stride];
                loss += (double)w.edges[e].weight * (w.abs_loss ? fabs(r) : r);
            }
            const int it = (int)st[1];
            const int upd = !(w.skip_first && it == 0);
            do_update = upd;
            if (it < w.max_losses) w.losses[it] = (float)loss;
            float t = st[0];
            if (upd) t += 1.f;
            const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
            neg_inv_bc1 = upd ? (float)(-1.0 / bc1) : 0.f;
            bc2s_f = upd ? (float)sqrt(bc2) : 1.f;
            // relative-loss early stop (odometery.py:907-915): checked AFTER this iteration's update went in
            const float prev = st[2];
            int done = 0;
            if (w.rel_tol > 0.f) {
                if (it > 0 && fabsf((float)loss - prev) / prev < w.rel_tol) done = 1;
                st[2] = (float)loss;
            }
            st[0] = t; st[1] = (float)(it + 1); st[3] = done ? 1.f : 0.f; st[4] = (float)loss;
        }
        __syncthreads();
        if (do_update) {
            // ---- log-depth blocks ----------------------------------------------------------------
            for (int b = 0; b < w.n_blocks; ++b) {
                const SpWindowBlock bk = w.blocks[b];
                if (!(bk.lr > 0.f)) continue;                       // frozen (odometery.py:594-603) or constant
                const float ns = bk.lr * neg_inv_bc1;
                for (int n = tid; n < bk.N; n += SP_BLOCK) {
                    double g = 0.0;
                    bool any = false;
                    for (int e = 0; e < w.n_edges; ++e)
                        if (w.edges[e].block == b) { g += (double)coef[e] * w.scratch[(size_t)e * w.stride + SP_WIN_REC + n]; any = true; }
                    if (any) bk.kld[n] += adam_torch((float)g, bk.m[n], bk.v[n], ns, bc2s_f);
                }
            }
            // ---- poses and affine pairs ----------------------------------------------------------
            for (int i = tid; i < w.n_nodes; i += SP_BLOCK) {
                SpWindowNode& nd = w.nodes[i];
                double g6[6] = {0, 0, 0, 0, 0, 0}, ga = 0.0, gb = 0.0;
                bool any = false;
                for (int e = 0; e < w.n_edges; ++e) {
                    const SpWindowEdge ed = w.edges[e];
                    const double* rec = w.scratch + (size_t)e * w.stride;
                    const double c = (double)coef[e];
                    if (ed.trg_node == i) {
                        any = true;
                        if (nd.kind == 0) {
                            for (int k = 0; k < 6; ++k) g6[k] += c * rec[1 + k];
                        } else {
                            // persistent tangent: pose = Exp(a) X; d/da through dExp, by dual numbers on the same code
                            Dual<6> ad[6], out[12];
                            for (int q = 0; q < 6; ++q) { ad[q].v = nd.a[q]; for (int k = 0; k < 6; ++k) ad[q].d[k] = (q == k) ? 1.f : 0.f; }
                            se3_exp_times<6>(ad, nd.T, out);
                            for (int k = 0; k < 6; ++k) {
                                double s = 0.0;
                                for (int q = 0; q < 12; ++q) s += (double)out[q].d[k] * rec[7 + q];
                                g6[k] += c * s;
                            }
                        }
                        ga += c * rec[19]; gb += c * rec[20];
                    }
                    if (ed.src_node == i) {
                        // P = M Exp(-d_src) = Exp(-Ad_M d_src) M  =>  d/dd_src = -Ad_P^T g_left,
                        // Ad^T [g_tau; g_phi] = [R^T g_tau ; R^T (g_phi - t x g_tau)]
                        any = true;
                        const float* P = w.pairs[e].pose;
                        const double gt0 = rec[1], gt1 = rec[2], gt2 = rec[3];
                        const double t0 = P[3], t1 = P[7], t2 = P[11];
                        const double u0 = rec[4] - (t1 * gt2 - t2 * gt1), u1 = rec[5] - (t2 * gt0 - t0 * gt2), u2 = rec[6] - (t0 * gt1 - t1 * gt0);
                        for (int k = 0; k < 3; ++k) {
                            g6[k] -= c * ((double)P[k] * gt0 + (double)P[4 + k] * gt1 + (double)P[8 + k] * gt2);
                            g6[3 + k] -= c * ((double)P[k] * u0 + (double)P[4 + k] * u1 + (double)P[8 + k] * u2);
                        }
                        ga -= c * rec[19]; gb -= c * rec[20];
                    }
                }
                if (any && nd.lr_pose > 0.f) {
                    const float ns = nd.lr_pose * neg_inv_bc1;
                    for (int k = 0; k < 6; ++k) nd.a[k] += adam_torch((float)g6[k], nd.m[k], nd.v[k], ns, bc2s_f);
                }
                if (any && nd.lr_aff > 0.f) {
                    const float ns = nd.lr_aff * neg_inv_bc1;
                    nd.aff[0] += adam_torch((float)ga, nd.aff_m[0], nd.aff_v[0], ns, bc2s_f);
                    nd.aff[1] += adam_torch((float)gb, nd.aff_m[1], nd.aff_v[1], ns, bc2s_f);
                }
                if (nd.kind == 0) {
                    // fold-in T <- T inv(Exp(d)) = T Exp(-d), then the tangent is zero again (Adam moments persist)
                    bool nz = false;
                    double xi[6];
                    for (int k = 0; k < 6; ++k) { xi[k] = -(double)nd.a[k]; nz = nz || nd.a[k] != 0.f; }
                    if (nz) {
                        double E[12], Tn[12];
                        se3_exp_d(xi, E);
                        for (int r = 0; r < 3; ++r) {
                            for (int cc = 0; cc < 4; ++cc) {
                                double s = (double)nd.T[4 * r] * E[cc] + (double)nd.T[4 * r + 1] * E[4 + cc] + (double)nd.T[4 * r + 2] * E[8 + cc];
                                if (cc == 3) s += (double)nd.T[4 * r + 3];
                                Tn[4 * r + cc] = s;
                            }
                        }
                        for (int q = 0; q < 12; ++q) nd.T[q] = (float)Tn[q];
                        for (int k = 0; k < 6; ++k) nd.a[k] = 0.f;
                    }
                    if (nd.flags & 1) renormalise_rotation(nd.T);
                }
            }
        }
        __syncthreads();
    }
    // ---- relative pose and affine slot of every edge for the next cost pass -----------------------------
    for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
        const SpWindowEdge ed = w.edges[e];
        const SpWindowNode& nt = w.nodes[ed.trg_node];
        float* P = w.pairs[e].pose;
        if (nt.kind == 1) {
            Dual<1> ad[6], out[12];
            for (int q = 0; q < 6; ++q) { ad[q].v = nt.a[q]; ad[q].d[0] = 0.f; }
            se3_exp_times<1>(ad, nt.T, out);
            for (int q = 0; q < 12; ++q) P[q] = out[q].v;
        } else {
            // inv(T_trg) T_src  (tangents are zero between iterations)
            double Rs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ts[3] = {0, 0, 0};
            if (ed.src_node >= 0) {
                const SpWindowNode& ns = w.nodes[ed.src_node];
                for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Rs[3 * r + cc] = ns.T[4 * r + cc]; ts[r] = ns.T[4 * r + 3]; }
            }
            for (int r = 0; r < 3; ++r) {
                for (int cc = 0; cc < 3; ++cc)
                    P[4 * r + cc] = (float)((double)nt.T[r] * Rs[cc] + (double)nt.T[4 + r] * Rs[3 + cc] + (double)nt.T[8 + r] * Rs[6 + cc]);
                P[4 * r + 3] = (float)((double)nt.T[r] * (ts[0] - nt.T[3]) + (double)nt.T[4 + r] * (ts[1] - nt.T[7]) +
                                       (double)nt.T[8 + r] * (ts[2] - nt.T[11]));
            }
        }
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
        float* af = w.pairs[e].aff;
        if (af) {
            af[0] = ed.src_node >= 0 ? w.nodes[ed.src_node].aff[0] : 0.f;
            af[1] = ed.src_node >= 0 ? w.nodes[ed.src_node].aff[1] : 0.f;
            af[2] = nt.aff[0];
            af[3] = nt.aff[1];
        }
    }
}

}  // namespace

extern "C" {

int sp_window_scratch_doubles(int n_edges, int max_N) { return (n_edges <= 0 || max_N <= 0) ? 0 : n_edges * (SP_WIN_REC + max_N); }

int sp_window_compose(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes, void* stream) {
    if (!pairs || !edges || !nodes || n_edges <= 0 || n_nodes <= 0) return SP_EINVAL;
    if (n_edges > SP_WIN_MAX_EDGES) return SP_ELIMIT;
    WinArgs w{};
    w.pairs = pairs; w.edges = edges; w.n_edges = n_edges; w.nodes = nodes; w.n_nodes = n_nodes; w.compose_only = 1;
    hipLaunchKernelGGL(k_window_update, dim3(1), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), w);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_window_step(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                   const SpWindowBlock* blocks, int n_blocks, int max_N, const float* span_partials, const float* seg_partials,
                   double* scratch, int abs_loss, int skip_first, float rel_tol, float* state, float* losses, int max_losses,
                   void* stream) {
    if (!pairs || !edges || !nodes || !blocks || !span_partials || !seg_partials || !scratch || !state || !losses)
        return SP_EINVAL;
    if (n_edges <= 0 || n_nodes <= 0 || n_blocks <= 0 || max_N <= 0 || max_losses < 0) return SP_EINVAL;
    if (n_edges > SP_WIN_MAX_EDGES) return SP_ELIMIT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int stride = SP_WIN_REC + max_N;
    hipLaunchKernelGGL(k_window_reduce, dim3(n_edges), dim3(SP_BLOCK), 0, s, pairs, span_partials, seg_partials, scratch, stride);
    SP_CHECK_LAUNCH();
    WinArgs w{pairs, edges, n_edges, nodes, n_nodes, blocks, n_blocks, scratch, stride, abs_loss, skip_first, rel_tol, state,
              losses, max_losses, 0};
    hipLaunchKernelGGL(k_window_update, dim3(1), dim3(SP_BLOCK), 0, s, w);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
