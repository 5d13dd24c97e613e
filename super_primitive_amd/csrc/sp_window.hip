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

static_assert(sizeof(SpWindowNode) == 176 && sizeof(SpWindowEdge) == 16 && sizeof(SpWindowBlock) == 32, "window structs are part of the ABI");

#define SP_WIN_REC 28            // doubles per edge record, followed by max_N per-segment sums
#define SP_WIN_MAX_EDGES 1024
#define SP_WIN_LDS_EDGES 96      // windows up to this many edges / 64 nodes are staged in LDS (records, edge list, slot
#define SP_WIN_LDS_BLOCKS 64     // pointers, blocks, nodes, dExp: 52 KB)
#define SP_WIN_LDS_NODES 64

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
        // gradient wrt the SOURCE node's tangent: P = M Exp(-d_src) = Exp(-Ad_M d_src) M  =>  d/dd_src = -Ad_P^T g_left,
        // Ad^T [g_tau; g_phi] = [R^T g_tau ; R^T (g_phi - t x g_tau)]   (computed here, one workgroup per edge, so that the
        // single-workgroup update kernel only has sums left to do)
        const double u0 = rec[4] - (t[1] * gt[2] - t[2] * gt[1]), u1 = rec[5] - (t[2] * gt[0] - t[0] * gt[2]),
                     u2 = rec[6] - (t[0] * gt[1] - t[1] * gt[0]);
        for (int k = 0; k < 3; ++k) {
            rec[21 + k] = -(R[k] * gt[0] + R[3 + k] * gt[1] + R[6 + k] * gt[2]);
            rec[24 + k] = -(R[k] * u0 + R[3 + k] * u1 + R[6 + k] * u2);
        }
        rec[27] = 0.0;
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

#define SP_WIN_LDS_DIRECT 32     // persistent-tangent nodes whose dExp is evaluated in parallel (6 lanes each)

// STAGED is a template parameter, not a run-time select: with it the compiler knows that every access below goes to LDS
// (ds_read / ds_write) -- a pointer that may be LDS or global compiles to flat accesses, ~0.5 us each on this critical path.
template <bool STAGED>
__device__ __forceinline__ void window_update_body(const WinArgs& w) {
    __shared__ float coef[SP_WIN_MAX_EDGES];
    __shared__ int do_update;
    __shared__ float neg_inv_bc1, bc2s_f;
    // One workgroup does all of this, so every global round trip and every serial stretch is on the critical path of an
    // optimiser iteration: the edge records, the edge list, the pose-slot pointers, the log-depth blocks and the node array
    // are staged in LDS once, coalesced; the per-edge heavy lifting (left-tangent and source-node gradients) was done by
    // k_window_reduce with one workgroup per edge; what is left here are short sums, Adam, the fold-in and the composition.
    __shared__ double s_rec[SP_WIN_LDS_EDGES * SP_WIN_REC];
    __shared__ SpWindowEdge s_edge[SP_WIN_LDS_EDGES];
    __shared__ float* s_pose_ptr[SP_WIN_LDS_EDGES];
    __shared__ float* s_aff_ptr[SP_WIN_LDS_EDGES];
    __shared__ SpWindowBlock s_block[SP_WIN_LDS_BLOCKS];
    __shared__ SpWindowNode s_node[SP_WIN_LDS_NODES];
    __shared__ float s_dP[SP_WIN_LDS_DIRECT][6][12];     // d(Exp(a) X)/da_k of the persistent-tangent nodes
    float* st = w.state;
    const int tid = threadIdx.x;
    constexpr bool staged = STAGED;
    SpWindowNode* const nodes = STAGED ? s_node : w.nodes;          // the nodes this launch works on (written back at the end)
    // thread 0 fetches the optimiser state first so that its latency overlaps the staging
    float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;
    double b1t = 1.0, b2t = 1.0;
    if (!w.compose_only && tid == 0) {
        st0 = st[0]; st1 = st[1]; st2 = st[2]; st3 = st[3];
        __builtin_memcpy(&b1t, st + 6, 8); __builtin_memcpy(&b2t, st + 8, 8);
    }
    if (staged) {
        // one coalesced round trip for the node array (176-byte structs = 44 dwords each)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(w.nodes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_node);
        for (int i = tid; i < w.n_nodes * 44; i += SP_BLOCK) dst[i] = src[i];
        for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
            s_edge[e] = w.edges[e];
            s_pose_ptr[e] = w.pairs[e].pose;
            s_aff_ptr[e] = w.pairs[e].aff;
        }
        if (!w.compose_only) {
            for (int i = tid; i < w.n_edges * SP_WIN_REC; i += SP_BLOCK) {
                const int e = i / SP_WIN_REC, k = i - e * SP_WIN_REC;
                s_rec[i] = w.scratch[(size_t)e * w.stride + k];
            }
            for (int b = tid; b < min(w.n_blocks, SP_WIN_LDS_BLOCKS); b += SP_BLOCK) s_block[b] = w.blocks[b];
        }
    }
    if (!w.compose_only) {
        __shared__ int frozen;
        if (tid == 0) frozen = st3 != 0.f;
        __syncthreads();
        if (frozen) return;                             // converged earlier: the window is frozen
        auto REC = [&](int e) -> const double* { if constexpr (STAGED) return s_rec + e * SP_WIN_REC; else return w.scratch + (size_t)e * w.stride; };
        auto EDGE = [&](int e) -> SpWindowEdge { if constexpr (STAGED) return s_edge[e]; else return w.edges[e]; };
        for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
            const double r = REC(e)[0];
            const float sg = w.abs_loss ? (r > 0.0 ? 1.f : (r < 0.0 ? -1.f : 0.f)) : 1.f;
            coef[e] = EDGE(e).weight * sg;
        }
        // d(Exp(a) X)/da_k of the persistent-tangent nodes, one lane per (node, k): forward-mode dual numbers with a single
        // tangent on the same code that evaluates Exp(a) X
        for (int idx = tid; idx < min(w.n_nodes, SP_WIN_LDS_DIRECT) * 6; idx += SP_BLOCK) {
            const int i = idx / 6, k = idx - 6 * i;
            const SpWindowNode& nd = nodes[i];
            if (nd.kind != 1) continue;
            Dual<1> ad[6], out[12];
            for (int q = 0; q < 6; ++q) { ad[q].v = nd.a[q]; ad[q].d[0] = (q == k) ? 1.f : 0.f; }
            se3_exp_times<1>(ad, nd.T, out);
            for (int q = 0; q < 12; ++q) s_dP[i][k][q] = out[q].d[0];
        }
        if (tid == 0) {
            double loss = 0.0;
            for (int e = 0; e < w.n_edges; ++e) {
                const double r = REC(e)[0];
                loss += (double)EDGE(e).weight * (w.abs_loss ? fabs(r) : r);
            }
            const int it = (int)st1;
            const int upd = !(w.skip_first && it == 0);
            do_update = upd;
            if (it < w.max_losses) w.losses[it] = (float)loss;
            float t = st0;
            // beta^t as running products in double (state[6..9]; restarted together with the step count): the same values
            // as pow() to 1e-13 after thousands of steps, without two double pow() calls on the critical path
            if (t == 0.f) { b1t = 1.0; b2t = 1.0; }
            if (upd) { t += 1.f; b1t *= 0.9; b2t *= 0.999; __builtin_memcpy(st + 6, &b1t, 8); __builtin_memcpy(st + 8, &b2t, 8); }
            const double bc1 = 1.0 - b1t, bc2 = 1.0 - b2t;
            neg_inv_bc1 = upd ? (float)(-1.0 / bc1) : 0.f;
            bc2s_f = upd ? (float)sqrt(bc2) : 1.f;
            // relative-loss early stop (odometery.py:907-915): checked AFTER this iteration's update went in
            int done = 0;
            if (w.rel_tol > 0.f) {
                if (it > 0 && fabsf((float)loss - st2) / st2 < w.rel_tol) done = 1;
                st[2] = (float)loss;
            }
            st[0] = t; st[1] = (float)(it + 1); st[3] = done ? 1.f : 0.f; st[4] = (float)loss;
        }
        __syncthreads();
        if (do_update) {
            // ---- log-depth blocks: one wave per block (blocks wave, wave + 4, ...), lanes over its segments -------
            const int wave = tid >> 6, lane = tid & 63;
            for (int b = wave; b < w.n_blocks; b += SP_WAVES) {
                SpWindowBlock bk;
                if (STAGED && b < SP_WIN_LDS_BLOCKS) bk = s_block[b]; else bk = w.blocks[b];
                if (!(bk.lr > 0.f)) continue;                       // frozen (odometery.py:594-603) or constant
                const float ns = bk.lr * neg_inv_bc1;
                for (int n = lane; n < bk.N; n += 64) {
                    double g = 0.0;
                    bool any = false;
                    for (int e = 0; e < w.n_edges; ++e)
                        if (EDGE(e).block == b) { g += (double)coef[e] * w.scratch[(size_t)e * w.stride + SP_WIN_REC + n]; any = true; }
                    if (any) bk.kld[n] += adam_torch((float)g, bk.m[n], bk.v[n], ns, bc2s_f);
                }
            }
            // ---- poses and affine pairs: one lane per node, sums over the edges in edge order -------------------
            for (int i = tid; i < w.n_nodes; i += SP_BLOCK) {
                SpWindowNode& nd = nodes[i];
                double g6[6] = {0, 0, 0, 0, 0, 0}, ga = 0.0, gb = 0.0;
                bool any = false;
                for (int e = 0; e < w.n_edges; ++e) {
                    const SpWindowEdge ed = EDGE(e);
                    if (ed.trg_node != i && ed.src_node != i) continue;
                    const double* rec = REC(e);
                    const double c = (double)coef[e];
                    any = true;
                    if (ed.trg_node == i) {
                        if (nd.kind == 0) {
                            for (int k = 0; k < 6; ++k) g6[k] += c * rec[1 + k];
                        } else if (i < SP_WIN_LDS_DIRECT) {
                            for (int k = 0; k < 6; ++k) {
                                double s = 0.0;
                                for (int q = 0; q < 12; ++q) s += (double)s_dP[i][k][q] * rec[7 + q];
                                g6[k] += c * s;
                            }
                        } else {
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
                        for (int k = 0; k < 6; ++k) g6[k] += c * rec[21 + k];
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
    }
    __syncthreads();
    // ---- relative pose and affine slot of every edge for the next cost pass -----------------------------
    for (int e = tid; e < w.n_edges; e += SP_BLOCK) {
        SpWindowEdge ed;
        float* P;
        float* af;
        if constexpr (STAGED) { ed = s_edge[e]; P = s_pose_ptr[e]; af = s_aff_ptr[e]; }
        else { ed = w.edges[e]; P = w.pairs[e].pose; af = w.pairs[e].aff; }
        const SpWindowNode& nt = nodes[ed.trg_node];
        if (nt.kind == 1) {
            Dual<1> ad[6], out[12];
            for (int q = 0; q < 6; ++q) { ad[q].v = nt.a[q]; ad[q].d[0] = 0.f; }
            se3_exp_times<1>(ad, nt.T, out);
            for (int q = 0; q < 12; ++q) P[q] = out[q].v;
        } else {
            // inv(T_trg) T_src  (tangents are zero between iterations)
            double Rs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ts[3] = {0, 0, 0};
            if (ed.src_node >= 0) {
                const SpWindowNode& ns = nodes[ed.src_node];
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
        if (af) {
            af[0] = ed.src_node >= 0 ? nodes[ed.src_node].aff[0] : 0.f;
            af[1] = ed.src_node >= 0 ? nodes[ed.src_node].aff[1] : 0.f;
            af[2] = nt.aff[0];
            af[3] = nt.aff[1];
        }
    }
    if (staged && !w.compose_only) {                 // nodes back to global memory, coalesced
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_node);
        uint32_t* dst = reinterpret_cast<uint32_t*>(w.nodes);
        for (int i = tid; i < w.n_nodes * 44; i += SP_BLOCK) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(SP_BLOCK) void k_window_update(WinArgs w) {
    if (w.n_edges <= SP_WIN_LDS_EDGES && w.n_nodes <= SP_WIN_LDS_NODES) window_update_body<true>(w);
    else window_update_body<false>(w);
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
