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
#include <algorithm>
#include <vector>

namespace {

#define SP_WGN_REC 46            // doubles per edge record: cost, valid points, H_z upper triangle (36), b_z (8); then 10 per segment
#define SP_WGN_SEG 10            // c (8) = [h_pd (6), h_ad, h_bd], D, b_d
#define SP_WGN_MAX_NODES 64
#define SP_WGN_MAX_Y (8 * SP_WGN_MAX_NODES)      // unknowns of the reduced camera system (6 per free pose + 2 per free affine pair)
#define SP_WGN_LDS_Y 192         // ... of which LDS holds this many (packed lower triangle, fp64); larger systems go through global scratch
#define SP_WGN_STATE 16
#define SP_WGN_INLINE_REDUCE 0x100   // (internal flag bit, set by sp_window_gn_step: k_window_gn_update reduces the edges itself)
#define SP_WGN_INLINE_EDGES 4        // ... for depth-less windows of at most this many edges

__device__ __forceinline__ int tri8(int i, int j) { return i * 8 - i * (i - 1) / 2 + (j - i); }      // (i <= j) in the upper triangle of an 8x8

// column `col` (0..15) of the 8 x 16 map z_e = G [y_t ; y_s]: y_t = columns 0..7 (identity), y_s = columns 8..15 (-Ad, -I_2)
__device__ __forceinline__ void wgn_gcol(const double* __restrict__ Ad, int col, double (&g)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) g[p] = 0.0;
    if (col < 8) g[col] = 1.0;
    else if (col < 14) { for (int p = 0; p < 6; ++p) g[p] = -Ad[6 * p + (col - 8)]; }
    else g[col - 8] = -1.0;
}

#define SP_WGN_LOC 272           // doubles per edge of its system in NODE coordinates: 16 x 16 block G^T H_z G over [y_trg (8) ; y_src (8)], then G^T b_z (16)

__device__ __forceinline__ void wgn_reduce_edge(const SpPair* __restrict__ pairs, const SpWindowEdge* __restrict__ edges,
                                                const float* __restrict__ partials, const float* __restrict__ seg_partials,
                                                double* __restrict__ scratch, int stride, double* __restrict__ Ad_all,
                                                double* __restrict__ loc_all, int e, bool with_segments) {
    // (callable from a workgroup of MORE than SP_BLOCK threads -- the update kernel's, when it reduces a depth-less window's edges itself:
    //  the threads beyond SP_BLOCK only take part in the barriers and in the segment loop)
    constexpr int NV = SP_GNA_PARTIAL_FLOATS, NS = SP_GNA_SEG_FLOATS;
    __shared__ double sums[NV];
    __shared__ double red[(SP_BLOCK / NV) * NV];
    __shared__ double Hz[36], bz[8], Ads[36];
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
        Hz[threadIdx.x] = v * scale;
        rec[2 + threadIdx.x] = v * scale;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 72) {
        const int i = threadIdx.x - 64;
        bz[i] = (i < 6 ? sums[22 + i] : sums[32 + (i - 6)]) * scale;
        rec[38 + i] = bz[i];
    }
    if (threadIdx.x >= 128 && threadIdx.x < 137) {
        // Ad of the edge's relative pose P = [R t]: tau' = R tau + [t]x R phi, phi' = R phi  (the pose slot is what the cost pass just used)
        const int r = (threadIdx.x - 128) / 3, c = (threadIdx.x - 128) % 3;
        const float* P = pr.pose;
        const double t0 = P[3], t1 = P[7], t2 = P[11];
        const double tx[9] = {0, -t2, t1, t2, 0, -t0, -t1, t0, 0};
        const double Rrc = P[4 * r + c];
        Ads[6 * r + c] = Rrc;
        Ads[6 * r + 3 + c] = tx[3 * r] * (double)P[c] + tx[3 * r + 1] * (double)P[4 + c] + tx[3 * r + 2] * (double)P[8 + c];
        Ads[6 * (r + 3) + c] = 0.0;
        Ads[6 * (r + 3) + 3 + c] = Rrc;
    }
    __syncthreads();
    if (threadIdx.x < 36) Ad_all[(size_t)e * 36 + threadIdx.x] = Ads[threadIdx.x];
    if (threadIdx.x < SP_BLOCK) {
        // the edge's system in node coordinates: z_e = G [y_trg ; y_src], thread (li, lj) of the 16 x 16 block G^T H_z G (row-major), and
        // G^T b_z from the threads of column 0 -- the update kernel only scatters these into the camera system
        const int li = threadIdx.x >> 4, lj = threadIdx.x & 15;
        double a[8], b[8];
        wgn_gcol(Ads, li, a);
        wgn_gcol(Ads, lj, b);
        double v = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            double row = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) row += Hz[p <= q ? tri8(p, q) : tri8(q, p)] * b[q];
            v += a[p] * row;
        }
        double* loc = loc_all + (size_t)e * SP_WGN_LOC;
        loc[threadIdx.x] = v;
        if (lj == 0) {
            double t = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) t += a[p] * bz[p];
            loc[256 + li] = t;
        }
    }
    // (the per-segment sums feed the Schur terms only: a step in which no depth may move -- flags bits 0 / 2 -- never reads them)
    if (!with_segments) return;
    const float* sp = seg_partials + (size_t)pr.rec0 * NS;
    for (int n = threadIdx.x; n < pr.N; n += blockDim.x) {
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

__global__ __launch_bounds__(SP_BLOCK) void k_window_gn_reduce(const SpPair* __restrict__ pairs, const SpWindowEdge* __restrict__ edges,
                                                               const float* __restrict__ partials, const float* __restrict__ seg_partials,
                                                               double* __restrict__ scratch, int stride, double* __restrict__ Ad_all,
                                                               double* __restrict__ loc_all, int with_segments) {
    wgn_reduce_edge(pairs, edges, partials, seg_partials, scratch, stride, Ad_all, loc_all, blockIdx.x, with_segments != 0);
}

struct WGnArgs {
    const float* span_partials; const float* seg_partials;      // (the per-edge reduction of a multi-window step reads them from here)
    const SpPair* pairs; const SpWindowEdge* edges; int n_edges;
    SpWindowNode* nodes; int n_nodes;
    const SpWindowBlock* blocks; int n_blocks; int max_N;
    double* scratch; int stride;          // edge records
    double* Ad;                           // n_edges x 36: Ad of every edge's relative pose
    double* loc;                          // n_edges x SP_WGN_LOC: every edge's system in node coordinates
    double* C; int ldc;                   // sum N x ldc: row r = coupling of depth unknown r with the nc_b camera unknowns ITS keyframe meets (compact: cols[b][0 .. nc_b))
    double* Dinv;                         // sum N: 1 / (D (1 + lambda)) or 0
    double* Bd;                           // sum N
    double* S; int lds;                   // n_blocks x lds: block b's Schur term C_b^T D_b^-1 C_b over its coupled unknowns (packed lower triangle, nc_b (nc_b + 1) / 2)
    double* Rhs;                          // n_blocks x ldc: C_b^T D_b^-1 b_d
    int* cols;                            // n_blocks x ldc: the coupled camera unknowns of block b, ascending
    int* nc;                              // n_blocks
    double* prof;                         // 16 doubles: time stamps of the update kernel's phases (last call)
    double* Hg; int cap_y;                // the reduced camera system when it does not fit LDS (packed lower triangle); unknowns the caller sized for
    SpWindowNode* nodes_backup; float* kld_backup;
    int flags; float lm_up, lm_down, lm_min, conv_tol;
    float* state; float* losses; int max_losses;
};

// ---- what every kernel of a step must agree on: the numbering of the camera unknowns and the LM decision ---------------------------------
// The head of both kernels is COOPERATIVE: the node flags, the block sizes and the edges' losses are fetched by one thread each (a serial
// walk by thread 0 paid one global-memory latency per node / block / edge: 8 us of each kernel at the reference's extent), staged in LDS,
// and thread 0 only does the ordered arithmetic on them.
// node flags -> offsets of node i's pose (6) / affine (2) unknowns in y, -1 = fixed; returns n_y.  flags[i]: bit 0 = pose free, bit 1 = affine free
__device__ __forceinline__ int wgn_node_flags(const SpWindowNode& nd) { return (nd.lr_pose > 0.f ? 1 : 0) | (nd.lr_aff > 0.f ? 2 : 0); }
__device__ int wgn_number_unknowns(int n_nodes, const int* flags, int* pose_off, int* aff_off) {
    int ny = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const int f = flags[i];               // (flags may alias aff_off: read before the write)
        pose_off[i] = (f & 1) ? ny : -1;
        if (f & 1) ny += 6;
        aff_off[i] = (f & 2) ? ny : -1;
        if (f & 2) ny += 2;
    }
    return ny;
}
// the window's loss, sum_e weight_e * loss_e in edge order: EB edges at a time through `ebuf` (LDS).  Every thread of the workgroup calls it
// (barriers inside); the value is thread 0's.
template <int EB>
__device__ double wgn_loss_sum(const WGnArgs& w, int tid, double* ebuf) {
    double total = 0.0;
    for (int e0 = 0; e0 < w.n_edges; e0 += EB) {
        const int e = e0 + tid;
        if (tid < EB && e < w.n_edges) ebuf[tid] = (double)w.edges[e].weight * w.scratch[(size_t)e * w.stride];
        __syncthreads();
        if (tid == 0) { const int m = min(EB, w.n_edges - e0); for (int q = 0; q < m; ++q) total += ebuf[q]; }
        __syncthreads();
    }
    return total;
}

struct WGnDecision { int dec; int too_many; int converged; float lam; double loss; };      // dec: 0 = step, 1 = reject (restore), 2 = frozen

// PURE (reads the state, writes nothing): k_window_gn_schur of every block and k_window_gn_update evaluate it on the same state.
// loss = wgn_loss_sum; free_depths = depth unknowns that may move (0 in a pose-only phase)
__device__ WGnDecision wgn_decide(const WGnArgs& w, int ny, int cap, double loss, int free_depths) {
    WGnDecision d;
    d.dec = 0; d.too_many = 0; d.converged = 0; d.lam = 0.f;
    d.loss = loss;
    const float* st = w.state;
    if (ny > cap || ny > w.cap_y) { d.too_many = 1; d.dec = 2; return d; }        // more camera unknowns than the caller sized the scratch for: refuse
    if ((ny == 0 && free_depths == 0) || st[6] != 0.f) { d.dec = 2; return d; }    // nothing to optimise / frozen earlier
    const float last = st[1];
    if (last >= 0.f && (float)loss > last * (1.f + 1e-6f) && st[4] == 0.f) d.dec = 1;
    else if (last >= 0.f && st[4] == 0.f && w.conv_tol > 0.f && (last - (float)loss) <= w.conv_tol * last) { d.dec = 2; d.converged = 1; }
    if (d.dec == 0) {
        float lam = st[0];
        if (st[4] == 0.f) lam = fmaxf(lam * w.lm_down, w.lm_min);
        d.lam = lam;
    }
    return d;
}

// packed lower triangle: (i, j), j <= i
__device__ __forceinline__ int ltri(int i, int j) { return ((i * (i + 1)) >> 1) + j; }
// row of packed index t (largest a with a (a + 1) / 2 <= t)
__device__ __forceinline__ int ltri_row(int t) {
    int a = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (((a + 1) * (a + 2) >> 1) <= t) ++a;
    while (((a * (a + 1)) >> 1) > t) --a;
    return a;
}

// ---- the depth side of the arrow system, one source keyframe (block) per blockIdx.x ---------------------------------------------------------
// A depth unknown of keyframe k only meets the frames keyframe k is matched against (its neighbouring keyframes and the supporting frames
// of k and k - 1, odometery.py:770-823): at most 7 of a reference-sized window's 15 free frames.  Per block: the list of those camera
// unknowns (cols), the coupling rows C (N x nc, staged in LDS a chunk of rows at a time), D^-1 with the LM damping, and the block's
// Schur term S_b = C^T D^-1 C + its right-hand side -- pairs (i, j) spread over blockIdx.y tiles of 2 x 256 pairs.  Tile 0 also leaves C,
// D^-1, b_d in global memory for the back-substitution.
#define SP_WGN_SCHUR_LDS 8192      // doubles of a staged chunk of C
#define SP_WGN_SCHUR_ROWS 512
#define SP_WGN_PPT 2               // pairs per thread
#define SP_WGN_BLOCK_EDGES 64     // edges of ONE source keyframe kept as a list in LDS

// S WINDOWS PER LAUNCH (round 6, sp_window_gn_step_multi): `wins` = S argument records in device memory, window = blockIdx.z; the grid is sized
// for the largest window, workgroups beyond a window's own edges / blocks / tiles return at once.  wins == NULL: the one window passed by value.
__global__ __launch_bounds__(SP_BLOCK) void k_window_gn_reduce_multi(const WGnArgs* __restrict__ wins) {
    const WGnArgs& w = wins[blockIdx.z];
    if ((int)blockIdx.x >= w.n_edges) return;
    wgn_reduce_edge(w.pairs, w.edges, w.span_partials, w.seg_partials, w.scratch, w.stride, w.Ad, w.loc, blockIdx.x, !(w.flags & 5));
}

__global__ __launch_bounds__(SP_BLOCK) void k_window_gn_schur(WGnArgs w, const WGnArgs* __restrict__ wins) {
    if (wins) { w = wins[blockIdx.z]; if ((int)blockIdx.x >= w.n_blocks) return; }
    __shared__ double Cs[SP_WGN_SCHUR_LDS];
    __shared__ double Dv[SP_WGN_SCHUR_ROWS], Bv[SP_WGN_SCHUR_ROWS];
    __shared__ int pose_off[SP_WGN_MAX_NODES], aff_off[SP_WGN_MAX_NODES], lpose[SP_WGN_MAX_NODES], laff[SP_WGN_MAX_NODES];
    __shared__ int cols[SP_WGN_MAX_Y];
    __shared__ int nc_s, dec_s, row0_s;
    __shared__ double lam_s;
    const int tid = threadIdx.x, b = blockIdx.x, tile = blockIdx.y;
    const SpWindowBlock bk = w.blocks[b];
    const bool frozen = (w.flags & 5) || !(bk.lr > 0.f);
    __shared__ int bN[SP_WGN_MAX_NODES];          // a block's N, negative if its depths are fixed
    for (int i = tid; i < w.n_nodes; i += SP_BLOCK) { aff_off[i] = wgn_node_flags(w.nodes[i]); lpose[i] = -1; laff[i] = -1; }
    for (int q = tid; q < w.n_blocks; q += SP_BLOCK) { const SpWindowBlock bq = w.blocks[q]; bN[q] = bq.lr > 0.f ? bq.N : -bq.N; }
    const double loss = wgn_loss_sum<SP_BLOCK>(w, tid, Cs);
    __syncthreads();
    // this block's edges, in edge order (an ordered compaction, SP_BLOCK edges at a time: the rows below walk the LIST -- a walk over ALL
    // edges paid a global-memory latency per edge and row for the test "is it mine"), and the nodes they touch (every writer stores the
    // same value).  Beyond SP_WGN_BLOCK_EDGES edges of one block the rows fall back to the walk over all edges.
    __shared__ int be_e[SP_WGN_BLOCK_EDGES], be_trg[SP_WGN_BLOCK_EDGES], be_src[SP_WGN_BLOCK_EDGES];
    __shared__ int be_cnt[SP_BLOCK / 64], be_n;
    __shared__ double be_Ad[SP_WGN_BLOCK_EDGES * 36];          // Ad of the listed edges' relative poses (every row multiplies by it)
    if (tid == 0) be_n = 0;
    __syncthreads();
    if (!frozen)
        for (int e0 = 0; e0 < w.n_edges; e0 += SP_BLOCK) {
            const int e = e0 + tid;
            SpWindowEdge ed{-1, -1, -1, 0.f};
            if (e < w.n_edges) ed = w.edges[e];
            const bool mine = e < w.n_edges && ed.block == b;
            if (mine) {
                lpose[ed.trg_node] = 0;
                if (ed.src_node >= 0) lpose[ed.src_node] = 0;
            }
            const unsigned long long bal = __ballot(mine);
            if ((tid & 63) == 0) be_cnt[tid >> 6] = __popcll(bal);
            __syncthreads();
            int off = be_n;
            for (int q = 0; q < (tid >> 6); ++q) off += be_cnt[q];
            const int slot = off + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if (mine && slot < SP_WGN_BLOCK_EDGES) { be_e[slot] = e; be_trg[slot] = ed.trg_node; be_src[slot] = ed.src_node; }
            __syncthreads();
            if (tid == 0) { int tot = be_n; for (int q = 0; q < SP_BLOCK / 64; ++q) tot += be_cnt[q]; be_n = tot; }
            __syncthreads();
        }
    if (!frozen && be_n <= SP_WGN_BLOCK_EDGES)
        for (int i = tid; i < be_n * 36; i += SP_BLOCK) be_Ad[i] = w.Ad[(size_t)be_e[i / 36] * 36 + (i % 36)];
    __syncthreads();
    if (tid == 0) {
        const int ny = wgn_number_unknowns(w.n_nodes, aff_off, pose_off, aff_off);
        int off = 0, free_depths = 0;
        for (int q = 0; q < w.n_blocks; ++q) {
            if (q == b) row0_s = off;
            off += abs(bN[q]);
            if (!(w.flags & 5) && bN[q] > 0) free_depths += bN[q];
        }
        const WGnDecision d = wgn_decide(w, ny, SP_WGN_MAX_Y, loss, free_depths);
        dec_s = d.dec;
        lam_s = (double)d.lam;
        int nc = 0;                               // touched nodes in node order => cols ascending
        for (int i = 0; i < w.n_nodes; ++i) {
            const bool touched = lpose[i] == 0;
            lpose[i] = -1;
            if (!touched) continue;
            if (pose_off[i] >= 0) { lpose[i] = nc; for (int k = 0; k < 6; ++k) cols[nc++] = pose_off[i] + k; }
            if (aff_off[i] >= 0) { laff[i] = nc; cols[nc++] = aff_off[i]; cols[nc++] = aff_off[i] + 1; }
        }
        nc_s = nc;
    }
    __syncthreads();
    if (dec_s != 0) return;
    const int nc = nc_s, row0 = row0_s;
    const double lam = lam_s;
    const int n_pairs = (nc * (nc + 1)) >> 1;
    const int p_base = tile * (SP_BLOCK * SP_WGN_PPT);
    if (tile > 0 && p_base >= n_pairs) return;
    int pi[SP_WGN_PPT], pj[SP_WGN_PPT];
    double acc[SP_WGN_PPT];
#pragma unroll
    for (int q = 0; q < SP_WGN_PPT; ++q) {
        const int p = p_base + q * SP_BLOCK + tid;
        acc[q] = 0.0;
        if (p < n_pairs) { pi[q] = ltri_row(p); pj[q] = p - ((pi[q] * (pi[q] + 1)) >> 1); } else { pi[q] = -1; pj[q] = 0; }
    }
    double racc[SP_WGN_MAX_Y / SP_BLOCK];
#pragma unroll
    for (int q = 0; q < SP_WGN_MAX_Y / SP_BLOCK; ++q) racc[q] = 0.0;
    const int chunk = nc > 0 ? min(SP_WGN_SCHUR_ROWS, SP_WGN_SCHUR_LDS / nc) : SP_WGN_SCHUR_ROWS;
    for (int r0 = 0; r0 < bk.N; r0 += chunk) {
        const int rows = min(chunk, bk.N - r0);
        for (int n = tid; n < rows; n += SP_BLOCK) {
            double* Crow = Cs + n * nc;
            double D = 0.0, bd = 0.0;
            if (!frozen) {
                for (int i = 0; i < nc; ++i) Crow[i] = 0.0;
                const bool listed = be_n <= SP_WGN_BLOCK_EDGES;
                const int n_walk = listed ? be_n : w.n_edges;
                for (int q = 0; q < n_walk; ++q) {
                    SpWindowEdge ed;
                    int e = q;
                    if (listed) { e = be_e[q]; ed.trg_node = be_trg[q]; ed.src_node = be_src[q]; ed.block = b; }
                    else { ed = w.edges[q]; if (ed.block != b) continue; }
                    const double* o = w.scratch + (size_t)e * w.stride + SP_WGN_REC + (size_t)(r0 + n) * SP_WGN_SEG;
                    D += o[8]; bd += o[9];
                    const int pt = lpose[ed.trg_node], at = laff[ed.trg_node];
                    if (pt >= 0) for (int k = 0; k < 6; ++k) Crow[pt + k] += o[k];
                    if (at >= 0) { Crow[at] += o[6]; Crow[at + 1] += o[7]; }
                    if (ed.src_node >= 0) {
                        const int ps = lpose[ed.src_node], as = laff[ed.src_node];
                        if (ps >= 0) {
                            const double* Ad = listed ? be_Ad + q * 36 : w.Ad + (size_t)e * 36;
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
            const double dinv = (!frozen && Dd > 1e-12) ? 1.0 / Dd : 0.0;
            Dv[n] = dinv; Bv[n] = bd;
            if (tile == 0) {
                const int r = row0 + r0 + n;
                w.Dinv[r] = dinv; w.Bd[r] = bd;
                double* Cg = w.C + (size_t)r * w.ldc;
                for (int i = 0; i < nc; ++i) Cg[i] = Crow[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SP_WGN_PPT; ++q) {
            if (pi[q] < 0) continue;
            double s = acc[q];
            for (int r = 0; r < rows; ++r) s += Cs[r * nc + pi[q]] * Cs[r * nc + pj[q]] * Dv[r];       // rows in order: a fixed summation order
            acc[q] = s;
        }
        if (tile == 0) {
#pragma unroll
            for (int q = 0; q < SP_WGN_MAX_Y / SP_BLOCK; ++q) {
                const int i = q * SP_BLOCK + tid;
                if (i >= nc) continue;
                double s = racc[q];
                for (int r = 0; r < rows; ++r) s += Cs[r * nc + i] * Bv[r] * Dv[r];
                racc[q] = s;
            }
        }
        __syncthreads();
    }
    double* Sb = w.S + (size_t)b * w.lds;
#pragma unroll
    for (int q = 0; q < SP_WGN_PPT; ++q)
        if (pi[q] >= 0) Sb[p_base + q * SP_BLOCK + tid] = acc[q];
    if (tile == 0) {
#pragma unroll
        for (int q = 0; q < SP_WGN_MAX_Y / SP_BLOCK; ++q) {
            const int i = q * SP_BLOCK + tid;
            if (i < nc) { w.Rhs[(size_t)b * w.ldc + i] = racc[q]; w.cols[(size_t)b * w.ldc + i] = cols[i]; }
        }
        if (tid == 0) w.nc[b] = nc;
    }
}

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

// The reduced camera system S (n_y x n_y, fp64, packed lower triangle) lives in LDS when n_y <= LDS_Y (instantiations 64 / 128 / 192:
// 17 / 66 / 148 KB of the CU's 160 KB) and in global scratch (w.Hg) in the LDS_Y = 0 instantiation: a reference-sized mapping window
// (config/tum/odom_desk.yaml: window_size 5, two supporting frames per keyframe + the two running ones, poses and affine pairs all
// free -- odometery.py:523-575, 611-616) is 14 free nodes = 112 unknowns, a window with three supporting frames per keyframe 168.
#define SP_WGN_THREADS 1024     // threads of the update kernel: 4 waves per SIMD -- its phases are chains of LDS / global latencies, not throughput
#define SP_WGN_GRID 32          // ... as a SP_WGN_GRID x SP_WGN_GRID grid over the trailing block of the factorisation
#define WGN_STAMP(i) do { if (tid == 0) w.prof[i] = (double)wall_clock64(); } while (0)      // 100 MHz constant clock; phase boundaries of the last step
// threads of the update kernel: 1024 (4 waves per SIMD -- its phases are chains of LDS / global latencies, not throughput) except the
// 128-unknown instantiation, whose factorisation keeps 36 doubles of the matrix per thread in registers: 512 threads = 256 VGPRs each
__host__ __device__ constexpr int wgn_update_threads(int lds_y) { return lds_y == 128 ? 512 : lds_y == 192 ? 256 : SP_WGN_THREADS; }
template <int LDS_Y>
__global__ __launch_bounds__(wgn_update_threads(LDS_Y)) void k_window_gn_update(WGnArgs w, const WGnArgs* __restrict__ wins) {
    if (wins) w = wins[blockIdx.z];
    constexpr int CAP = LDS_Y > 0 ? LDS_Y : SP_WGN_MAX_Y;
    constexpr int NTHR = wgn_update_threads(LDS_Y);
    __shared__ double Hs[LDS_Y > 0 ? LDS_Y * (LDS_Y + 1) / 2 : 1];
    __shared__ double g[CAP], dy[CAP], dinvs[CAP];
    __shared__ double g0[CAP];             // the camera-side gradient as assembled (before the Schur terms): the predicted exit's b . delta
    constexpr int PANELS = LDS_Y == 0 ? 0 : (LDS_Y > 128 ? 1 : 2);      // column panels of the factorisation (two: written while the other is read)
    __shared__ __align__(16) double panel[PANELS > 0 ? PANELS : 1][LDS_Y > 0 ? LDS_Y : 1][4];      // [row][column of the panel]
    __shared__ int pose_off[SP_WGN_MAX_NODES], aff_off[SP_WGN_MAX_NODES];
    __shared__ int blk_off[SP_WGN_MAX_NODES + 1];
    __shared__ int n_y_s, decision, fail_s, free_s;
    __shared__ double lam_s;
    double* H;
    if constexpr (LDS_Y > 0) H = Hs; else H = w.Hg;
    const int tid = threadIdx.x;
    float* st = w.state;
    WGN_STAMP(0);
    if constexpr (LDS_Y == 64) if (w.flags & SP_WGN_INLINE_REDUCE) {      // (the smallest instantiation only: the routine's LDS buffers come on top)
        // a depth-less window of a few edges (the tracker's: one edge, eight unknowns): the edges' reductions HERE instead of in a launch of
        // their own in front of this one -- same routine, same sums; a launch boundary less per iteration of a chain of small dependent launches
        for (int e = 0; e < w.n_edges; ++e) {
            wgn_reduce_edge(w.pairs, w.edges, w.span_partials, w.seg_partials, w.scratch, w.stride, w.Ad, w.loc, e, false);
            __syncthreads();                  // (the routine's LDS buffers, before the next edge overwrites them)
        }
    }
    for (int i = tid; i < w.n_nodes; i += NTHR) aff_off[i] = wgn_node_flags(w.nodes[i]);
    for (int b = tid; b < w.n_blocks; b += NTHR) { const SpWindowBlock bq = w.blocks[b]; blk_off[b] = bq.lr > 0.f ? bq.N : -bq.N; }
    const double loss_now = wgn_loss_sum<64>(w, tid, g);
    __syncthreads();
    if (tid == 0) {
        const int ny = wgn_number_unknowns(w.n_nodes, aff_off, pose_off, aff_off);
        n_y_s = ny;
        fail_s = 0;
        int off = 0, free_depths = 0;
        for (int b = 0; b < w.n_blocks; ++b) {
            const int N = blk_off[b];
            if (!(w.flags & 5) && N > 0) free_depths += N;
            blk_off[b] = off;
            off += abs(N);
        }
        blk_off[w.n_blocks] = off;
        free_s = free_depths;
        // ---- loss, LM decision (the one k_window_gn_schur took), committed to the state ---------------------------------
        const WGnDecision d = wgn_decide(w, ny, CAP, loss_now, free_depths);
        if (d.too_many) { st[9] = 1.f; st[6] = 1.f; }
        if (d.dec != 2 || d.converged) {
            const int it = (int)st[5];
            if (it < w.max_losses) w.losses[it] = (float)d.loss;
            st[5] = (float)(it + 1);
            st[7] = (float)d.loss;
        }
        if (d.converged) st[6] = 1.f;
        if (d.dec == 1) { st[0] *= w.lm_up; st[3] += 1.f; st[4] = 1.f; }
        if (d.dec == 0) { st[0] = d.lam; st[1] = (float)d.loss; st[2] += 1.f; st[4] = 0.f; }
        lam_s = (double)d.lam;
        decision = d.dec;
    }
    __syncthreads();
    const int n_y = n_y_s;
    const int ldc = w.ldc;
    WGN_STAMP(1);
    if (decision == 2) return;
    if (decision == 1) {
        // undo the previous step: nodes (pose, tangent, affine) and log-depths back to the stored point
        for (int i = tid; i < w.n_nodes * 44; i += NTHR) reinterpret_cast<uint32_t*>(w.nodes)[i] = reinterpret_cast<const uint32_t*>(w.nodes_backup)[i];
        for (int b = 0; b < w.n_blocks; ++b)
            for (int n = tid; n < w.blocks[b].N; n += NTHR) w.blocks[b].kld[n] = w.kld_backup[blk_off[b] + n];
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < w.n_edges; e += NTHR) wgn_compose_edge(w, e);
        return;
    }
    const double lam = lam_s;
    // ---- back up the point we are about to leave; clear the system ------------------------------------------------
    for (int i = tid; i < w.n_nodes * 44; i += NTHR) reinterpret_cast<uint32_t*>(w.nodes_backup)[i] = reinterpret_cast<const uint32_t*>(w.nodes)[i];
    for (int b = 0; b < w.n_blocks; ++b)
        for (int n = tid; n < w.blocks[b].N; n += NTHR) w.kld_backup[blk_off[b] + n] = w.blocks[b].kld[n];
    const int n_tri = (n_y * (n_y + 1)) >> 1;
    for (int i = tid; i < n_tri; i += NTHR) H[i] = 0.0;
    for (int i = tid; i < n_y; i += NTHR) g[i] = 0.0;
    __syncthreads();
    WGN_STAMP(2);
    // ---- assembly: scatter every edge's 16 x 16 node-coordinate block (k_window_gn_reduce), thread (i, j), one edge at a time (edges
    //      share entries: a fixed order keeps the sums reproducible); the loads of a group of edges are issued before its barriers
    // (ONE loop with ONE barrier site, executed by every thread -- the scatter is what is guarded, not the barrier)
    if (n_y > 0) {
        const bool worker = tid < 256;
        const int li = (tid >> 4) & 15, lj = tid & 15;
        for (int e0 = 0; e0 < w.n_edges; e0 += 8) {
            double v[8], t[8];
            int gi[8], gj[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q;
                gi[q] = gj[q] = -1; v[q] = t[q] = 0.0;
                if (!worker || e >= w.n_edges) continue;
                const SpWindowEdge ed = w.edges[e];
                auto global_index = [&](int l) -> int {
                    const int node = l < 8 ? ed.trg_node : ed.src_node;
                    if (node < 0) return -1;
                    const int k = l & 7;
                    const int base = k < 6 ? pose_off[node] : aff_off[node];
                    return base < 0 ? -1 : base + (k < 6 ? k : k - 6);
                };
                gi[q] = global_index(li); gj[q] = global_index(lj);
                const double* loc = w.loc + (size_t)e * SP_WGN_LOC;
                if (gi[q] >= 0 && gj[q] >= 0 && gi[q] >= gj[q]) v[q] = loc[tid];
                if (gi[q] >= 0 && lj == 0) t[q] = loc[256 + li];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (worker && e0 + q < w.n_edges) {
                    if (gi[q] >= 0 && gj[q] >= 0 && gi[q] >= gj[q]) H[ltri(gi[q], gj[q])] += v[q];
                    if (gi[q] >= 0 && lj == 0) g[gi[q]] += t[q];
                }
                __syncthreads();
            }
        }
    }
    WGN_STAMP(3);
    // ---- LM damping of the camera block, then minus the blocks' Schur terms (k_window_gn_schur), block after block ----
    for (int i = tid; i < n_y; i += NTHR) { H[ltri(i, i)] = H[ltri(i, i)] * (1.0 + lam) + 1e-12; g0[i] = g[i]; g[i] = -g[i]; }
    __syncthreads();
    // (no depth may move -- a pose-only phase, or the caller's word that every block is fixed, flags bit 2: k_window_gn_schur was not launched,
    //  its terms would all be zero)
    const int n_schur_blocks = free_s > 0 ? w.n_blocks : 0;
    for (int b = 0; b < n_schur_blocks; ++b) {
        const int nc = w.nc[b];
        const int* cb = w.cols + (size_t)b * ldc;
        const double* Sb = w.S + (size_t)b * w.lds;
        const int np = (nc * (nc + 1)) >> 1;
        for (int p = tid; p < np; p += NTHR) {
            const int i = ltri_row(p), j = p - ((i * (i + 1)) >> 1);
            H[ltri(cb[i], cb[j])] -= Sb[p];                            // cols ascending: cb[i] >= cb[j]
        }
        for (int i = tid; i < nc; i += NTHR) g[cb[i]] += w.Rhs[(size_t)b * ldc + i];
        __syncthreads();
    }
    for (int i = tid; i < n_y; i += NTHR) dy[i] = g[i];      // right-hand side of S dy = -(g - C^T D^-1 b_d)
    __syncthreads();
    WGN_STAMP(4);
    // ---- S = L D L^T, right-looking by blocks of 4 columns; column j keeps the UNSCALED entries c_ij = l_ij d_j.
    //      LDS-resident systems (LDS_Y > 0): the trailing matrix lives in REGISTERS -- thread (ti, tj) of a G x G grid owns the entries
    //      (G a + ti, G b + tj), a >= b, of the symmetric matrix (block-cyclic: every thread stays busy until the last columns; 36
    //      doubles at 128 unknowns) -- and only the 4-column panel of a step travels through LDS:
    //        (a) the four thread columns that own the panel write it (all rows);                                         barrier
    //        (b) every thread factors the 4 x 4 diagonal block in registers (10 broadcast reads: the positive-definiteness test is
    //            uniform); thread i eliminates inside ITS row of the panel, leaves the final c_i. in the panel and in H (the
    //            substitutions read L there) and takes the row's right-hand side along: y_i -= sum_q l_iq y~_q, i.e. the forward
    //            substitution L y = rhs is done when the factorisation is;                                                barrier
    //        (c) rank-4 update of the register tiles: per panel column T + T LDS reads (rows broadcast, columns consecutive) for
    //            T (T + 1) / 2 multiply-adds, no load / store of the matrix.
    //      The global-scratch instantiation (LDS_Y = 0, > 192 unknowns) keeps the in-place form: (1) every thread factors the diagonal
    //      block and eliminates inside its row of the panel, (2) rank-4 update of the trailing triangle, thread (ti, tj) of a 32 x 32 grid
    //      over rows = ti, columns = tj (mod 32): 4 + 1 accesses per 4 multiply-adds.  Two barriers per 4 columns.
    bool fail = false;
    if constexpr (LDS_Y > 0) {
        constexpr int G = 16, T = CAP / G;
        static_assert(G * G <= NTHR && NTHR >= CAP && CAP % G == 0 && G % 4 == 0, "tile grid of the register-resident factorisation");
        const int ti = tid / G, tj = tid % G;
        const bool owner = tid < G * G;
        double R[T][T];                       // (a, b), b <= a only
        double rhs_keep = 0.0;
        if (owner) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const int i = G * a + ti, j = G * b + tj;      // a diagonal tile holds both triangles (the update is symmetric)
                    R[a][b] = (i < n_y && j < n_y) ? H[ltri(max(i, j), min(i, j))] : (i == j ? 1.0 : 0.0);     // identity beyond n_y
                }
        }
        int buf = 0;
        auto row4 = [](const double (*P)[4], int i, double (&v)[4]) {          // one panel row: two 16-byte LDS reads
            const double2 lo = *reinterpret_cast<const double2*>(&P[i][0]), hi = *reinterpret_cast<const double2*>(&P[i][2]);
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
        };
#pragma unroll
        for (int b0 = 0; b0 < T; ++b0) {      // tile column of the panel: STATIC, so that the tiles a step touches are known at compile time
        for (int k = G * b0; k < min(G * (b0 + 1), n_y); k += 4) {
            double (*P)[4] = panel[buf];
            const int c0 = k - G * b0;
            const bool detail = tid == 0 && k == 4;           // phase stamps of the second block step (diagnostics)
            if (detail) w.prof[10] = (double)wall_clock64();
            if (owner && (unsigned)(tj - c0) < 4u) {
                const int q = tj - c0;
#pragma unroll
                for (int a = b0; a < T; ++a) P[G * a + ti][q] = R[a][b0];      // (tile rows above the panel's are finished and never read)
            }
            __syncthreads();
            if (detail) w.prof[11] = (double)wall_clock64();
            double dinv[4];
            if (owner) {
                double c[4][4], yt[4], r[4];
                const int i = tid;
                const bool below = i >= k + 4 && i < n_y;             // a row of the panel under the diagonal block
#pragma unroll
                for (int a = 0; a < 4; ++a) row4(P, k + a, c[a]);      // (the upper triangle comes along, unused)
                row4(P, i < CAP ? i : 0, r);                           // (issued with the block's loads: the row's latency hides behind the block's chain)
                double yi = i < CAP ? dy[i] : 0.0;
#pragma unroll
                for (int a = 0; a < 4; ++a) yt[a] = k + a < n_y ? dy[k + a] : 0.0;
                bool bad = false;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int q = 0; q < a; ++q) {
                        const double l = c[a][q] * dinv[q];
#pragma unroll
                        for (int q2 = q + 1; q2 <= a; ++q2) c[a][q2] -= l * c[q2][q];
                        yt[a] -= l * yt[q];
                    }
                    if (!(c[a][a] > 0.0)) bad = true;                 // (columns beyond n_y: the identity)
                    // 1 / d by v_rcp_f64 and two Newton steps: a third of the IEEE division's dependent chain, which every block step waits for
                    const double d = c[a][a];
                    double y = __builtin_amdgcn_rcp(d);
                    y = fma(fma(-d, y, 1.0), y, y);
                    dinv[a] = fma(fma(-d, y, 1.0), y, y);
                }
                if (detail) w.prof[12] = (double)wall_clock64();
                if (bad) { if (tid == 0) fail_s = 1; }
                else if (below) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double l = r[q] * dinv[q];
#pragma unroll
                        for (int q2 = q + 1; q2 < 4; ++q2) r[q2] -= l * c[q2][q];
                        yi -= l * yt[q];
                    }
                    *reinterpret_cast<double2*>(&P[i][0]) = double2{r[0], r[1]};
                    *reinterpret_cast<double2*>(&P[i][2]) = double2{r[2], r[3]};
                    const int rowi = ltri(i, k);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (k + q < n_y) H[rowi + q] = r[q];
                    dy[i] = yi;
                } else if (i >= k && i < n_y) {                       // a row of the diagonal block: its final entries
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        if (i == k + a) {
#pragma unroll
                            for (int q = 0; q <= a; ++q) H[ltri(i, k + q)] = c[a][q];
                            dinvs[i] = dinv[a];
                            rhs_keep = yt[a];
                        }
                }
            }
            if (detail) w.prof[13] = (double)wall_clock64();
            __syncthreads();
            if (detail) w.prof[14] = (double)wall_clock64();
            if (fail_s) { fail = true; break; }
            if (tid >= k && tid < k + 4 && tid < n_y) dy[tid] = rhs_keep;       // (after the barrier: every thread has read the block's right-hand side)
            if (owner) {
                // entries (i, j) with i or j up to the diagonal block's rows are final -- never read from the registers again -- so nothing
                // is masked: tile rows / columns before the panel's are skipped (statically), the panel's own are updated whole.
                // Straight-line code: the loads of all tile rows are in flight before the first multiply-add
                double cv[T][4];
#pragma unroll
                for (int a = b0; a < T; ++a) row4(P, G * a + tj, cv[a]);
#pragma unroll
                for (int a = b0; a < T; ++a) {
                    double rv[4];
                    row4(P, G * a + ti, rv);
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] *= dinv[q];
#pragma unroll
                    for (int b = b0; b <= a; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) R[a][b] -= rv[q] * cv[b][q];
                }
            }
            if (detail) w.prof[15] = (double)wall_clock64();
            if constexpr (PANELS == 2) buf ^= 1; else __syncthreads();
        }
        if (fail) break;
        }
        __syncthreads();                      // (the last block's right-hand side entries, before the substitution reads them)
    } else {
    {
        const int ti = tid / SP_WGN_GRID, tj = tid % SP_WGN_GRID;
        for (int k = 0; k < n_y && !fail; k += 4) {
            const int B = min(4, n_y - k);
            double c[4][4], dinv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q <= a; ++q) c[a][q] = a < B ? H[ltri(k + a, k + q)] : (a == q ? 1.0 : 0.0);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // row a of the diagonal block: eliminate with the columns q < a (rows q are final by now)
#pragma unroll
                for (int q = 0; q < a; ++q)
#pragma unroll
                    for (int q2 = q + 1; q2 <= a; ++q2) c[a][q2] -= c[a][q] * c[q2][q] * dinv[q];
                if (!(c[a][a] > 0.0)) fail = true;
                dinv[a] = 1.0 / c[a][a];
            }
            if (fail) break;
            if (tid < B) dinvs[k + tid] = tid == 0 ? dinv[0] : tid == 1 ? dinv[1] : tid == 2 ? dinv[2] : dinv[3];
            for (int i = k + B + tid; i < n_y; i += NTHR) {
                double r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = q < B ? H[ltri(i, k + q)] : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int q2 = q + 1; q2 < 4; ++q2) r[q2] -= r[q] * c[q2][q] * dinv[q];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (q < B) H[ltri(i, k + q)] = r[q];
            }
            __syncthreads();
            if (tid < 10) {               // the diagonal block's final entries (thread t writes one of the 10; after the barrier: every thread has read the block)
                int a = 0, q = tid;
                while (q > a) { q -= a + 1; ++a; }
                double v = 0.0;
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                    for (int qq = 0; qq <= aa; ++qq) if (aa == a && qq == q) v = c[aa][qq];
                if (a < B) H[ltri(k + a, k + q)] = v;
            }
            for (int i = k + B + ti; i < n_y; i += SP_WGN_GRID) {
                double cis[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) cis[q] = q < B ? H[ltri(i, k + q)] * dinv[q] : 0.0;
                const int rowi = (i * (i + 1)) >> 1;
                // four columns j at a time: all their loads first (a load-modify-store per element would serialise on LDS latency:
                // the compiler cannot move the next element's loads above a store that may alias them)
                for (int j0 = k + B + tj; j0 <= i; j0 += 4 * SP_WGN_GRID) {
                    double hj[4][4], old[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + SP_WGN_GRID * u;
                        const bool ok = j <= i;
                        const int rowj = ok ? ((j * (j + 1)) >> 1) + k : 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) hj[u][q] = (ok && q < B) ? H[rowj + q] : 0.0;
                        old[u] = ok ? H[rowi + j] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + SP_WGN_GRID * u;
                        if (j <= i) H[rowi + j] = old[u] - (cis[0] * hj[u][0] + cis[1] * hj[u][1] + cis[2] * hj[u][2] + cis[3] * hj[u][3]);
                    }
                }
            }
            __syncthreads();
        }
    }
    }
    WGN_STAMP(5);
    if (!fail) {
        // L y = rhs, z = D^-1 y, L^T x = z by ONE wave, wave-synchronously: lane l holds entries l, l + 64, ... of the vector in registers,
        // the pivot entry of a column travels through a scalar register (v_readlane) -- no block barrier per column
        if (tid < 64) {
            constexpr int PER = (CAP + 63) / 64;
            double x[PER];
            auto bcast = [&](int k) -> double {
                double xk = 0.0;
#pragma unroll
                for (int q = 0; q < PER; ++q) if ((k >> 6) == q) xk = x[q];
                const int lo = __builtin_amdgcn_readlane(__double2loint(xk), k & 63), hi = __builtin_amdgcn_readlane(__double2hiint(xk), k & 63);
                return __hiloint2double(hi, lo);
            };
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; x[q] = i < n_y ? dy[i] : 0.0; }
            // (the matrix entries and 1 / d_k of step k + 1 are loaded while step k's pivot travels: the loop-carried chain is
            //  readlane -> multiply -> fma only)
            double hn[PER];
            if constexpr (LDS_Y == 0) {        // (LDS-resident systems: the factorisation took the right-hand side along)
                double dn = n_y > 0 ? dinvs[0] : 0.0;
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; hn[q] = (i > 0 && i < n_y) ? H[ltri(i, 0)] : 0.0; }
                for (int k = 0; k < n_y; ++k) {                  // forward: y_i -= (c_ik / d_k) y_k for i > k
                    double hc[PER];
                    const double dc = dn;
#pragma unroll
                    for (int q = 0; q < PER; ++q) hc[q] = hn[q];
                    if (k + 1 < n_y) {
                        dn = dinvs[k + 1];
#pragma unroll
                        for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; hn[q] = (i > k + 1 && i < n_y) ? H[ltri(i, k + 1)] : 0.0; }
                    }
                    const double f = bcast(k) * dc;
#pragma unroll
                    for (int q = 0; q < PER; ++q) x[q] -= hc[q] * f;
                }
            }
            double rd[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; rd[q] = i < n_y ? dinvs[i] : 0.0; x[q] *= rd[q]; }
            // backward: x_i -= (c_ki / d_i) x_k for i < k, GS columns (rows of the packed triangle) per trip; the raw entries of the
            // NEXT ones are loaded while this trip's chain (readlane -> fma) runs, and scaled when they are used
            constexpr int GS = PER <= 3 ? 4 : 1;          // (columns per trip: what the registers hold)
            double hraw[GS][PER];
            auto load4 = [&](int ktop) {
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const int k = ktop - u, rowk = k > 0 ? ((k * (k + 1)) >> 1) : 0;
#pragma unroll
                    for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; hraw[u][q] = (k > 0 && i < k) ? H[rowk + i] : 0.0; }
                }
            };
            load4(n_y - 1);
            for (int ktop = n_y - 1; ktop >= 0; ktop -= GS) {
                double hc[GS][PER];
#pragma unroll
                for (int u = 0; u < GS; ++u)
#pragma unroll
                    for (int q = 0; q < PER; ++q) hc[u][q] = hraw[u][q] * rd[q];
                load4(ktop - GS);
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const int k = ktop - u;
                    if (k < 0) break;
                    const double xk = bcast(k);
#pragma unroll
                    for (int q = 0; q < PER; ++q) x[q] -= hc[u][q] * xk;
                }
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int i = q * 64 + tid; if (i < n_y) dy[i] = x[q]; }
        }
        __syncthreads();
    } else {
        // not positive definite at this damping: no step.  Raise lambda and mark the point as "rejected last" (st[4]): the next call
        // evaluates the same point, must neither test it for convergence (its loss equals the stored one) nor lower lambda, and solves
        // again with the larger damping.
        if (tid == 0) { st[0] *= w.lm_up; st[8] += 1.f; st[4] = 1.f; st[2] -= 1.f; }
        for (int i = tid; i < n_y; i += NTHR) dy[i] = 0.0;
        __syncthreads();
    }
    WGN_STAMP(6);
    // ---- depth steps (back-substitution), clamped like the pair solver's ------------------------------------------
    double gain = 0.0;                     // -(b . delta): what the step is predicted to buy (first-order change of the loss, flags bit 1)
    if (!fail) {
        for (int i = tid; i < n_y; i += NTHR) gain -= g0[i] * dy[i];
        for (int b = 0; b < n_schur_blocks; ++b) {
            const SpWindowBlock bk = w.blocks[b];
            const int nc = w.nc[b];
            const int* cb = w.cols + (size_t)b * ldc;
            for (int n0 = 0; n0 < bk.N; n0 += NTHR / 4) {        // four lanes per row, their partial sums added in a fixed order
                const int n = n0 + (tid >> 2), part = tid & 3;
                const bool row_ok = n < bk.N;
                const int r = blk_off[b] + (row_ok ? n : 0);
                const double dinv = row_ok ? w.Dinv[r] : 0.0;
                const double* Crow = w.C + (size_t)r * ldc;
                double s = 0.0;
                if (dinv != 0.0)
                    for (int i = part; i < nc; i += 4) s -= Crow[i] * dy[cb[i]];
                const double s1 = __shfl_xor(s, 1);
                s += s1;
                const double s2 = __shfl_xor(s, 2);
                s += s2;
                if (part == 0 && dinv != 0.0) {
                    double dd = (s - w.Bd[r]) * dinv;
                    dd = fmin(fmax(dd, -0.5), 0.5);
                    bk.kld[n] += (float)dd;
                    gain -= w.Bd[r] * dd;
                }
            }
        }
        WGN_STAMP(7);
        // ---- poses and affine pairs ---------------------------------------------------------------------------------
        for (int i = tid; i < w.n_nodes; i += NTHR) {
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
    if ((w.flags & 2) && w.conv_tol > 0.f) {
        // PREDICTED EXIT (as SP_PHASE_PREDICTED_EXIT of the pair solver, include/sp_hip.h): the step just taken is predicted to buy less than
        // conv_tol of the loss -> the phase is over now, without the evaluation that would confirm it (on the few-edge windows of the
        // config-3 chain one evaluation is a fifth of a tracking phase).  Not under a heavy damping, not after a failed factorisation.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gain += __shfl_xor(gain, o, 64);
        __syncthreads();                   // (dinvs: the factorisation's reciprocals, no longer read)
        if ((tid & 63) == 0) dinvs[tid >> 6] = gain;
        __syncthreads();
        if (tid == 0 && !fail && lam <= 1e-2) {
            double tot = 0.0;
            for (int k = 0; k < NTHR / 64; ++k) tot += dinvs[k];
            if (tot <= (double)w.conv_tol * (double)st[7]) st[6] = 1.f;
        }
    }
    __threadfence_block();
    __syncthreads();
    WGN_STAMP(8);
    for (int e = tid; e < w.n_edges; e += NTHR) wgn_compose_edge(w, e);
    WGN_STAMP(9);
}

}  // namespace

extern "C" {

// n_unknowns: camera unknowns the window can have (6 per node with lr_pose > 0 + 2 per node with lr_aff > 0)
int sp_window_gn_scratch_doubles(int n_edges, int n_blocks, int sum_N, int max_N, int n_unknowns) {
    if (n_edges <= 0 || n_blocks <= 0 || sum_N <= 0 || max_N <= 0 || n_unknowns < 0 || n_unknowns > SP_WGN_MAX_Y) return 0;
    const long long ldc = (n_unknowns + 1) & ~1;
    const long long tri = (long long)n_unknowns * (n_unknowns + 1) / 2;
    const long long hg = n_unknowns > SP_WGN_LDS_Y ? tri : 0;
    const long long n = (long long)n_edges * (SP_WGN_REC + SP_WGN_SEG * max_N) + (long long)n_edges * (36 + SP_WGN_LOC) + (long long)sum_N * (ldc + 2) + (long long)n_blocks * (tri + 2 * ldc + 2) + hg + 8 + 16;
    return n > 0x7fffffffLL ? 0 : (int)n;
}

// offset (in doubles) inside the scratch of the 16 time stamps the update kernel leaves (wall_clock64 ticks, 100 MHz): [0] entry,
// [1] decision taken, [2] backup + clear, [3] assembly, [4] Schur terms subtracted, [5] factorisation, [6] substitutions, [7] depth
// steps, [8] poses, [9] edges recomposed
int sp_window_gn_profile_offset(int n_edges, int n_blocks, int sum_N, int max_N, int n_unknowns) {
    if (n_edges <= 0 || n_blocks <= 0 || sum_N <= 0 || max_N <= 0 || n_unknowns < 0 || n_unknowns > SP_WGN_MAX_Y) return -1;
    const long long ldc = (n_unknowns + 1) & ~1;
    const long long tri = (long long)n_unknowns * (n_unknowns + 1) / 2;
    return (int)((long long)n_edges * (SP_WGN_REC + SP_WGN_SEG * max_N) + (long long)n_edges * (36 + SP_WGN_LOC) + (long long)sum_N * (ldc + 2) +
                 (long long)n_blocks * (tri + 2 * ldc) + n_blocks + 2);
}

// the argument record of one window (what sp_window_gn_step passes its kernels by value and sp_window_gn_run_multi keeps per window in device memory)
static int wgn_fill_args(WGnArgs& w, const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                         const SpWindowBlock* blocks, int n_blocks, int sum_N, int max_N, int n_unknowns, const float* span_partials,
                         const float* seg_partials, double* scratch, SpWindowNode* nodes_backup, float* kld_backup, int flags,
                         float lm_up, float lm_down, float lm_min, float conv_tol, float* state, float* losses, int max_losses) {
    if (!pairs || !edges || !nodes || !blocks || !span_partials || !seg_partials || !scratch || !nodes_backup || !kld_backup || !state ||
        !losses)
        return SP_EINVAL;
    if (n_edges <= 0 || n_nodes <= 0 || n_blocks <= 0 || sum_N <= 0 || max_N <= 0 || max_losses < 0 || n_unknowns < 0) return SP_EINVAL;
    if (n_nodes > SP_WGN_MAX_NODES || n_blocks > SP_WGN_MAX_NODES || n_unknowns > SP_WGN_MAX_Y) return SP_ELIMIT;
    const int stride = SP_WGN_REC + SP_WGN_SEG * max_N;
    w.span_partials = span_partials; w.seg_partials = seg_partials;
    w.pairs = pairs; w.edges = edges; w.n_edges = n_edges; w.nodes = nodes; w.n_nodes = n_nodes; w.blocks = blocks; w.n_blocks = n_blocks;
    w.max_N = max_N; w.scratch = scratch; w.stride = stride;
    w.ldc = (n_unknowns + 1) & ~1;
    w.lds = n_unknowns * (n_unknowns + 1) / 2;
    w.cap_y = n_unknowns;
    w.Ad = scratch + (size_t)n_edges * stride;
    w.loc = w.Ad + (size_t)n_edges * 36;
    w.C = w.loc + (size_t)n_edges * SP_WGN_LOC;
    w.Dinv = w.C + (size_t)sum_N * w.ldc;
    w.Bd = w.Dinv + sum_N;
    w.S = w.Bd + sum_N;
    w.Rhs = w.S + (size_t)n_blocks * w.lds;
    w.cols = reinterpret_cast<int*>(w.Rhs + (size_t)n_blocks * w.ldc);           // n_blocks x ldc ints in n_blocks x ldc / 2 doubles' room ...
    w.nc = reinterpret_cast<int*>(w.Rhs + (size_t)n_blocks * w.ldc * 2);         // ... (a full n_blocks x ldc doubles are reserved)
    w.prof = w.Rhs + (size_t)n_blocks * w.ldc * 2 + n_blocks + 2;
    w.Hg = w.prof + 16;
    w.nodes_backup = nodes_backup; w.kld_backup = kld_backup; w.flags = flags;
    w.lm_up = lm_up; w.lm_down = lm_down; w.lm_min = lm_min; w.conv_tol = conv_tol;
    w.state = state; w.losses = losses; w.max_losses = max_losses;
    return 0;
}

int sp_window_gn_step(const SpPair* pairs, const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes,
                      const SpWindowBlock* blocks, int n_blocks, int sum_N, int max_N, int n_unknowns, const float* span_partials,
                      const float* seg_partials, double* scratch, SpWindowNode* nodes_backup, float* kld_backup, int flags,
                      float lm_up, float lm_down, float lm_min, float conv_tol, float* state, float* losses, int max_losses,
                      void* stream) {
    WGnArgs w;
    if (int rc = wgn_fill_args(w, pairs, edges, n_edges, nodes, n_nodes, blocks, n_blocks, sum_N, max_N, n_unknowns, span_partials, seg_partials, scratch,
                               nodes_backup, kld_backup, flags, lm_up, lm_down, lm_min, conv_tol, state, losses, max_losses))
        return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int stride = w.stride;
    // (no depth moves and a handful of edges -- the tracker: the update kernel reduces them itself, one launch less per iteration)
    const bool inline_reduce = (flags & 5) && n_edges <= SP_WGN_INLINE_EDGES && n_unknowns <= 64 && !getenv("SP_WGN_NO_INLINE");
    if (inline_reduce) w.flags |= SP_WGN_INLINE_REDUCE;
    else {
        hipLaunchKernelGGL(k_window_gn_reduce, dim3(n_edges), dim3(SP_BLOCK), 0, s, pairs, edges, span_partials, seg_partials, scratch, stride, w.Ad, w.loc,
                           (flags & 5) ? 0 : 1);
        SP_CHECK_LAUNCH();
    }
    const int tiles = max(1, (w.lds + SP_BLOCK * SP_WGN_PPT - 1) / (SP_BLOCK * SP_WGN_PPT));
    const WGnArgs* none = nullptr;
    if (!(flags & 5)) {                // (flags bit 0 / bit 2: no depth moves -- every Schur term is zero and the update kernel does not read them)
        hipLaunchKernelGGL(k_window_gn_schur, dim3(n_blocks, tiles), dim3(SP_BLOCK), 0, s, w, none);
        SP_CHECK_LAUNCH();
    }
    if (n_unknowns <= 64) hipLaunchKernelGGL(k_window_gn_update<64>, dim3(1), dim3(wgn_update_threads(64)), 0, s, w, none);
    else if (n_unknowns <= 128) hipLaunchKernelGGL(k_window_gn_update<128>, dim3(1), dim3(wgn_update_threads(128)), 0, s, w, none);
    else if (n_unknowns <= SP_WGN_LDS_Y) hipLaunchKernelGGL(k_window_gn_update<SP_WGN_LDS_Y>, dim3(1), dim3(wgn_update_threads(SP_WGN_LDS_Y)), 0, s, w, none);
    else hipLaunchKernelGGL(k_window_gn_update<0>, dim3(1), dim3(wgn_update_threads(0)), 0, s, w, none);
    SP_CHECK_LAUNCH();
    return 0;
}

/* Up to max_iters iterations (cost pass in mode 2 over the window's work list + sp_window_gn_step) as ONE foreign call: the state is
 * copied to pinned host memory every check_every iterations (synchronising this stream only) and the loop ends once the window froze
 * (converged, odometery.py:907-915's break).  Returns the iterations launched or a negative error. */
int sp_window_gn_run(const SpPair* pairs, const int32_t* chunks, const int32_t* spans, int n_spans, float irls_eps,
                     const SpWindowEdge* edges, int n_edges, SpWindowNode* nodes, int n_nodes, const SpWindowBlock* blocks, int n_blocks,
                     int sum_N, int max_N, int n_unknowns, float* span_partials, float* seg_partials, double* scratch,
                     SpWindowNode* nodes_backup, float* kld_backup, int flags, float lm_up, float lm_down, float lm_min, float conv_tol,
                     float* state, float* losses, int max_losses, int max_iters, int check_every, float* state_host, void* stream) {
    if (!state_host || max_iters < 0 || check_every <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int it = 0;
    while (it < max_iters) {
        const int n = (max_iters - it) < check_every ? (max_iters - it) : check_every;
        for (int k = 0; k < n; ++k, ++it) {
            int rc = sp_pairs_cost(pairs, chunks, spans, n_spans, 2, irls_eps, span_partials, seg_partials, stream);
            if (rc == 0)
                rc = sp_window_gn_step(pairs, edges, n_edges, nodes, n_nodes, blocks, n_blocks, sum_N, max_N, n_unknowns, span_partials,
                                       seg_partials, scratch, nodes_backup, kld_backup, flags, lm_up, lm_down, lm_min, conv_tol, state, losses,
                                       max_losses, stream);
            if (rc != 0) return rc < 0 ? rc : -(1000 + rc);
        }
        hipError_t e = hipMemcpyAsync(state_host, state, SP_WGN_STATE * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return -(1000 + (int)e);
        if (conv_tol > 0.f && static_cast<volatile float*>(state_host)[6] != 0.f) break;
    }
    return it;
}

// ---- S windows per launch (round 6, VERDICT r05 item 4b: config 3's throughput form) ---------------------------------------------------
// gather the 16-float states of the windows into one array (one copy per poll instead of S)
__global__ void k_wgn_gather_states(const WGnArgs* __restrict__ wins, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * SP_WGN_STATE) out[i] = wins[i / SP_WGN_STATE].state[i % SP_WGN_STATE];
}

int sp_window_gn_multi_bytes(void) { return (int)((sizeof(WGnArgs) + sizeof(MultiList) + 15) / 16 * 16); }

int sp_window_gn_run_multi(const SpWindowGn* windows, int n_windows, float irls_eps, int flags, float lm_up, float lm_down, float lm_min,
                           float conv_tol, int max_iters, int check_every, void* args_dev, float* states_dev, float* states_host, void* stream) {
    if (!windows || n_windows <= 0 || n_windows > 65535 || !args_dev || !states_dev || !states_host || max_iters < 0 || check_every <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<WGnArgs> host(n_windows);
    std::vector<MultiList> lists(n_windows);
    int max_edges = 0, max_blocks = 0, max_y = 0, total_blocks = 0;
    for (int i = 0; i < n_windows; ++i) {
        const SpWindowGn& g = windows[i];
        if (!g.chunks || !g.spans || g.n_spans <= 0) return SP_EINVAL;
        if (int rc = wgn_fill_args(host[i], g.pairs, g.edges, g.n_edges, g.nodes, g.n_nodes, g.blocks, g.n_blocks, g.sum_N, g.max_N, g.n_unknowns,
                                   g.span_partials, g.seg_partials, g.scratch, g.nodes_backup, g.kld_backup, flags, lm_up, lm_down, lm_min, conv_tol,
                                   g.state, g.losses, g.max_losses))
            return rc;
        lists[i] = MultiList{g.pairs, g.chunks, g.spans, g.span_partials, g.seg_partials, g.n_spans, total_blocks};
        total_blocks += (g.n_spans + 7) / 8 * 8;
        max_edges = std::max(max_edges, g.n_edges); max_blocks = std::max(max_blocks, g.n_blocks); max_y = std::max(max_y, g.n_unknowns);
    }
    // the update kernel's instantiation is chosen for the LARGEST window; every window must fit the scratch it was sized for (cap_y): a window
    // whose own scratch has no room for the global-memory triangle cannot ride in a batch that needs it
    if (max_y > SP_WGN_LDS_Y)
        for (int i = 0; i < n_windows; ++i) if (windows[i].n_unknowns <= SP_WGN_LDS_Y) return SP_EINVAL;
    WGnArgs* wins = static_cast<WGnArgs*>(args_dev);
    MultiList* lists_dev = reinterpret_cast<MultiList*>(static_cast<char*>(args_dev) + (sizeof(WGnArgs) * (size_t)n_windows + 15) / 16 * 16);
    hipError_t e = hipMemcpyAsync(wins, host.data(), sizeof(WGnArgs) * (size_t)n_windows, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(lists_dev, lists.data(), sizeof(MultiList) * (size_t)n_windows, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return -(1000 + (int)e);
    const int lds = max_y * (max_y + 1) / 2;
    const int tiles = std::max(1, (lds + SP_BLOCK * SP_WGN_PPT - 1) / (SP_BLOCK * SP_WGN_PPT));
    const WGnArgs dummy{};
    int it = 0;
    while (it < max_iters) {
        const int n = (max_iters - it) < check_every ? (max_iters - it) : check_every;
        for (int k = 0; k < n; ++k, ++it) {
            int rc = cost_pairs_multi(lists_dev, n_windows, total_blocks, 2, irls_eps, stream);
            if (rc != 0) return rc < 0 ? rc : -(1000 + rc);
            hipLaunchKernelGGL(k_window_gn_reduce_multi, dim3(max_edges, 1, n_windows), dim3(SP_BLOCK), 0, s, (const WGnArgs*)wins);
            if (!(flags & 5)) hipLaunchKernelGGL(k_window_gn_schur, dim3(max_blocks, tiles, n_windows), dim3(SP_BLOCK), 0, s, dummy, (const WGnArgs*)wins);
            if (max_y <= 64) hipLaunchKernelGGL(k_window_gn_update<64>, dim3(1, 1, n_windows), dim3(wgn_update_threads(64)), 0, s, dummy, (const WGnArgs*)wins);
            else if (max_y <= 128) hipLaunchKernelGGL(k_window_gn_update<128>, dim3(1, 1, n_windows), dim3(wgn_update_threads(128)), 0, s, dummy, (const WGnArgs*)wins);
            else if (max_y <= SP_WGN_LDS_Y) hipLaunchKernelGGL(k_window_gn_update<SP_WGN_LDS_Y>, dim3(1, 1, n_windows), dim3(wgn_update_threads(SP_WGN_LDS_Y)), 0, s, dummy, (const WGnArgs*)wins);
            else hipLaunchKernelGGL(k_window_gn_update<0>, dim3(1, 1, n_windows), dim3(wgn_update_threads(0)), 0, s, dummy, (const WGnArgs*)wins);
            hipError_t el = hipGetLastError();
            if (el != hipSuccess) return -(1000 + (int)el);
        }
        hipLaunchKernelGGL(k_wgn_gather_states, dim3((n_windows * SP_WGN_STATE + 255) / 256), dim3(256), 0, s, (const WGnArgs*)wins, n_windows, states_dev);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(states_host, states_dev, sizeof(float) * SP_WGN_STATE * (size_t)n_windows, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return -(1000 + (int)e);
        if (conv_tol > 0.f) {
            bool all = true;
            for (int i = 0; i < n_windows && all; ++i) all = static_cast<volatile float*>(states_host)[i * SP_WGN_STATE + 6] != 0.f;
            if (all) break;
        }
    }
    return it;
}

}  // extern "C"
