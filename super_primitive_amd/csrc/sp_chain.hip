// sp_chain.hip -- one foreign call per FRAME of the monocular-odometry chain (include/sp_hip.h sp_chain_step; reference:
// odometery/odometery.py:1018-1075 -- track_frame, mapping(mode='supp'), is_kf of a frame that is not a keyframe).
//
// Nothing new is computed here: the stages are the library's own entry points (sp_blur_decimate, sp_pack_rgb, sp_window_compose,
// sp_window_gn_run, sp_renormalise_se3, sp_depth_splat, sp_kf_criterion) strung together on one stream, with four small kernels in
// place of what the Python loop did on the host between them -- overwrite a node's pose from a device buffer, reset / re-phase the LM
// state, read the node back out, inv(A) B of two poses.  State stays on the device; the host sees the LM states the run loop polls
// and the four floats of the keyframe criterion.
#include <hip/hip_runtime.h>

#include "../../include/sp_hip.h"
#include "sp_device.h"

#define SP_CHAIN_STATE 16   // floats of a window's LM state (sp_window_gn_step)

static_assert(sizeof(SpChainPhase) == 16 && sizeof(SpChainWindow) == 680 && sizeof(SpChainTarget) == 56 && sizeof(SpChainStep) == 1752,
              "SpChain* layouts are part of the ABI (super_primitive_amd/_lib.py mirrors them)");

namespace {

// the node's pose / affine pair from device buffers; tangent and Adam moments cleared (optim/window.py set_nodes)
__global__ void k_chain_set_nodes(SpWindowNode* __restrict__ nodes, int n, int node0, const float* __restrict__ pose0, const float* __restrict__ aff0,
                                  int node1, const float* __restrict__ pose1, const float* __restrict__ aff1) {
    const int which = threadIdx.x >> 5, t = threadIdx.x & 31;
    if (which >= n) return;
    SpWindowNode& nd = nodes[which ? node1 : node0];
    const float* pose = which ? pose1 : pose0;
    const float* aff = which ? aff1 : aff0;
    if (t < 16) nd.T[t] = pose[t];
    if (t < 6) { nd.a[t] = 0.f; nd.m[t] = 0.f; nd.v[t] = 0.f; }
    if (t < 2) {
        if (aff) nd.aff[t] = aff[t];
        nd.aff_m[t] = 0.f; nd.aff_v[t] = 0.f;
    }
}

// fresh = 1: the state of a new optimisation {lambda, no accepted point, ...}; 0: a new phase of the schedule (optim/window.py
// begin_gn_phase: the accept and convergence tests start afresh, lambda and the iteration count carry over)
__global__ void k_chain_state(float* __restrict__ st, float lam, int fresh) {
    const int t = threadIdx.x;
    if (t >= SP_CHAIN_STATE) return;
    if (fresh) st[t] = t == 0 ? lam : (t == 1 ? -1.f : 0.f);
    else if (t == 1) st[t] = -1.f;
    else if (t == 4 || t == 6) st[t] = 0.f;
}

__global__ void k_chain_read_node(const SpWindowNode* __restrict__ nodes, int node, float* __restrict__ out_pose, float* __restrict__ out_aff) {
    const int t = threadIdx.x;
    if (t < 16) out_pose[t] = nodes[node].T[t];
    if (out_aff && t < 2) out_aff[t] = nodes[node].aff[t];
}

// rel = inv(A) B for rigid A (lie/lie_algebra.py invertSE3 followed by a matrix product)
__global__ void k_chain_rel_pose(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ rel) {
    const int t = threadIdx.x;
    if (t >= 16) return;
    const int r = t >> 2, c = t & 3;
    float inv[4];                                   // row r of inv(A)
    if (r < 3) {
        inv[0] = A[0 * 4 + r]; inv[1] = A[1 * 4 + r]; inv[2] = A[2 * 4 + r];
        inv[3] = -(inv[0] * A[3] + inv[1] * A[7] + inv[2] * A[11]);
    } else {
        inv[0] = inv[1] = inv[2] = 0.f; inv[3] = 1.f;
    }
    rel[t] = inv[0] * B[c] + inv[1] * B[4 + c] + inv[2] * B[8 + c] + inv[3] * B[12 + c];
}

int first_level(const SpChainWindow& w) {
    for (int l = 0; l < SP_CHAIN_LEVELS; ++l) if (w.gn[l].pairs) return l;
    return -1;
}

// fresh LM state, then the phases of the window's schedule.  The loop of sp_window_gn_run with the polls that decide nothing left out: a look
// at the state only BETWEEN the iterations of a phase (the last iteration of a phase is followed by the next phase whatever the state says), and
// ONE asynchronous copy of the final state at the end (*state_pending: the caller synchronises before it reads the iteration count).
int run_phases(const SpChainWindow& w, void* stream, bool* state_pending) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int l0 = first_level(w);
    if (l0 < 0 || !w.state_host || w.n_phases < 0 || w.n_phases > SP_CHAIN_PHASES || w.check_every <= 0) return SP_EINVAL;
    float* state = w.gn[l0].state;
    hipLaunchKernelGGL(k_chain_state, dim3(1), dim3(64), 0, s, state, w.lam0, 1);
    for (int p = 0; p < w.n_phases; ++p) {
        const SpChainPhase& ph = w.phase[p];
        if (ph.level < 0 || ph.level >= SP_CHAIN_LEVELS || !w.gn[ph.level].pairs) return SP_EINVAL;
        if (ph.max_iters <= 0) continue;
        const SpWindowGn& g = w.gn[ph.level];
        hipLaunchKernelGGL(k_chain_state, dim3(1), dim3(64), 0, s, state, 0.f, 0);
        int it = 0;
        int look = (w.check_first > 0 && ph.conv_tol > 0.f) ? w.check_first : w.check_every;
        while (it < ph.max_iters) {
            const int n = (ph.max_iters - it) < look ? (ph.max_iters - it) : look;
            for (int k = 0; k < n; ++k, ++it) {
                int rc = sp_pairs_cost(g.pairs, g.chunks, g.spans, g.n_spans, 2, ph.irls_eps, g.span_partials, g.seg_partials, stream);
                if (rc == 0)
                    rc = sp_window_gn_step(g.pairs, g.edges, g.n_edges, g.nodes, g.n_nodes, g.blocks, g.n_blocks, g.sum_N, g.max_N, g.n_unknowns, g.span_partials,
                                           g.seg_partials, g.scratch, g.nodes_backup, g.kld_backup, w.flags, w.lm_up, w.lm_down, w.lm_min, ph.conv_tol, g.state,
                                           g.losses, g.max_losses, stream);
                if (rc != 0) return rc < 0 ? rc : -(1000 + rc);
            }
            if (it >= ph.max_iters || !(ph.conv_tol > 0.f)) continue;          // (nothing to decide: the phase is over, or it has no convergence test)
            hipError_t e = hipMemcpyAsync(w.state_host, state, SP_CHAIN_STATE * sizeof(float), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return -(1000 + (int)e);
            if (static_cast<volatile float*>(w.state_host)[6] != 0.f) break;
            look = w.check_every;
        }
    }
    const hipError_t e = hipMemcpyAsync(w.state_host, state, SP_CHAIN_STATE * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return -(1000 + (int)e);
    *state_pending = true;
    return 0;
}

}  // namespace

extern "C" int sp_chain_step(SpChainStep* st, void* stream) {
    if (!st) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int stages = st->stages;
    if (stages & ~(SP_CHAIN_TRACK | SP_CHAIN_SUPP | SP_CHAIN_CRITERION)) return SP_EINVAL;
    if (st->n_levels < 1 || st->n_levels > SP_CHAIN_LEVELS || st->H <= 0 || st->W <= 0) return SP_EINVAL;
    bool track_pending = false, supp_pending = false, synced = false;
    int Hl[SP_CHAIN_LEVELS], Wl[SP_CHAIN_LEVELS];
    Hl[0] = st->H; Wl[0] = st->W;
    for (int l = 1; l < SP_CHAIN_LEVELS; ++l) { Hl[l] = (Hl[l - 1] + 1) / 2; Wl[l] = (Wl[l - 1] + 1) / 2; }

    if (stages & SP_CHAIN_TRACK) {
        const SpChainWindow& w = st->track;
        const SpChainTarget& tg = st->track_target;
        const int l0 = first_level(w);
        if (l0 < 0 || !st->image || !tg.pose || !st->out_pose || tg.node < 0 || tg.node >= w.gn[l0].n_nodes) return SP_EINVAL;
        // the frame's pyramid (image/gaussian_pyramid.py:53-85), every level the tracker matches at packed into its target buffers
        const float* prev = st->image;
        for (int l = 0; l < st->n_levels; ++l) {
            if (l > 0) {
                if (!st->level[l]) return SP_EINVAL;
                if (int rc = sp_blur_decimate(prev, 3, Hl[l - 1], Wl[l - 1], st->level[l], stream)) return rc;
                prev = st->level[l];
            }
            if (tg.packed[l])
                if (int rc = sp_pack_rgb(prev, 1, Hl[l], Wl[l], tg.packed[l], stream)) return rc;
        }
        const SpWindowGn& g = w.gn[l0];
        hipLaunchKernelGGL(k_chain_set_nodes, dim3(1), dim3(64), 0, s, g.nodes, 1, tg.node, tg.pose, tg.aff, 0, (const float*)nullptr, (const float*)nullptr);
        SP_CHECK_LAUNCH();
        if (int rc = sp_window_compose(g.pairs, g.edges, g.n_edges, g.nodes, g.n_nodes, stream)) return rc;
        if (int rc = run_phases(w, stream, &track_pending)) return rc;
        hipLaunchKernelGGL(k_chain_read_node, dim3(1), dim3(64), 0, s, (const SpWindowNode*)g.nodes, tg.node, st->out_pose, st->out_aff);
        SP_CHECK_LAUNCH();
        if (int rc = sp_renormalise_se3(st->out_pose, 1, stream)) return rc;
    }

    if (stages & SP_CHAIN_SUPP) {
        const SpChainWindow& w = st->supp;
        const int l0 = first_level(w);
        if (l0 < 0) return SP_EINVAL;
        const SpWindowGn& g = w.gn[l0];
        const SpChainTarget& a = st->supp_target[0];
        const SpChainTarget& b = st->supp_target[1];
        if (!a.pose || !b.pose || a.node < 0 || b.node < 0 || a.node >= g.n_nodes || b.node >= g.n_nodes) return SP_EINVAL;
        for (int l = 0; l < SP_CHAIN_LEVELS; ++l) {
            if (!w.gn[l].pairs) continue;
            const size_t bytes = sizeof(float) * 3 * (size_t)Hl[l] * Wl[l];
            hipError_t e = hipSuccess;
            if (st->supp_images & 1) {
                if (!a.packed[l] || !b.packed[l]) return SP_EINVAL;
                e = hipMemcpyAsync(a.packed[l], b.packed[l], bytes, hipMemcpyDeviceToDevice, s);
            }
            if (e == hipSuccess && (st->supp_images & 2)) {
                if (!b.packed[l] || !st->track_target.packed[l]) return SP_EINVAL;
                e = hipMemcpyAsync(b.packed[l], st->track_target.packed[l], bytes, hipMemcpyDeviceToDevice, s);
            }
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(k_chain_set_nodes, dim3(1), dim3(64), 0, s, g.nodes, 2, a.node, a.pose, a.aff, b.node, b.pose, b.aff);
        SP_CHECK_LAUNCH();
        if (int rc = sp_window_compose(g.pairs, g.edges, g.n_edges, g.nodes, g.n_nodes, stream)) return rc;
        if (int rc = run_phases(w, stream, &supp_pending)) return rc;
        if (st->kld_n > 0) {
            if (!st->kld_src || !st->kld_dst) return SP_EINVAL;
            hipError_t e = hipMemcpyAsync(st->kld_dst, st->kld_src, sizeof(float) * (size_t)st->kld_n, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        }
    }

    if (stages & SP_CHAIN_CRITERION) {
        if (!st->out_pose || !st->kf_pose || !st->rel_pose || !st->crit || !st->crit_ws || !st->crit_host || !st->depth_out || !st->keys) return SP_EINVAL;
        hipLaunchKernelGGL(k_chain_rel_pose, dim3(1), dim3(64), 0, s, (const float*)st->out_pose, st->kf_pose, st->rel_pose);
        SP_CHECK_LAUNCH();
        if (int rc = sp_depth_splat(st->pix, st->baseL, st->seg_off, st->kp_L, st->kld, st->N, st->P, st->H, st->W, st->K, st->rel_pose, st->keys,
                                    st->depth_out, stream))
            return rc;
        if (int rc = sp_kf_criterion_ws(st->depth_out, st->H * st->W, st->valid_thresh, st->out_pose, st->kf_pose, st->crit_ws, st->crit, stream)) return rc;
        hipError_t e = hipMemcpyAsync(st->crit_host, st->crit, 4 * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        synced = true;
    }
    // the windows' final LM states were copied asynchronously: the iteration counts are read behind a synchronisation
    if ((track_pending || supp_pending) && !synced) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
    }
    if (track_pending) st->track_iters = (int)static_cast<volatile float*>(st->track.state_host)[5];
    if (supp_pending) st->supp_iters = (int)static_cast<volatile float*>(st->supp.state_host)[5];
    return 0;
}
