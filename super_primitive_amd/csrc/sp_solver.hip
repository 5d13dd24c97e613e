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

static_assert(sizeof(SpPhase) == 64 && sizeof(SpSchedule) == 800 && sizeof(SpVerdict) == 104 && sizeof(SpQueue) == 296 && sizeof(SpPair) == 136,
              "SpSchedule / SpVerdict / SpQueue / SpPair are part of the ABI");

// per-pair schedules: the pair's current phase selects the level descriptors, the partial records of that level's work list,
// the convergence threshold and the iteration budget.
// THE VERDICT (SpVerdict, v.status != NULL): the workgroup that takes a pair out of its last phase looks at what it ended with --
// finiteness, how the last phase ended, the largest log-depth excursion from the initial values, the final cost, the valid
// fraction -- and either files status / diag under the pair's index or, once per pair, puts the pair back to its initial values
// and restarts it at the schedule's retry_entry (a second attempt on a different phase list) in the slot it already has.
// SLOT-LEVEL CONTINUOUS BATCHING (SpQueue, q.n_queue > 0): the launch's "pairs" are SLOTS of a resident set; the slot whose pair just
// finished files the pair's result under the pair's own index, takes the next waiting pair off the queue (one atomic per finished
// pair), copies that pair's descriptor of every phase over its own and starts it at the entry phase -- in this very launch, so the
// next cost pass already works on the new pair and the resident set stays full until the queue is empty.  Pairs never interact and a
// descriptor carries the pair's own span range (the cost pass of a queue run is launched over virtual spans): each pair's result
// is bitwise what it is with all pairs resident.
__global__ __launch_bounds__(SP_BLOCK) void k_pairs_gn_sched(SpSchedule sched, GnArgs h, SpQueue q, SpVerdict v, const int32_t* __restrict__ active) {
    const int slot = active ? active[blockIdx.x] : blockIdx.x;        // (the tail of a queue run is launched over the listed slots only)
    const int ph = h.phase[slot];
    if (ph >= sched.n_phases || ph < 0) return;           // finished, and the queue was empty when it did
    if ((h.idle_mask >> ph) & 1u) return;                 // its work list was not launched this round (a third attempt waiting for the next poll)
    if (v.evals && threadIdx.x == 0) v.evals[(size_t)(q.n_queue > 0 ? q.slot_pair[slot] : slot) * SP_MAX_PHASES + ph] += 1;
    {
        const SpPhase& s = sched.phase[ph];
        h.conv_tol = s.conv_tol;
        h.max_iters = s.max_iters;
        h.pose_only = s.flags & SP_PHASE_POSE_ONLY;
        h.next_phase = s.next > 0 ? s.next : ph + 1;
        h.depth_damp = 0.125f * (float)((s.flags >> SP_PHASE_DEPTH_DAMP_SHIFT) & 0xff);
        h.adam_lr_pose = sched.adam_lr_pose; h.adam_lr_kld = sched.adam_lr_kld; h.adam_state = sched.adam_state;
        h.predicted_exit = s.flags & SP_PHASE_PREDICTED_EXIT;
        if ((s.flags & SP_PHASE_ADAM) && sched.adam_state) solve_adam_sched(s.pairs, slot, s.span_partials, s.seg_partials, h);
        else solve_gn(s.pairs, slot, s.span_partials, s.seg_partials, h);
    }
    if (q.n_queue <= 0 && !v.status) return;
    __shared__ int next_s, retry_s;
    __shared__ float dmax_s[SP_WAVES];
    __shared__ int bad_s[SP_WAVES];
    __shared__ float segc_s[SP_VERDICT_SEGMENTS];      // cost per segment (the within-pair test); < 0 = too few valid points to say
    __shared__ float seg_med_s, seg_max_s[SP_WAVES];
    __shared__ int seg_n_s[SP_WAVES];
    __syncthreads();                            // every write of solve_gn (thread 0's phase update included) is visible
    const int pid = q.n_queue > 0 ? q.slot_pair[slot] : slot;
    float* ls = h.lm_state + (size_t)slot * SP_LM_STRIDE;
    if (v.diag && threadIdx.x == 0) {           // the cost a pair's attempt starts from (diag[5], zeroed by the caller)
        float* first = v.diag + (size_t)pid * SP_DIAG_FLOATS + 5;
        if (*first == 0.f) *first = h.costs[slot];
    }
    if (h.phase[slot] < sched.n_phases) return;
    // ---- the pair has just left its last phase ----
    if (v.status) {
        const SpPair& pr = sched.phase[ph].pairs[slot];
        const float* kld0 = v.kld0 + (pr.kld - v.kld_base);
        const float* pose0 = v.pose0 + (pr.pose - v.pose_base);
        float dmax = 0.f;
        int bad = 0;
        for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) {
            const float d = fabsf(pr.kld[n] - kld0[n]);
            if (!(d <= 3.0e38f)) bad = 1;       // (NaN and infinity both fail the comparison)
            else dmax = fmaxf(dmax, d);
        }
        if (threadIdx.x < 12 && !(fabsf(pr.pose[threadIdx.x]) <= 3.0e38f)) bad = 1;
        // THE WITHIN-PAIR TEST (ABI 13): mean |r| of every segment at the last evaluated point, from the segment records of the last phase's
        // cost pass ([8] sum |r|, [9] valid points) -- a pair that sits in the second solution of a near-plane's homography explains some
        // segments and not others (worst segment 3-40 x the median one; a converged pair's lie within 3 x), whatever the other pairs of
        // the batch do: the one test a batch of ONE has against a well-behaved local minimum
        const int n_seg = min(pr.N, SP_VERDICT_SEGMENTS);
        float smax = 0.f;
        int sn = 0;
        {
            const float* sp = sched.phase[ph].seg_partials + (size_t)pr.rec0 * SP_GN_SEG_FLOATS;
            for (int n = threadIdx.x; n < n_seg; n += SP_BLOCK) {
                double a = 0.0, c = 0.0;
                for (int t = pr.seg_tile_off[n]; t < pr.seg_tile_off[n + 1]; ++t) {
                    a += (double)sp[(size_t)t * SP_GN_SEG_FLOATS + 8];
                    c += (double)sp[(size_t)t * SP_GN_SEG_FLOATS + 9];
                }
                const float sc = c >= (double)SP_VERDICT_SEGMENT_POINTS ? (float)(a / (3.0 * c)) : -1.f;
                segc_s[n] = sc;
                if (sc >= 0.f) { smax = fmaxf(smax, sc); ++sn; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); bad |= __shfl_xor(bad, o, 64);
            smax = fmaxf(smax, __shfl_xor(smax, o, 64)); sn += __shfl_xor(sn, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { dmax_s[threadIdx.x >> 6] = dmax; bad_s[threadIdx.x >> 6] = bad; seg_max_s[threadIdx.x >> 6] = smax; seg_n_s[threadIdx.x >> 6] = sn; }
        if (threadIdx.x == 0) seg_med_s = 0.f;
        __syncthreads();
        {
            // lower median of the judged segments by rank counting (N^2 / 256 comparisons per thread, once per pair)
            const int n_judged = seg_n_s[0] + seg_n_s[1] + seg_n_s[2] + seg_n_s[3];
            const int want = (n_judged - 1) >> 1;
            for (int n = threadIdx.x; n < n_seg && n_judged >= SP_VERDICT_MIN_SEGMENTS; n += SP_BLOCK) {
                const float x = segc_s[n];
                if (x < 0.f) continue;
                int rank = 0;
                for (int j = 0; j < n_seg; ++j) {
                    const float y = segc_s[j];
                    rank += (y >= 0.f && (y < x || (y == x && j < n))) ? 1 : 0;
                }
                if (rank == want) seg_med_s = x;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            dmax = fmaxf(fmaxf(dmax_s[0], dmax_s[1]), fmaxf(dmax_s[2], dmax_s[3]));
            bad = bad_s[0] | bad_s[1] | bad_s[2] | bad_s[3];
            const float cost = ls[5], first = v.diag ? v.diag[(size_t)pid * SP_DIAG_FLOATS + 5] : 0.f;
            int st = 0;
            if (bad || !(fabsf(cost) <= 3.0e38f)) st |= SP_STATUS_NONFINITE;
            if (ls[7] > 0.f) st |= SP_STATUS_LAST_CAP;
            if (v.kld_bound > 0.f && dmax > v.kld_bound) st |= SP_STATUS_DEPTH_RANGE;
            if ((v.cost_bound > 0.f && cost > v.cost_bound) || (v.cost_ratio > 0.f && first > 0.f && cost > v.cost_ratio * first)) st |= SP_STATUS_COST;
            if (v.valid_min > 0.f && ls[6] < v.valid_min) st |= SP_STATUS_VALID;
            const float seg_med = seg_med_s, seg_max = fmaxf(fmaxf(seg_max_s[0], seg_max_s[1]), fmaxf(seg_max_s[2], seg_max_s[3]));
            if (seg_med > 0.f && ((v.seg_max_ratio > 0.f && seg_max > v.seg_max_ratio * seg_med) || (v.seg_mean_ratio > 0.f && cost > v.seg_mean_ratio * seg_med) ||
                                  (v.seg_product > 0.f && (cost / seg_med - 1.f) * (seg_max / seg_med) > v.seg_product)))
                st |= SP_STATUS_SEGMENTS;
            // attempts[pair]: 0 = first attempt, 1 = restarted at retry_entry, 2 = restarted at retry2_entry (the last one there is)
            const int attempt = v.attempts ? v.attempts[pid] : 2;
            const bool failed = (st & v.retry_mask) != 0;
            const bool has1 = sched.retry_entry >= 0 && sched.retry_entry < sched.n_phases, has2 = sched.retry2_entry >= 0 && sched.retry2_entry < sched.n_phases;
            int again = 0;
            if (failed && attempt == 0 && has1) again = 1;
            else if (failed && attempt <= 1 && has2) again = 2;
            retry_s = again;
            if (again) {
                v.attempts[pid] = again;
                h.phase[slot] = again == 1 ? sched.retry_entry : sched.retry2_entry; h.iters[slot] = 0;
                ls[0] = v.lam0; ls[1] = -1.f; ls[4] = 0.f; ls[7] = 0.f;       // (the iteration counts [2], [3] run on over all attempts)
                if (v.diag) v.diag[(size_t)pid * SP_DIAG_FLOATS + 5] = 0.f;
            } else {
                v.status[pid] = st | (attempt > 0 && v.attempts ? SP_STATUS_RETRIED : 0) | (attempt > 1 && v.attempts ? SP_STATUS_ADAM : 0);
                if (v.diag) {
                    float* d = v.diag + (size_t)pid * SP_DIAG_FLOATS;
                    d[0] = cost; d[1] = dmax; d[2] = ls[6]; d[3] = fabsf(ls[7]); d[4] = (float)(v.attempts ? attempt + 1 : 1); d[6] = seg_med; d[7] = seg_max;
                }
            }
        }
        __syncthreads();
        if (retry_s) {
            for (int n = threadIdx.x; n < pr.N; n += SP_BLOCK) pr.kld[n] = kld0[n];
            if (threadIdx.x < 16) pr.pose[threadIdx.x] = pose0[threadIdx.x];
            if (sched.adam_state) {             // the moments of SP_PHASE_ADAM phases start every attempt at zero
                float* st = sched.adam_state + (size_t)slot * (2 + 2 * (h.max_N + 8));
                for (int i = threadIdx.x; i < 2 + 2 * (h.max_N + 8); i += SP_BLOCK) st[i] = 0.f;
            }
            return;                             // the pair keeps its slot
        }
    }
    if (q.n_queue <= 0) return;
    if (threadIdx.x < SP_LM_STRIDE) q.q_lm[(size_t)pid * SP_LM_STRIDE + threadIdx.x] = ls[threadIdx.x];
    if (threadIdx.x == 0) {
        q.q_costs[pid] = h.costs[slot];
        next_s = atomicAdd(q.head, 1);
    }
    __syncthreads();
    const int next = next_s;
    if (next >= q.n_queue) return;              // nothing is waiting: the slot stays finished
    {
        // the waiting pair's descriptor of every phase over the slot's (phases of one level / lattice share a descriptor array: they
        // write the same values) -- whole records: the pair brings its own span range and record offsets
        constexpr int WORDS = (int)(sizeof(SpPair) / sizeof(uint32_t));
        for (int i = threadIdx.x; i < sched.n_phases * WORDS; i += SP_BLOCK) {
            const int p = i / WORDS, k = i - p * WORDS;
            reinterpret_cast<uint32_t*>(q.slot_pairs[p] + slot)[k] = reinterpret_cast<const uint32_t*>(q.qpairs[p] + next)[k];
        }
    }
    if (sched.adam_state) {
        float* st = sched.adam_state + (size_t)slot * (2 + 2 * (h.max_N + 8));
        for (int i = threadIdx.x; i < 2 + 2 * (h.max_N + 8); i += SP_BLOCK) st[i] = 0.f;
    }
    if (threadIdx.x == 0) {
        h.phase[slot] = sched.entry; h.iters[slot] = 0;
        ls[0] = q.lam0; ls[1] = -1.f; ls[2] = 0.f; ls[3] = 0.f; ls[4] = 0.f; ls[5] = 0.f; ls[6] = 0.f; ls[7] = 0.f;
        q.slot_pair[slot] = next;
    }
}

// pairs still in a slot when a queue run ends on its round limit: SP_STATUS_UNFINISHED under the pair's index
__global__ void k_mark_unfinished(const int32_t* __restrict__ phase, const int32_t* __restrict__ slot_pair, int n_slots, int n_phases,
                                  int32_t* __restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots && phase[i] < n_phases) status[slot_pair ? slot_pair[i] : i] = SP_STATUS_UNFINISHED;
}

// min over the per-pair phases (one workgroup): what the host polls to end a scheduled run.  out[0] = min phase, out[1] = queue head
// (queue runs), out[2] = 1 when some unfinished pair has an attempt left (verdict runs with a second / third attempt: such a pair
// may restart at retry_entry / retry2_entry at any time, so no work list can be left out while one exists)
__global__ __launch_bounds__(SP_BLOCK) void k_phase_min(const int32_t* __restrict__ phase, int n, int32_t* __restrict__ out,
                                                        const int32_t* __restrict__ head, const int32_t* __restrict__ attempts,
                                                        const int32_t* __restrict__ slot_pair, int n_phases, int last_attempt,
                                                        int32_t* __restrict__ active_out) {
    __shared__ int part[SP_WAVES], first[SP_WAVES];
    __shared__ int n_active_s;
    __shared__ unsigned occ_s;
    if (threadIdx.x == 0) { n_active_s = 0; occ_s = 0u; }
    __syncthreads();
    int m = 0x7fffffff, f = 0;
    unsigned occ = 0u;
    for (int i = threadIdx.x; i < n; i += SP_BLOCK) {
        const int ph = phase[i];
        m = min(m, ph);
        // out[3] / active_out: how many slots still work on a pair, and which (in no particular order: a pair's result does not depend
        // on where it runs) -- the host launches the tail of a queue run over those alone
        if (ph >= 0 && ph < n_phases) { const int at = atomicAdd(&n_active_s, 1); if (active_out) active_out[at] = i; occ |= 1u << ph; }
        if (attempts && ph < n_phases && attempts[slot_pair ? slot_pair[i] : i] < last_attempt) f = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = min(m, __shfl_xor(m, o, 64)); f |= __shfl_xor(f, o, 64); }
    if (occ) atomicOr(&occ_s, occ);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = m; first[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[4] = (int32_t)occ_s;          // which phases hold a pair: the host launches the third attempt's own work lists only while one is in them
        out[0] = min(min(part[0], part[1]), min(part[2], part[3]));
        out[1] = head ? *head : 0;
        out[2] = first[0] | first[1] | first[2] | first[3];
        out[3] = n_active_s;
    }
}

// lie/lie_algebra.py:41-119, one thread per matrix (body: renormalise_rotation in sp_solve_device.h)
__global__ void k_renormalise(float* __restrict__ T, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    renormalise_rotation(T + 16 * (size_t)i);
}

// ---------------------------------------------------------------------------------------------------
// Pose parameter of the reference's drivers: T = Exp(a) * X with a = [tau, phi] (lietorch LieGroupParameter.retr()
// followed by .matrix(), odometery/two_frame_sfm.py:83-84, odometery/odometery.py:224-228).  As eager torch code
// the exponential and its autograd graph are ~100 tiny launches (1.5 ms per iteration measured); here forward
// and backward are one launch each.  The backward pass differentiates the very same code with forward-mode dual
// numbers (6 tangents at once), so value and derivative cannot drift apart, also at a = 0.
// ---------------------------------------------------------------------------------------------------
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

int sp_pairs_gn_step_conv(const SpPair* pairs, int n_pairs, int max_N, const float* partials, const float* seg_partials,
                          float lm_up, float lm_down, float lm_min, float* lm_state, float* backup, float* costs, float conv_tol,
                          int32_t* done, void* stream) {
    if (!pairs || !partials || !seg_partials || !lm_state || !backup || !costs || n_pairs <= 0 || max_N <= 0) return SP_EINVAL;
    if (conv_tol > 0.f && !done) return SP_EINVAL;
    hipLaunchKernelGGL(k_pairs_gn, dim3(n_pairs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), pairs, partials,
                       seg_partials, GnArgs{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs, conv_tol, done, nullptr, nullptr, 0, 0});
    SP_CHECK_LAUNCH();
    return 0;
}

// phases that are SP_PHASE_ADAM: their (fine-grained) work lists are launched only while a pair is in one -- a pair that restarts into its
// third attempt between two polls sits out the rounds until the next poll sees it (at most check_every - 1 of its 1500)
static uint32_t adam_phases(const SpSchedule* sched) {
    uint32_t m = 0;
    for (int p = 0; p < sched->n_phases; ++p) if (sched->phase[p].flags & SP_PHASE_ADAM) m |= 1u << p;
    return m;
}

// what every scheduled entry point checks of an SpSchedule (and of the verdict that goes with it)
static int check_schedule(const SpSchedule* sched, const SpVerdict* v) {
    if (!sched || sched->n_phases <= 0 || sched->n_phases > SP_MAX_PHASES) return SP_EINVAL;
    if (sched->entry < 0 || sched->entry >= sched->n_phases || sched->retry_entry < -1 || sched->retry_entry >= sched->n_phases) return SP_EINVAL;
    if (sched->retry2_entry < -1 || sched->retry2_entry >= sched->n_phases) return SP_EINVAL;
    for (int p = 0; p < sched->n_phases; ++p) {
        const SpPhase& ph = sched->phase[p];
        if (!ph.pairs || !ph.span_partials || !ph.seg_partials || ph.max_iters <= 0) return SP_EINVAL;
        if ((ph.flags & SP_PHASE_ADAM) && (!sched->adam_state || !(sched->adam_lr_pose > 0.f) || !(sched->adam_lr_kld > 0.f))) return SP_EINVAL;
        if (ph.next != 0 && (ph.next <= p || ph.next > sched->n_phases)) return SP_EINVAL;      // pairs only move forward
    }
    if (v && v->status) {
        if (!v->pose0 || !v->kld0 || !v->pose_base || !v->kld_base) return SP_EINVAL;
        if ((sched->retry_entry >= 0 || sched->retry2_entry >= 0) && v->retry_mask != 0 && !v->attempts) return SP_EINVAL;
    }
    return 0;
}

int sp_pairs_schedule_gn_step(const SpSchedule* sched, int n_pairs, int max_N, float lm_up, float lm_down, float lm_min,
                              float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters, const SpVerdict* verdict,
                              void* stream) {
    if (!sched || !lm_state || !backup || !costs || !phase || !iters || n_pairs <= 0 || max_N <= 0) return SP_EINVAL;
    if (int rc = check_schedule(sched, verdict)) return rc;
    hipLaunchKernelGGL(k_pairs_gn_sched, dim3(n_pairs), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), *sched,
                       GnArgs{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs, 0.f, nullptr, phase, iters, 0, 0, 0}, SpQueue{},
                       verdict ? *verdict : SpVerdict{}, (const int32_t*)nullptr);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_pairs_schedule_run_queue(const SpSchedule* sched, const SpQueue* queue, int n_slots, int max_N, float lm_up, float lm_down,
                                float lm_min, float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters,
                                int check_every, int max_rounds, int32_t* flag_dev, int32_t* flag_host, const SpVerdict* verdict, void* stream) {
    if (!sched || !queue || !phase || !iters || !flag_dev || !flag_host || !lm_state || !backup || !costs) return SP_EINVAL;
    if (check_every <= 0 || max_rounds < 0 || n_slots <= 0 || max_N <= 0 || queue->n_queue < n_slots) return SP_EINVAL;
    if (int rc = check_schedule(sched, verdict)) return rc;
    if (!queue->head || !queue->slot_pair || !queue->q_costs || !queue->q_lm) return SP_EINVAL;
    for (int p = 0; p < sched->n_phases; ++p) {
        const SpPhase& ph = sched->phase[p];
        if (!queue->qpairs[p] || queue->slot_pairs[p] != ph.pairs || queue->max_spans[p] < 0) return SP_EINVAL;
        if ((ph.flags & SP_PHASE_WAVE_SPANS) && (queue->max_spans[p] & 3)) return SP_EINVAL;
        if ((long long)queue->max_spans[p] * n_slots > 0x7fffffffLL) return SP_ELIMIT;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SpVerdict vd = verdict ? *verdict : SpVerdict{};
    // a second attempt restarts a pair at retry_entry at any time: no work list can be left out while a first attempt is still running
    const bool may_retry = vd.status && vd.attempts && vd.retry_mask != 0 && (sched->retry_entry >= 0 || sched->retry2_entry >= 0);
    const int last_attempt = sched->retry2_entry >= 0 ? 2 : 1;
    int it = 0;
    int reached = 0;                  // a phase every slot has passed -- only meaningful once the queue is empty (refilled slots restart at the entry)
    int min_phase = 0;
    // THE TAIL (round 6): once the queue is empty and few slots still work on a pair -- a third attempt runs 1500 rounds on its own --
    // both launches of a round go over the slots k_phase_min listed at the last poll instead of over all of them (a slot that finishes
    // between two polls returns at once, none can become active again): a round of one pair costs its own kernels' latency, not the
    // dispatch of n_slots x max_spans workgroups that find nothing to do
    int n_active = 0;
    const uint32_t adam_mask = adam_phases(sched);
    uint32_t occupied = 1u << sched->entry;           // (what the last poll saw; before the first one every slot is at the entry)
    uint32_t seen = 0;                                // ... without the entry's bit: where the busy slots really are
    while (it < max_rounds) {
        const bool tail = queue->active && n_active > 0 && 8 * n_active <= n_slots;
        // a tail of third attempts only -- every busy slot in an SP_PHASE_ADAM phase, hundreds of rounds from leaving it: the host looks in
        // four times less often (a poll is a fifth of such a round's 25 us; finishing is noticed at most a dozen empty rounds late)
        const int every = (tail && seen != 0 && (seen & ~adam_mask) == 0) ? 4 * check_every : check_every;
        const int n = (max_rounds - it) < every ? (max_rounds - it) : every;
        const uint32_t idle = adam_mask & ~occupied;
        GnArgs ga{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs, 0.f, nullptr, phase, iters, 0, 0, 0};
        ga.idle_mask = idle;
        for (int k = 0; k < n; ++k, ++it) {
            int rc = schedule_cost_from(sched, phase, stream, reached, queue, n_slots, tail ? queue->active : nullptr, tail ? n_active : 0, idle);
            if (rc != 0) return rc < 0 ? rc : -(1000 + rc);
            hipLaunchKernelGGL(k_pairs_gn_sched, dim3(tail ? n_active : n_slots), dim3(SP_BLOCK), 0, s, *sched, ga, *queue, vd,
                               tail ? (const int32_t*)queue->active : (const int32_t*)nullptr);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return -(1000 + (int)e);
        }
        hipLaunchKernelGGL(k_phase_min, dim3(1), dim3(SP_BLOCK), 0, s, phase, n_slots, flag_dev, queue->head, may_retry ? vd.attempts : nullptr,
                           queue->slot_pair, sched->n_phases, last_attempt, queue->active);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(flag_host, flag_dev, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return -(1000 + (int)e);
        min_phase = static_cast<volatile int32_t*>(flag_host)[0];
        seen = (uint32_t)static_cast<volatile int32_t*>(flag_host)[4];
        occupied = seen | (1u << sched->entry);           // (a refilled slot starts at the entry)
        const int head = static_cast<volatile int32_t*>(flag_host)[1];
        n_active = head >= queue->n_queue ? static_cast<volatile int32_t*>(flag_host)[3] : 0;       // (while pairs wait, every slot is busy)
        const bool first_attempts_left = static_cast<volatile int32_t*>(flag_host)[2] != 0;
        if (min_phase >= sched->n_phases) break;          // (a slot only stays finished when the queue was empty)
        reached = (head >= queue->n_queue && !(may_retry && first_attempts_left)) ? (min_phase < 0 ? 0 : min_phase) : 0;
    }
    if (min_phase < sched->n_phases && vd.status) {
        hipLaunchKernelGGL(k_mark_unfinished, dim3((n_slots + 255) / 256), dim3(256), 0, s, phase, queue->slot_pair, n_slots, sched->n_phases, vd.status);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(1000 + (int)e);
    }
    return it;
}

int sp_pairs_schedule_run(const SpSchedule* sched, int n_pairs, int max_N, float lm_up, float lm_down, float lm_min,
                          float* lm_state, float* backup, float* costs, int32_t* phase, int32_t* iters, int check_every,
                          int max_rounds, int32_t* flag_dev, int32_t* flag_host, const SpVerdict* verdict, void* stream) {
    if (!sched || !phase || !iters || !flag_dev || !flag_host || check_every <= 0 || max_rounds < 0) return SP_EINVAL;
    if (!lm_state || !backup || !costs || n_pairs <= 0 || max_N <= 0) return SP_EINVAL;
    if (int rc = check_schedule(sched, verdict)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool may_retry = verdict && verdict->status && verdict->attempts && verdict->retry_mask != 0 && (sched->retry_entry >= 0 || sched->retry2_entry >= 0);
    const int last_attempt = sched->retry2_entry >= 0 ? 2 : 1;
    int it = 0;
    int reached = 0;                  // min(phase) at the last poll: pairs only move forward, so work lists behind it are not launched
    int min_phase = 0;
    const uint32_t adam_mask = adam_phases(sched);
    uint32_t occupied = 0xffffffffu;  // (the caller set the phases: nothing is known before the first poll)
    // (Staying one group of iterations AHEAD of the poll being waited for -- events instead of a stream synchronisation, the GPU
    //  never idle while the host wakes up -- measured no better than this loop, 30.1 k against 30.7 k frame pairs/s: the tail of a
    //  schedule is bound by the latency of a few pairs' own iterations, ~50 us each, not by the launch path.)
    while (it < max_rounds) {
        const int n = (max_rounds - it) < check_every ? (max_rounds - it) : check_every;
        const uint32_t idle = adam_mask & ~occupied;
        for (int k = 0; k < n; ++k, ++it) {
            int rc = schedule_cost_from(sched, phase, stream, reached, nullptr, 0, nullptr, 0, idle);
            if (rc == 0) rc = check_schedule(sched, verdict);
            if (rc != 0) return rc < 0 ? rc : -(1000 + rc);
            GnArgs ga{max_N, lm_up, lm_down, lm_min, lm_state, backup, costs, 0.f, nullptr, phase, iters, 0, 0, 0};
            ga.idle_mask = idle;
            hipLaunchKernelGGL(k_pairs_gn_sched, dim3(n_pairs), dim3(SP_BLOCK), 0, s, *sched, ga, SpQueue{}, verdict ? *verdict : SpVerdict{}, (const int32_t*)nullptr);
            hipError_t el = hipGetLastError();
            if (el != hipSuccess) return -(1000 + (int)el);
        }
        hipLaunchKernelGGL(k_phase_min, dim3(1), dim3(SP_BLOCK), 0, s, phase, n_pairs, flag_dev, (const int32_t*)nullptr,
                           may_retry ? (const int32_t*)verdict->attempts : (const int32_t*)nullptr, (const int32_t*)nullptr, sched->n_phases, last_attempt,
                           (int32_t*)nullptr);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(flag_host, flag_dev, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return -(1000 + (int)e);
        reached = min_phase = static_cast<volatile int32_t*>(flag_host)[0];
        occupied = (uint32_t)static_cast<volatile int32_t*>(flag_host)[4];
        if (reached >= sched->n_phases) break;
        // (a second attempt restarts a pair at retry_entry at any time: no work list can be left out while a first attempt is still running)
        if (reached < 0 || (may_retry && static_cast<volatile int32_t*>(flag_host)[2] != 0)) reached = 0;
    }
    if (verdict && verdict->status && min_phase < sched->n_phases) {
        hipLaunchKernelGGL(k_mark_unfinished, dim3((n_pairs + 255) / 256), dim3(256), 0, s, phase, (const int32_t*)nullptr, n_pairs, sched->n_phases, verdict->status);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(1000 + (int)e);
    }
    return it;
}

int sp_pairs_gn_step(const SpPair* pairs, int n_pairs, int max_N, const float* partials, const float* seg_partials,
                     float lm_up, float lm_down, float lm_min, float* lm_state, float* backup, float* costs, void* stream) {
    return sp_pairs_gn_step_conv(pairs, n_pairs, max_N, partials, seg_partials, lm_up, lm_down, lm_min, lm_state, backup, costs, 0.f,
                                 nullptr, stream);
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
