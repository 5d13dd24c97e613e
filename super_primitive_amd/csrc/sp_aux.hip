// Helpers either side of the cost pass: dense depth expansion (VOID path), depth splat render (keyframe
// decision / new-keyframe init), per-segment mean/median log-depth re-initialisation, per-pixel depth average.
#include "sp_device.h"

namespace {

__device__ __forceinline__ int segment_of(const int32_t* __restrict__ seg_off, int N, int i) {
    int lo = 0, hi = N;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    while (lo + 1 < N && seg_off[lo + 1] <= i) ++lo;
    return lo;
}

__device__ __forceinline__ float keypoint_L(const float* __restrict__ logdepth, const float* __restrict__ keypoints,
                                            int n, int H, int W) {
    int r = (int)rintf(0.5f * (float)(H - 1) * (keypoints[2 * n] + 1.f));
    int c = (int)rintf(0.5f * (float)(W - 1) * (keypoints[2 * n + 1] + 1.f));
    if (r < 0) r += H;
    if (c < 0) c += W;
    r = min(max(r, 0), H - 1);
    c = min(max(c, 0), W - 1);
    return logdepth[((size_t)n * H + r) * W + c];
}

// core/dense_optim.py:38-86,164-174: out = exp((L + (kld[n] - L[n,kp])) * mask), dense
__global__ __launch_bounds__(SP_BLOCK) void k_depth_expand(const uint8_t* __restrict__ masks, const float* __restrict__ logdepth,
                                                           const float* __restrict__ keypoints, const float* __restrict__ kld,
                                                           int H, int W, int log_space, float* __restrict__ out) {
    const int n = blockIdx.y;
    const int HW = H * W;
    const float shift = kld[n] - keypoint_L(logdepth, keypoints, n, H, W);
    const size_t base = (size_t)n * HW;
    for (int i = blockIdx.x * SP_BLOCK + threadIdx.x; i < HW; i += gridDim.x * SP_BLOCK) {
        const float m = masks[base + i] ? 1.f : 0.f;
        const float l = (logdepth[base + i] + shift) * m;
        out[base + i] = log_space ? l : expf(l);
    }
}

// table point -> source-camera 3-D point
__device__ __forceinline__ void table_point(const uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                            const int32_t* __restrict__ seg_off, const float* __restrict__ kp_L,
                                            const float* __restrict__ kld, int N, int i, const float* __restrict__ K9,
                                            float& x, float& y, float& d, int& n) {
    n = segment_of(seg_off, N, i);
    const uint32_t pw = pix[i] & 0x7fffffffu;
    const float col = (float)(pw & 0xffffu), row = (float)(pw >> 16);
    d = expf(baseL[i] + (kld[n] - kp_L[n]));
    x = __fdiv_rn(__fmul_rn(col - K9[2], d), K9[0]);
    y = __fdiv_rn(__fmul_rn(row - K9[5], d), K9[4]);
}

// core/ops.py:59-96 (mean=False): key = (point index + 1) << 32 | z bits, atomicMax => highest index wins,
// i.e. the result of a sequential scatter_
__global__ __launch_bounds__(SP_BLOCK) void k_splat_keys(const uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                                         const int32_t* __restrict__ seg_off, const float* __restrict__ kp_L,
                                                         const float* __restrict__ kld, int N, int P, int H, int W,
                                                         const float* __restrict__ K9, const float* __restrict__ T16,
                                                         unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= P) return;
    float x, y, d; int n;
    table_point(pix, baseL, seg_off, kp_L, kld, N, i, K9, x, y, d, n);
    const float qx = fmaf(T16[0], x, fmaf(T16[1], y, T16[2] * d)) + T16[3];
    const float qy = fmaf(T16[4], x, fmaf(T16[5], y, T16[6] * d)) + T16[7];
    const float qz = fmaf(T16[8], x, fmaf(T16[9], y, T16[10] * d)) + T16[11];
    const float zinv = (fabsf(qz) > 1e-6f) ? __fdiv_rn(1.0f, qz) : 1e-6f;
    const float u = qx * K9[0] * zinv + K9[2];
    const float v = qy * K9[4] * zinv + K9[5];
    if (!(qz > 1e-6f) || !isfinite(u) || !isfinite(v)) return;
    if (fabsf(u) > 1e9f || fabsf(v) > 1e9f) return;
    const long long c = (long long)u, r = (long long)v;   // truncation toward zero, like .long()
    if (r < 0 || r >= H || c < 0 || c >= W) return;
    const unsigned long long key = ((unsigned long long)(i + 1) << 32) | (unsigned long long)__float_as_uint(qz);
    atomicMax(&keys[(size_t)r * W + c], key);
}

// core/ops.py:84-92 (mean=True): scatter_reduce_(0, index, depth, reduce='mean') with its default include_self=True -- the zero
// the image was initialised with COUNTS as one sample: a pixel hit by c points holds sum / (c + 1), an untouched pixel 0.
// Sums in 32.32 fixed point (order independent, hence reproducible; the reference's fp32 sum depends on the scatter order).
__global__ __launch_bounds__(SP_BLOCK) void k_splat_mean_scatter(const uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                                                 const int32_t* __restrict__ seg_off, const float* __restrict__ kp_L,
                                                                 const float* __restrict__ kld, int N, int P, int H, int W,
                                                                 const float* __restrict__ K9, const float* __restrict__ T16,
                                                                 unsigned long long* __restrict__ sums, uint32_t* __restrict__ counts) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= P) return;
    float x, y, d; int n;
    table_point(pix, baseL, seg_off, kp_L, kld, N, i, K9, x, y, d, n);
    const float qx = fmaf(T16[0], x, fmaf(T16[1], y, T16[2] * d)) + T16[3];
    const float qy = fmaf(T16[4], x, fmaf(T16[5], y, T16[6] * d)) + T16[7];
    const float qz = fmaf(T16[8], x, fmaf(T16[9], y, T16[10] * d)) + T16[11];
    const float zinv = (fabsf(qz) > 1e-6f) ? __fdiv_rn(1.0f, qz) : 1e-6f;
    const float u = qx * K9[0] * zinv + K9[2];
    const float v = qy * K9[4] * zinv + K9[5];
    if (!(qz > 1e-6f) || !isfinite(u) || !isfinite(v)) return;
    if (fabsf(u) > 1e9f || fabsf(v) > 1e9f) return;
    const long long c = (long long)u, r = (long long)v;
    if (r < 0 || r >= H || c < 0 || c >= W) return;
    const size_t o = (size_t)r * W + c;
    atomicAdd(&sums[o], (unsigned long long)((double)qz * 4294967296.0));
    atomicAdd(&counts[o], 1u);
}

__global__ __launch_bounds__(SP_BLOCK) void k_splat_mean_finish(const unsigned long long* __restrict__ sums,
                                                                const uint32_t* __restrict__ counts, int HW, float* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= HW) return;
    const uint32_t c = counts[i];
    out[i] = c ? (float)((double)sums[i] * (1.0 / 4294967296.0)) / (float)(c + 1u) : 0.f;
}

__global__ __launch_bounds__(SP_BLOCK) void k_splat_decode(const unsigned long long* __restrict__ keys, int HW,
                                                           float* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= HW) return;
    const unsigned long long k = keys[i];
    out[i] = k ? __uint_as_float((uint32_t)(k & 0xffffffffull)) : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// odometery/depth_init.py:10-67
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th smallest (0-based) of vals[0..cnt) by 4 x 8-bit radix select; whole block cooperates
__device__ float block_select(const float* __restrict__ vals, int cnt, int k, uint32_t* hist /* 256 */) {
    uint32_t prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = threadIdx.x; b < 256; b += SP_BLOCK) hist[b] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += SP_BLOCK) {
            const uint32_t key = orderable(vals[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        __shared__ uint32_t chosen, remaining;
        if (threadIdx.x == 0) {
            uint32_t run = 0; int b = 0;
            for (; b < 256; ++b) { if (run + hist[b] > (uint32_t)k) break; run += hist[b]; }
            chosen = (uint32_t)b; remaining = (uint32_t)k - run;
        }
        __syncthreads();
        prefix |= chosen << shift;
        mask |= 0xffu << shift;
        k = (int)remaining;
        __syncthreads();
    }
    return from_orderable(prefix);
}

// one block per segment
__global__ __launch_bounds__(SP_BLOCK) void k_segment_reinit(const uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                                             const int32_t* __restrict__ seg_off, const float* __restrict__ kp_L,
                                                             int W, const float* __restrict__ est, int mode,
                                                             float* __restrict__ scratch, float* __restrict__ out_val,
                                                             uint8_t* __restrict__ out_visible) {
    __shared__ uint32_t hist[256];
    __shared__ int cursor;
    __shared__ double wsum[SP_WAVES];
    const int n = blockIdx.x;
    const int s0 = seg_off[n], s1 = seg_off[n + 1];
    float* vals = scratch + s0;
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    double sum = 0.0;
    // compaction order does not matter for mean/median
    for (int i = s0 + threadIdx.x; i < s1; i += SP_BLOCK) {
        const uint32_t pw = pix[i] & 0x7fffffffu;
        const float e = est[(size_t)(pw >> 16) * W + (pw & 0xffffu)];
        if (!(e < 1e-6f)) {
            const float v = logf(e) - baseL[i];
            vals[atomicAdd(&cursor, 1)] = v;
            sum += (double)v;
        }
    }
    __syncthreads();
    const int cnt = cursor;
    if (cnt == 0) {
        if (threadIdx.x == 0) { out_visible[n] = 0; out_val[n] = 0.f; }
        return;
    }
    float res;
    if (mode == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
        __syncthreads();
        double t = 0.0;
        for (int w = 0; w < SP_WAVES; ++w) t += wsum[w];
        res = (float)(t / (double)cnt);
    } else {
        __threadfence_block();
        res = block_select(vals, cnt, (cnt - 1) / 2, hist);   // lower middle, like torch.median
    }
    if (threadIdx.x == 0) { out_visible[n] = 1; out_val[n] = res + kp_L[n]; }
}

// invisible segments <- lower median of the visible segments' values (single block, rank counting)
__global__ __launch_bounds__(SP_BLOCK) void k_fill_invisible(float* __restrict__ val, const uint8_t* __restrict__ visible, int N) {
    __shared__ int nvis;
    __shared__ float med;
    if (threadIdx.x == 0) { nvis = 0; med = 0.f; }
    __syncthreads();
    int c = 0;
    for (int n = threadIdx.x; n < N; n += SP_BLOCK) c += visible[n] != 0;
    atomicAdd(&nvis, c);
    __syncthreads();
    if (nvis == 0 || nvis == N) return;
    const int k = (nvis - 1) / 2;
    for (int n = threadIdx.x; n < N; n += SP_BLOCK) {
        if (!visible[n]) continue;
        const float v = val[n];
        int rank = 0;
        for (int j = 0; j < N; ++j)
            if (visible[j]) rank += (val[j] < v) || (val[j] == v && j < n);
        if (rank == k) med = v;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += SP_BLOCK)
        if (!visible[n]) val[n] = med;
}

// depth_completion/segment_based_completion.py:21-27 + :45-52, fused: 32.32 fixed-point sums make the
// per-pixel accumulation order-independent (bitwise reproducible) and more accurate than fp32 adds
__global__ __launch_bounds__(SP_BLOCK) void k_average_scatter(const uint32_t* __restrict__ pix, const float* __restrict__ baseL,
                                                              const int32_t* __restrict__ seg_off, const float* __restrict__ kp_L,
                                                              const float* __restrict__ kld, const uint8_t* __restrict__ visible,
                                                              int N, int P, int W, unsigned long long* __restrict__ sums,
                                                              uint32_t* __restrict__ counts) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= P) return;
    const int n = segment_of(seg_off, N, i);
    if (visible && !visible[n]) return;
    const float d = expf(baseL[i] + (kld[n] - kp_L[n]));
    if (!(d > 1e-6f)) return;
    const uint32_t pw = pix[i] & 0x7fffffffu;
    const size_t o = (size_t)(pw >> 16) * W + (pw & 0xffffu);
    atomicAdd(&sums[o], (unsigned long long)((double)d * 4294967296.0));
    atomicAdd(&counts[o], 1u);
}

__global__ __launch_bounds__(SP_BLOCK) void k_average_finish(const unsigned long long* __restrict__ sums,
                                                             const uint32_t* __restrict__ counts, int HW,
                                                             float* __restrict__ out, uint8_t* __restrict__ invalid) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= HW) return;
    const float total = (float)((double)sums[i] * (1.0 / 4294967296.0));
    const uint32_t c = counts[i];
    out[i] = total / ((float)c + 1e-6f);
    invalid[i] = c == 0;
}

// ---------------------------------------------------------------------------------------------------
// odometery/kf_criteria.py:7-34 + the validity ratio of odometery/odometery.py:1003-1004, one launch, no host sync:
//   out[0] = #(depth > thresh) / n            out[1] = scale = lower median of the valid depths (torch.median)
//   out[2] = |t_src - t_trg| / (scale + 1e-6) out[3] = rotation angle of inv(pose_src) pose_trg in degrees
// Single workgroup: one counting pass + 4 radix passes over an image that sits in L2.
// ---------------------------------------------------------------------------------------------------
constexpr int KF_BLOCK = 1024;
__global__ __launch_bounds__(KF_BLOCK) void k_kf_criterion(const float* __restrict__ depth, int n, float thresh,
                                                           const float* __restrict__ pose_src,
                                                           const float* __restrict__ pose_trg, float* __restrict__ out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t chosen, remaining;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    // pass 0: the valid count and the top-byte histogram together (depth > thresh > 0 => raw bits are orderable)
    for (int i = threadIdx.x; i < n; i += KF_BLOCK) {
        const float v = depth[i];
        if (v > thresh) atomicAdd(&hist[__float_as_uint(v) >> 24], 1u);
    }
    __syncthreads();
    uint32_t cnt = 0;
    for (int b = 0; b < 256; ++b) cnt += hist[b];           // every thread: same value, LDS broadcast reads
    uint32_t prefix = 0, mask = 0;
    uint32_t k = cnt ? (cnt - 1u) / 2u : 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (shift != 24) {
            __syncthreads();
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += KF_BLOCK) {
                const float v = depth[i];
                const uint32_t key = __float_as_uint(v);
                if (v > thresh && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            uint32_t run = 0; int b = 0;
            for (; b < 255; ++b) { if (run + hist[b] > k) break; run += hist[b]; }
            chosen = (uint32_t)b; remaining = k - run;
        }
        __syncthreads();
        prefix |= chosen << shift;
        mask |= 0xffu << shift;
        k = remaining;
    }
    if (threadIdx.x == 0) {
        const float scale = cnt ? __uint_as_float(prefix) : __builtin_nanf("");     // torch.median of nothing raises
        out[0] = (float)cnt / (float)n;
        out[1] = scale;
        const float dx = pose_src[3] - pose_trg[3], dy = pose_src[7] - pose_trg[7], dz = pose_src[11] - pose_trg[11];
        out[2] = sqrtf(dx * dx + dy * dy + dz * dz) / (scale + 1e-6f);
        // rotation part of inv(pose_src) @ pose_trg = R_s^T R_t; angle = atan2(|axis part|, (trace - 1) / 2), in fp64
        double D[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int m = 0; m < 3; ++m) a += (double)pose_src[4 * m + i] * (double)pose_trg[4 * m + j];
                D[3 * i + j] = a;
            }
        const double ax = D[7] - D[5], ay = D[2] - D[6], az = D[3] - D[1];
        const double sn = 0.5 * sqrt(ax * ax + ay * ay + az * az), cs = 0.5 * (D[0] + D[4] + D[8] - 1.0);
        out[3] = (float)(atan2(sn, cs) * 57.29577951308232);
    }
}

// The same four values from a GRID of workgroups (sp_kf_criterion_ws): one launch per radix pass -- every workgroup histograms its slice
// of the image in LDS and adds it to the pass's global histogram; the workgroup that finishes last picks the digit (and, after the last
// pass, writes out[]).  ws: KF_WS_WORDS uint32, zeroed by the entry point: [pass * 256 + bin] histograms, then {prefix, k, count, -} and
// the four passes' tickets.  Exactly k_kf_criterion's selection -- the k-th smallest key -- so the same bits come out.
constexpr int KF_WS_WORDS = 4 * 256 + 8;
constexpr int KF_GRID_BLOCK = 256, KF_PER_THREAD = 16;
__global__ __launch_bounds__(KF_GRID_BLOCK) void k_kf_select_pass(const float* __restrict__ depth, int n, float thresh, int pass, uint32_t* __restrict__ ws,
                                                                  const float* __restrict__ pose_src, const float* __restrict__ pose_trg,
                                                                  float* __restrict__ out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t last;
    uint32_t* g_hist = ws + pass * 256;
    uint32_t* st = ws + 4 * 256;                         // [0] prefix, [1] k, [2] count, [4 + pass] ticket
    const int shift = 24 - 8 * pass;
    const uint32_t prefix = pass ? st[0] : 0u, mask = pass ? (0xffffffffu << (shift + 8)) : 0u;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * (KF_GRID_BLOCK * KF_PER_THREAD);
#pragma unroll 4
    for (int j = 0; j < KF_PER_THREAD; ++j) {
        const int i = base + j * KF_GRID_BLOCK + threadIdx.x;
        if (i < n) {
            const float v = depth[i];
            const uint32_t key = __float_as_uint(v);
            if (v > thresh && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
        }
    }
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&g_hist[threadIdx.x], hist[threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&st[4 + pass], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    hist[threadIdx.x] = __hip_atomic_load(&g_hist[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t cnt = st[2], k = st[1];
    if (pass == 0) {
        cnt = 0;
        for (int b = 0; b < 256; ++b) cnt += hist[b];
        k = cnt ? (cnt - 1u) / 2u : 0u;
        st[2] = cnt;
    }
    uint32_t run = 0; int b = 0;
    for (; b < 255; ++b) { if (run + hist[b] > k) break; run += hist[b]; }
    const uint32_t pre = prefix | ((uint32_t)b << shift);
    st[0] = pre; st[1] = k - run;
    if (pass != 3) return;
    const float scale = cnt ? __uint_as_float(pre) : __builtin_nanf("");
    out[0] = (float)cnt / (float)n;
    out[1] = scale;
    const float dx = pose_src[3] - pose_trg[3], dy = pose_src[7] - pose_trg[7], dz = pose_src[11] - pose_trg[11];
    out[2] = sqrtf(dx * dx + dy * dy + dz * dz) / (scale + 1e-6f);
    double D[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int m = 0; m < 3; ++m) a += (double)pose_src[4 * m + i] * (double)pose_trg[4 * m + j];
            D[3 * i + j] = a;
        }
    const double ax = D[7] - D[5], ay = D[2] - D[6], az = D[3] - D[1];
    const double sn = 0.5 * sqrt(ax * ax + ay * ay + az * az), cs = 0.5 * (D[0] + D[4] + D[8] - 1.0);
    out[3] = (float)(atan2(sn, cs) * 57.29577951308232);
}

}  // namespace

extern "C" {

int sp_depth_expand(const uint8_t* masks, const float* logdepth, const float* keypoints, const float* kld, int N,
                    int H, int W, int log_space, float* out, void* stream) {
    if (!masks || !logdepth || !keypoints || !kld || !out || N <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    const int HW = H * W;
    int gx = (HW + SP_BLOCK - 1) / SP_BLOCK;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_depth_expand, dim3(gx, N), dim3(SP_BLOCK), 0, static_cast<hipStream_t>(stream), masks, logdepth,
                       keypoints, kld, H, W, log_space, out);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_depth_splat(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                   const float* kld, int N, int P, int H, int W, const float* K, const float* pose,
                   unsigned long long* keys, float* out, void* stream) {
    if (!pix || !baseL || !seg_off || !kp_L || !kld || !K || !pose || !keys || !out) return SP_EINVAL;
    if (N <= 0 || P <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)H * W, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_splat_keys, dim3((P + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, pix, baseL, seg_off, kp_L,
                       kld, N, P, H, W, K, pose, keys);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_splat_decode, dim3((H * W + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, keys, H * W, out);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_depth_splat_mean(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L, const float* kld, int N,
                        int P, int H, int W, const float* K, const float* pose, void* acc, float* out, void* stream) {
    if (!pix || !baseL || !seg_off || !kp_L || !kld || !K || !pose || !acc || !out) return SP_EINVAL;
    if (N <= 0 || P <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)H * W;
    hipError_t e = hipMemsetAsync(acc, 0, 12 * HW, s);
    if (e != hipSuccess) return (int)e;
    unsigned long long* sums = static_cast<unsigned long long*>(acc);
    uint32_t* counts = reinterpret_cast<uint32_t*>(sums + HW);
    hipLaunchKernelGGL(k_splat_mean_scatter, dim3((P + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, pix, baseL, seg_off, kp_L, kld, N,
                       P, H, W, K, pose, sums, counts);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_splat_mean_finish, dim3((unsigned)((HW + SP_BLOCK - 1) / SP_BLOCK)), dim3(SP_BLOCK), 0, s, sums, counts, (int)HW, out);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_segment_reinit(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L, int N,
                      int P, int H, int W, const float* est_depth, int mode, float* scratch, float* out_kld,
                      uint8_t* out_visible, void* stream) {
    if (!pix || !baseL || !seg_off || !kp_L || !est_depth || !scratch || !out_kld || !out_visible) return SP_EINVAL;
    if (N <= 0 || P <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_segment_reinit, dim3(N), dim3(SP_BLOCK), 0, s, pix, baseL, seg_off, kp_L, W, est_depth, mode,
                       scratch, out_kld, out_visible);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_fill_invisible, dim3(1), dim3(SP_BLOCK), 0, s, out_kld, out_visible, N);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_depth_accumulate(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                        const float* kld, const uint8_t* visible, int N, int P, int H, int W, void* acc, void* stream) {
    if (!pix || !baseL || !seg_off || !kp_L || !kld || !acc) return SP_EINVAL;
    if (N <= 0 || P <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t HW = (size_t)H * W;
    hipError_t e = hipMemsetAsync(acc, 0, 12 * HW, s);
    if (e != hipSuccess) return (int)e;
    unsigned long long* sums = static_cast<unsigned long long*>(acc);
    uint32_t* counts = reinterpret_cast<uint32_t*>(sums + HW);
    hipLaunchKernelGGL(k_average_scatter, dim3((P + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, pix, baseL, seg_off,
                       kp_L, kld, visible, N, P, W, sums, counts);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_depth_average_finish(const void* acc, int H, int W, float* out_depth, uint8_t* out_invalid, void* stream) {
    if (!acc || !out_depth || !out_invalid || H <= 0 || W <= 0) return SP_EINVAL;
    const size_t HW = (size_t)H * W;
    const unsigned long long* sums = static_cast<const unsigned long long*>(acc);
    const uint32_t* counts = reinterpret_cast<const uint32_t*>(sums + HW);
    hipLaunchKernelGGL(k_average_finish, dim3((unsigned)((HW + SP_BLOCK - 1) / SP_BLOCK)), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), sums, counts, (int)HW, out_depth, out_invalid);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_depth_average(const uint32_t* pix, const float* baseL, const int32_t* seg_off, const float* kp_L,
                     const float* kld, const uint8_t* visible, int N, int P, int H, int W, void* acc,
                     float* out_depth, uint8_t* out_invalid, void* stream) {
    if (!out_depth || !out_invalid) return SP_EINVAL;
    const int rc = sp_depth_accumulate(pix, baseL, seg_off, kp_L, kld, visible, N, P, H, W, acc, stream);
    if (rc != 0) return rc;
    return sp_depth_average_finish(acc, H, W, out_depth, out_invalid, stream);
}

int sp_kf_criterion(const float* depth, int n, float thresh, const float* pose_src, const float* pose_trg, float* out,
                    void* stream) {
    if (!depth || !pose_src || !pose_trg || !out || n <= 0 || !(thresh >= 0.f)) return SP_EINVAL;
    hipLaunchKernelGGL(k_kf_criterion, dim3(1), dim3(KF_BLOCK), 0, static_cast<hipStream_t>(stream), depth, n, thresh,
                       pose_src, pose_trg, out);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_kf_criterion_ws_words(void) { return KF_WS_WORDS; }

int sp_kf_criterion_ws(const float* depth, int n, float thresh, const float* pose_src, const float* pose_trg, uint32_t* ws, float* out,
                       void* stream) {
    if (!depth || !pose_src || !pose_trg || !out || !ws || n <= 0 || !(thresh >= 0.f)) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(uint32_t) * KF_WS_WORDS, s);
    if (e != hipSuccess) return (int)e;
    const int grid = (n + KF_GRID_BLOCK * KF_PER_THREAD - 1) / (KF_GRID_BLOCK * KF_PER_THREAD);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(k_kf_select_pass, dim3(grid), dim3(KF_GRID_BLOCK), 0, s, depth, n, thresh, pass, ws, pose_src, pose_trg, out);
        SP_CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
