// Keyframe post-processing next to the hot path: split segments at depth discontinuities and
// into connected components -- the reference's frontend/segment/post_processer.py, which needs cupy's
// ndimage.label (CUDA only).  Here: fused exp + masked max-pool, Scharr magnitude threshold, and a union-find
// connected-component labelling (4-connectivity per mask slice, root = smallest linear pixel index, so sorting
// components by label reproduces scipy/cupy's scan-order numbering).
#include "sp_device.h"

namespace {

// frontend/segment/post_processer.py:17-21: depth = exp(logdepth), -1 where invalid, fs x fs max-pool, stride 1,
// implicit -inf padding
__global__ __launch_bounds__(SP_BLOCK) void k_masked_depth_maxpool(const float* __restrict__ logdepth,
                                                                   const uint8_t* __restrict__ valid, int H, int W, int fs,
                                                                   float* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int r = i / W, c = i - r * W, h = fs / 2;
    float m = -INFINITY;
    for (int dy = -h; dy <= h; ++dy) {
        const int y = r + dy;
        if (y < 0 || y >= H) continue;
        for (int dx = -h; dx <= h; ++dx) {
            const int x = c + dx;
            if (x < 0 || x >= W) continue;
            const size_t j = base + (size_t)y * W + x;
            m = fmaxf(m, valid[j] ? expf(logdepth[j]) : -1.f);
        }
    }
    out[base + i] = m;
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// image/image_processing.py:4-30 (Scharr /32, reflect padding) + post_processer.py:26-27,31-36
__global__ __launch_bounds__(SP_BLOCK) void k_scharr_split(const float* __restrict__ pooled, const uint8_t* __restrict__ valid,
                                                           int H, int W, float threshold, uint8_t* __restrict__ split,
                                                           uint8_t* __restrict__ disc) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int r = i / W, c = i - r * W;
    float v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            v[dy][dx] = pooled[base + (size_t)reflect1(r + dy - 1, H) * W + reflect1(c + dx - 1, W)];
    const float k = 1.f / 32.f;
    const float gx = k * (-3.f * v[0][0] + 3.f * v[0][2] - 10.f * v[1][0] + 10.f * v[1][2] - 3.f * v[2][0] + 3.f * v[2][2]);
    const float gy = k * (-3.f * v[0][0] - 10.f * v[0][1] - 3.f * v[0][2] + 3.f * v[2][0] + 10.f * v[2][1] + 3.f * v[2][2]);
    const bool ok = valid[base + i] != 0;
    const bool d = ok && (sqrtf(gx * gx + gy * gy) > threshold);
    split[base + i] = ok && !d;
    if (disc) disc[base + i] = d;
}

// ---- union-find connected components ----------------------------------------------------------------
__device__ __forceinline__ int uf_load(const int* P, int i) { return __hip_atomic_load(P + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int uf_find(const int* P, int i) {
    int p = uf_load(P, i);
    while (p != i) { i = p; p = uf_load(P, i); }
    return i;
}
__device__ void uf_union(int* P, int a, int b) {
    bool done;
    do {
        a = uf_find(P, a);
        b = uf_find(P, b);
        if (a < b) { const int old = atomicMin(P + b, a); done = old == b; b = old; }
        else if (b < a) { const int old = atomicMin(P + a, b); done = old == a; a = old; }
        else done = true;
    } while (!done);
}

__global__ __launch_bounds__(SP_BLOCK) void k_ccl_init(const uint8_t* __restrict__ fg, int total, int* __restrict__ P) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i < total) P[i] = fg[i] ? i : -1;
}

// 4-connectivity inside each (H,W) slice: unite with the west and north neighbours
__global__ __launch_bounds__(SP_BLOCK) void k_ccl_merge(const uint8_t* __restrict__ fg, int total, int H, int W, int* __restrict__ P) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= total || !fg[i]) return;
    const int in_slice = i % (H * W);
    const int r = in_slice / W, c = in_slice - r * W;
    if (c > 0 && fg[i - 1]) uf_union(P, i, i - 1);
    if (r > 0 && fg[i - W]) uf_union(P, i, i - W);
}

// labels = root + 1 (background 0); sizes[root] = pixel count of the component
__global__ __launch_bounds__(SP_BLOCK) void k_ccl_compress(const int* __restrict__ P, int total, int32_t* __restrict__ labels,
                                                           int32_t* __restrict__ sizes) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= total) return;
    if (P[i] < 0) { labels[i] = 0; return; }
    const int root = uf_find(P, i);
    labels[i] = root + 1;
    if (sizes) atomicAdd(sizes + root, 1);
}

// every root pixel appends {slice, root, size}; per slice also the size of "mask minus split" (post_processer.py:
// the label-0 part that post_process_kf forms by AND-ing the background with the segment mask)
__global__ __launch_bounds__(SP_BLOCK) void k_collect_parts(const int32_t* __restrict__ labels, const int32_t* __restrict__ sizes,
                                                            const uint8_t* __restrict__ masks, const uint8_t* __restrict__ split,
                                                            int total, int HW, int cap, int32_t* __restrict__ parts,
                                                            int32_t* __restrict__ n_parts, int32_t* __restrict__ bg_sizes) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= total) return;
    if (labels[i] == i + 1) {
        const int k = atomicAdd(n_parts, 1);
        if (k < cap) { parts[3 * k] = i / HW; parts[3 * k + 1] = i; parts[3 * k + 2] = sizes[i]; }
    }
    if (masks[i] && !split[i]) atomicAdd(bg_sizes + i / HW, 1);
}

// part k = {slice, kind, root}: kind 0 = component `root` (AND the segment mask), 1 = segment mask minus split,
// 2 = the whole original segment mask
__global__ __launch_bounds__(SP_BLOCK) void k_build_part_masks(const uint8_t* __restrict__ masks, const uint8_t* __restrict__ split,
                                                               const int32_t* __restrict__ labels, int HW,
                                                               const int32_t* __restrict__ parts, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    if (i >= HW) return;
    const int k = blockIdx.y;
    const int n = parts[3 * k], kind = parts[3 * k + 1], root = parts[3 * k + 2];
    const size_t j = (size_t)n * HW + i;
    bool on;
    if (kind == 0) on = masks[j] && labels[j] == root + 1;
    else if (kind == 1) on = masks[j] && !split[j];
    else on = masks[j] != 0;
    out[(size_t)k * HW + i] = on;
}

// the kth[k]-th set pixel of mask k in raster order (= torch.where(mask)[kth]); row_off = exclusive per-mask scan of
// the per-row counts (sp_mask_count)
__global__ void k_kth_pixel(const uint8_t* __restrict__ masks, const int32_t* __restrict__ row_off, int K, int H, int W,
                            const int32_t* __restrict__ kth, int32_t* __restrict__ out_rc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int32_t* ro = row_off + (size_t)k * H;
    const int target = kth[k];
    int lo = 0, hi = H;                       // last row with ro[row] <= target
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ro[mid] <= target) lo = mid; else hi = mid; }
    const uint8_t* m = masks + ((size_t)k * H + lo) * W;
    int seen = ro[lo], col = -1;
    for (int x = 0; x < W; ++x)
        if (m[x]) { if (seen == target) { col = x; break; } ++seen; }
    out_rc[2 * k] = lo;
    out_rc[2 * k + 1] = col;
}

}  // namespace

extern "C" {

int sp_depth_discontinuity(const float* logdepth, const uint8_t* valid, int N, int H, int W, int filter_size,
                           float threshold, float* scratch, uint8_t* split, uint8_t* disc, void* stream) {
    if (!logdepth || !valid || !scratch || !split || N <= 0 || H < 2 || W < 2) return SP_EINVAL;
    if (filter_size < 1 || !(filter_size & 1)) return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((H * W + SP_BLOCK - 1) / SP_BLOCK, N);
    hipLaunchKernelGGL(k_masked_depth_maxpool, grid, dim3(SP_BLOCK), 0, s, logdepth, valid, H, W, filter_size, scratch);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scharr_split, grid, dim3(SP_BLOCK), 0, s, scratch, valid, H, W, threshold, split, disc);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_label_components(const uint8_t* fg, int N, int H, int W, int32_t* parent, int32_t* labels, int32_t* sizes,
                        void* stream) {
    if (!fg || !parent || !labels || N <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    if ((long long)N * H * W > 0x7fffffffLL) return SP_ELIMIT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int total = N * H * W;
    const dim3 grid((total + SP_BLOCK - 1) / SP_BLOCK);
    if (sizes) {
        hipError_t e = hipMemsetAsync(sizes, 0, sizeof(int32_t) * (size_t)total, s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_ccl_init, grid, dim3(SP_BLOCK), 0, s, fg, total, parent);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_merge, grid, dim3(SP_BLOCK), 0, s, fg, total, H, W, parent);
    SP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_compress, grid, dim3(SP_BLOCK), 0, s, parent, total, labels, sizes);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_collect_parts(const int32_t* labels, const int32_t* sizes, const uint8_t* masks, const uint8_t* split, int N, int H,
                     int W, int cap, int32_t* parts, int32_t* n_parts, int32_t* bg_sizes, void* stream) {
    if (!labels || !sizes || !masks || !split || !parts || !n_parts || !bg_sizes || N <= 0 || H <= 0 || W <= 0 || cap <= 0)
        return SP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(n_parts, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(bg_sizes, 0, sizeof(int32_t) * (size_t)N, s);
    if (e != hipSuccess) return (int)e;
    const int total = N * H * W;
    hipLaunchKernelGGL(k_collect_parts, dim3((total + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, labels, sizes, masks,
                       split, total, H * W, cap, parts, n_parts, bg_sizes);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_build_part_masks(const uint8_t* masks, const uint8_t* split, const int32_t* labels, int H, int W,
                        const int32_t* parts, int K, uint8_t* out, void* stream) {
    if (!masks || !split || !labels || !parts || !out || K <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_build_part_masks, dim3((H * W + SP_BLOCK - 1) / SP_BLOCK, K), dim3(SP_BLOCK), 0,
                       static_cast<hipStream_t>(stream), masks, split, labels, H * W, parts, out);
    SP_CHECK_LAUNCH();
    return 0;
}

int sp_kth_mask_pixel(const uint8_t* masks, const int32_t* row_off, int K, int H, int W, const int32_t* kth,
                      int32_t* out_rc, void* stream) {
    if (!masks || !row_off || !kth || !out_rc || K <= 0 || H <= 0 || W <= 0) return SP_EINVAL;
    hipLaunchKernelGGL(k_kth_pixel, dim3((K + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), masks, row_off, K, H, W,
                       kth, out_rc);
    SP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
