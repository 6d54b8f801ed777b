// Instance front-end (gfx950): displacement-field centroid refinement, 4-connected component
// labelling with raster-order ids, centroid clustering.
//
// Replaces reference step/make_ins_seg_labels.py:18-75 (300 numpy iterations of bilinear gather on
// the CPU; skimage.measure.label; misc/imutils.compress_range).  Every pixel's trajectory is
// independent, so the refinement is one thread per pixel; dp (2*h*w fp32 = 128 KB at 128^2) stays
// in L2.  Bit-exactness with numpy needs the reference's mixed precision replayed exactly
// (SURVEY.md §3.5): float32 state, float64 increment evaluated left to right with NO fused
// multiply-add, float32 rounding after each +=, round-half-even at the end.
#include "kernels.hpp"

namespace irn {
namespace {

// ---------------------------------------------------------------------------------------------
// find_centroids_with_refinement
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double bilinear_inc(const float *__restrict__ f, int w, int uy, int ly, int ux,
                                               int lx, double fy, double fx) {
    // f[uy,ux]*fy*fx + f[ly,ux]*(1-fy)*fx + f[uy,lx]*fy*(1-fx) + f[ly,lx]*(1-fy)*(1-fx)
    // (step/make_ins_seg_labels.py:39-42), float32 * float64 -> float64, evaluated left to right
    const double gy = __dsub_rn(1.0, fy), gx = __dsub_rn(1.0, fx);
    const double t0 = __dmul_rn(__dmul_rn((double)f[uy * w + ux], fy), fx);
    const double t1 = __dmul_rn(__dmul_rn((double)f[ly * w + ux], gy), fx);
    const double t2 = __dmul_rn(__dmul_rn((double)f[uy * w + lx], fy), gx);
    const double t3 = __dmul_rn(__dmul_rn((double)f[ly * w + lx], gy), gx);
    return __dadd_rn(__dadd_rn(__dadd_rn(t0, t1), t2), t3);
}

__global__ __launch_bounds__(256) void centroid_kernel(const float *__restrict__ dp, int h, int w, int iters,
                                                       int32_t *__restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int n = h * w;
    if (p >= n) return;
    const float *dy_f = dp, *dx_f = dp + n;
    float cy = (float)(p / w), cx = (float)(p % w);
    const float ymax = (float)(h - 1), xmax = (float)(w - 1);
    for (int it = 0; it < iters; ++it) {
        const float cyu = ceilf(cy), cyl = floorf(cy), cxu = ceilf(cx), cxl = floorf(cx);
        const int uy = (int)cyu, ly = (int)cyl, ux = (int)cxu, lx = (int)cxl;
        const double fy = __dsub_rn((double)cy, (double)ly);   // float32 - int32 -> float64 in numpy
        const double fx = __dsub_rn((double)cx, (double)lx);
        const double iy = bilinear_inc(dy_f, w, uy, ly, ux, lx, fy, fx);
        const double ix = bilinear_inc(dx_f, w, uy, ly, ux, lx, fy, fx);
        cy = (float)__dadd_rn((double)cy, iy);                 // in-place += rounds back to float32
        cx = (float)__dadd_rn((double)cx, ix);
        cy = fminf(fmaxf(cy, 0.f), ymax);
        cx = fminf(fmaxf(cx, 0.f), xmax);
    }
    out[p] = (int32_t)rintf(cy);       // np.round: half to even
    out[n + p] = (int32_t)rintf(cx);
}

// ---------------------------------------------------------------------------------------------
// 4-connected component labelling: union-find with "smaller root wins", so a component's root
// is its first pixel in raster order; ids = rank of the root among roots (+1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int *parent, int a) {
    int r = a;
    while (true) {
        const int q = __hip_atomic_load(parent + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q == r) return r;
        r = q;
    }
}

__device__ __forceinline__ void uf_union(int *parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        // hang the larger root b under the smaller root a, unless b stopped being a root meanwhile
        const int old = atomicMin(parent + b, a);
        if (old == b) return;
        b = old;
    }
}

__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t *__restrict__ mask, int *__restrict__ parent,
                                                       long total, int npx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    parent[i] = mask[i] ? (int)(i % npx) : -1;
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(int *__restrict__ parent_all, int h, int w, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int npx = h * w;
    const int img = (int)(i / npx), p = (int)(i - (long)img * npx);
    int *parent = parent_all + (long)img * npx;
    if (parent[p] < 0) return;
    const int y = p / w, x = p - y * w;
    if (x > 0 && parent[p - 1] >= 0) uf_union(parent, p, p - 1);
    if (y > 0 && parent[p - w] >= 0) uf_union(parent, p, p - w);
}

__global__ __launch_bounds__(256) void ccl_flatten_kernel(int *__restrict__ parent_all, int npx, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int img = (int)(i / npx), p = (int)(i - (long)img * npx);
    int *parent = parent_all + (long)img * npx;
    if (parent[p] < 0) return;
    // roots never change in this pass (all unions are done), so plain chasing is race-free
    int r = p;
    while (parent[r] != r) r = parent[r];
    parent[p] = r;   // may shortcut another thread's chase; every value it can read still leads to r
}

// One workgroup per image: rank[root] = 1 + #roots before it (raster order); n_labels[img] = #roots.
__global__ __launch_bounds__(1024) void ccl_rank_kernel(const int *__restrict__ parent_all, int *__restrict__ rank_all,
                                                        int *__restrict__ n_labels, int npx) {
    __shared__ int sums[1024];
    const int img = blockIdx.x;
    const int *parent = parent_all + (long)img * npx;
    int *rank = rank_all + (long)img * npx;
    const int chunk = (npx + 1023) / 1024;
    const int lo = min(npx, (int)threadIdx.x * chunk), hi = min(npx, lo + chunk);
    int cnt = 0;
    for (int p = lo; p < hi; ++p) cnt += (parent[p] == p);
    sums[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {   // Hillis-Steele inclusive scan
        const int v = threadIdx.x >= s ? sums[threadIdx.x - s] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sums[threadIdx.x] - cnt;     // exclusive prefix
    for (int p = lo; p < hi; ++p)
        if (parent[p] == p) rank[p] = ++run;
    if (threadIdx.x == 1023 && n_labels) n_labels[img] = sums[1023];
}

__global__ __launch_bounds__(256) void ccl_relabel_kernel(const int *__restrict__ parent_all,
                                                          const int *__restrict__ rank_all,
                                                          int32_t *__restrict__ labels, int npx, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long base = (i / npx) * npx;
    const int r = parent_all[i];
    labels[i] = r < 0 ? 0 : rank_all[base + r];
}

// ---------------------------------------------------------------------------------------------
// cluster_centroids
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weak_mask_kernel(const float *__restrict__ dp, int n, float thres,
                                                        uint8_t *__restrict__ mask) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    // np.sqrt(dp[1]**2 + dp[0]**2) in float32, each op rounded (step/make_ins_seg_labels.py:61)
    const float a = dp[n + p], b = dp[p];
    const float s = __fsqrt_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)));
    mask[p] = s < thres ? 1 : 0;
}

// picked[p] = label at centroid(p) (+1, as the reference adds before compress_range); mark presence
__global__ __launch_bounds__(256) void pick_kernel(const int32_t *__restrict__ centroids,
                                                   const int32_t *__restrict__ labels, int n, int w,
                                                   int32_t *__restrict__ picked, int *__restrict__ present) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int v = labels[centroids[p] * w + centroids[n + p]] + 1;
    picked[p] = v;
    present[v] = 1;
}

// single workgroup: renumber the distinct values of `picked` ascending to 0..K-1 (compress_range,
// misc/imutils.py:182-190; its final "- min" is a no-op because the smallest value maps to 0)
__global__ __launch_bounds__(1024) void compress_kernel(int *__restrict__ present, int n_vals, int *__restrict__ k_out) {
    __shared__ int sums[1024];
    const int chunk = (n_vals + 1023) / 1024;
    const int lo = min(n_vals, (int)threadIdx.x * chunk), hi = min(n_vals, lo + chunk);
    int cnt = 0;
    for (int v = lo; v < hi; ++v) cnt += present[v] != 0;
    sums[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        const int t = threadIdx.x >= s ? sums[threadIdx.x - s] : 0;
        __syncthreads();
        sums[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sums[threadIdx.x] - cnt;
    for (int v = lo; v < hi; ++v) present[v] = present[v] ? run++ : -1;   // now: new id of value v
    if (threadIdx.x == 1023) *k_out = sums[1023];
}

__global__ __launch_bounds__(256) void remap_kernel(const int32_t *__restrict__ picked, const int *__restrict__ newid,
                                                    int n, int32_t *__restrict__ cluster_map) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    cluster_map[p] = newid[picked[p]];
}

// ---------------------------------------------------------------------------------------------
// detect_instance (reference step/make_ins_seg_labels.py:82-105) on the device.
//
// The reference labels the 4-connected components of every channel's mask separately; the masks are
// the one-hot planes of ONE argmax map (:145-147), so they are disjoint and all components come
// out of a single labelling pass over the class map (neighbours join when their classes agree).
// Detections are ordered like the reference's: channel ascending, then skimage's label order =
// raster order of each component's first pixel = of its union-find root.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void det_init_kernel(const int32_t *__restrict__ cls, int *__restrict__ parent, int npx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < npx) parent[p] = cls[p] > 0 ? p : -1;
}

__global__ __launch_bounds__(256) void det_merge_kernel(const int32_t *__restrict__ cls, int *__restrict__ parent, int h, int w) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const int c = cls[p];
    if (c <= 0) return;
    const int y = p / w, x = p - y * w;
    if (x > 0 && cls[p - 1] == c) uf_union(parent, p, p - 1);
    if (y > 0 && cls[p - w] == c) uf_union(parent, p, p - w);
}

// Every root (= first raster pixel of its component) claims a provisional id and records its sort key
// channel * npx + pixel: detections are ordered channel ascending, then by first pixel (skimage's
// label order inside a channel).
__global__ __launch_bounds__(256) void det_roots_kernel(const int32_t *__restrict__ cls, const int *__restrict__ parent,
                                                         int *__restrict__ prov, long long *__restrict__ keys,
                                                         int *__restrict__ counter, int npx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    if (parent[p] == p) {
        const int id = atomicAdd(counter, 1);
        prov[p] = id;
        keys[id] = (long long)(cls[p] - 1) * npx + p;
    }
}

// final id of provisional detection i = number of keys below its own (keys are distinct); n is a few
// hundred at most, one workgroup does the n^2 / 1024 comparisons.  Also zeroes the statistics.
__global__ __launch_bounds__(1024) void det_order_kernel(const long long *__restrict__ keys, int *__restrict__ newid,
                                                         int *__restrict__ area, int *__restrict__ score_bits,
                                                         int32_t *__restrict__ channel, int n, int npx) {
    for (int i = threadIdx.x; i < n; i += 1024) {
        const long long k = keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += keys[j] < k;
        newid[i] = r;
        channel[r] = (int32_t)(k / npx);
        area[i] = 0;
        score_bits[i] = 0;
    }
}

// Pixel p belongs to detection newid[prov[root(p)]]: its mask byte, and area / max score per detection.  Scores are compared as int bit patterns: the reference takes max(score * mask), which
// is >= 0 whatever the scores are, and so is a maximum that starts from +0.
__global__ __launch_bounds__(256) void det_stats_kernel(const int32_t *__restrict__ cls, const int *__restrict__ parent,
                                                        const int *__restrict__ prov, const int *__restrict__ newid,
                                                        const float *__restrict__ rw_up, int *__restrict__ area,
                                                        int *__restrict__ score_bits, uint8_t *__restrict__ mask, int npx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c = p < npx ? cls[p] : 0;
    const bool fg = c > 0;
    int d = -1, bits = 0;
    if (fg) {
        d = newid[prov[parent[p]]];
        mask[(long)d * npx + p] = 1;
        const float sc = rw_up[(long)(c - 1) * npx + p];
        bits = sc > 0.f ? __float_as_int(sc) : 0;
    }
    // A wave's 64 consecutive pixels nearly always lie in ONE detection: one pair of atomics per wave then
    // (per-pixel atomics onto a handful of addresses took 1.4 ms per 512^2 image — 80 % of the whole step).
    const unsigned long long act = __ballot(fg);
    if (act == 0) return;
    const int d0 = __shfl(d, __ffsll((long long)act) - 1);
    if (__all(!fg || d == d0)) {
        int mx = bits;
        for (int sft = 32; sft > 0; sft >>= 1) mx = max(mx, __shfl_xor(mx, sft));
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(area + d0, __popcll(act));
            if (mx > 0) atomicMax(score_bits + d0, mx);
        }
    } else if (fg) {
        atomicAdd(area + d, 1);
        if (bits > 0) atomicMax(score_bits + d, bits);
    }
}

__global__ __launch_bounds__(256) void det_final_kernel(const int *__restrict__ area, const int *__restrict__ score_bits,
                                                        double min_area, float *__restrict__ score, int n_det) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n_det) return;
    score[d] = ((double)area[d] < min_area) ? 0.f : __int_as_float(score_bits[d]);
}

int run_label4(const uint8_t *mask, int n, int h, int w, int32_t *labels, int32_t *n_labels, void *scratch,
               hipStream_t stream) {
    const int npx = h * w;
    const long total = (long)n * npx;
    int *parent = (int *)scratch;
    int *rank = parent + total;   // separate from `labels`: relabel overwrites roots other pixels still need
    const int nb = (int)((total + 255) / 256);
    hipLaunchKernelGGL(ccl_init_kernel, dim3(nb), dim3(256), 0, stream, mask, parent, total, npx);
    IRN_LAUNCH_CHECK("ccl_init_kernel");
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(nb), dim3(256), 0, stream, parent, h, w, total);
    IRN_LAUNCH_CHECK("ccl_merge_kernel");
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3(nb), dim3(256), 0, stream, parent, npx, total);
    IRN_LAUNCH_CHECK("ccl_flatten_kernel");
    hipLaunchKernelGGL(ccl_rank_kernel, dim3(n), dim3(1024), 0, stream, parent, rank, n_labels, npx);
    IRN_LAUNCH_CHECK("ccl_rank_kernel");
    hipLaunchKernelGGL(ccl_relabel_kernel, dim3(nb), dim3(256), 0, stream, parent, rank, labels, npx, total);
    IRN_LAUNCH_CHECK("ccl_relabel_kernel");
    return IRN_OK;
}

}  // namespace
}  // namespace irn

using namespace irn;

extern "C" int irn_find_centroids(const float *dp_dev, int h, int w, int iterations, int32_t *centroids_dev,
                                  void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dp_dev || !centroids_dev || h < 1 || w < 1 || iterations < 0)
        return fail(IRN_ERR_ARG, "irn_find_centroids: bad argument");
    hipLaunchKernelGGL(centroid_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, stream, dp_dev, h, w, iterations,
                       centroids_dev);
    IRN_LAUNCH_CHECK("centroid_kernel");
    return IRN_OK;
}

extern "C" size_t irn_ccl_scratch_bytes(int n, int h, int w) {
    if (n < 1 || h < 1 || w < 1) return 0;
    return round_up(sizeof(int) * 2 * (size_t)n * h * w, 256);
}

extern "C" int irn_label4(const uint8_t *mask_dev, int n, int h, int w, int32_t *labels_dev, int32_t *n_labels_dev,
                          void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!mask_dev || !labels_dev || !scratch_dev || n < 1 || h < 1 || w < 1)
        return fail(IRN_ERR_ARG, "irn_label4: bad argument");
    if ((long)h * w > (1L << 30)) return fail(IRN_ERR_ARG, "irn_label4: image too large");
    return run_label4(mask_dev, n, h, w, labels_dev, n_labels_dev, scratch_dev, stream);
}

// scratch layout of irn_cluster_centroids:
//   [ccl scratch (2*npx ints)] [mask npx bytes, padded] [labels npx] [picked npx] [present npx+2] [k 1]
extern "C" size_t irn_cluster_scratch_bytes(int h, int w) {
    if (h < 1 || w < 1) return 0;
    const size_t npx = (size_t)h * w;
    return irn_ccl_scratch_bytes(1, h, w) + round_up(npx, 256) + round_up(4 * npx, 256) * 2 +
           round_up(4 * (npx + 2), 256) + 256;
}

extern "C" int irn_cluster_centroids(const int32_t *centroids_dev, const float *dp_dev, int h, int w, float thres,
                                     int32_t *cluster_map_dev, int *k_out, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!centroids_dev || !dp_dev || !cluster_map_dev || !k_out || !scratch_dev || h < 1 || w < 1)
        return fail(IRN_ERR_ARG, "irn_cluster_centroids: bad argument");
    const int npx = h * w;
    char *s = (char *)scratch_dev;
    void *ccl = s;
    s += irn_ccl_scratch_bytes(1, h, w);
    uint8_t *mask = (uint8_t *)s;
    s += round_up((size_t)npx, 256);
    int32_t *labels = (int32_t *)s;
    s += round_up(4 * (size_t)npx, 256);
    int32_t *picked = (int32_t *)s;
    s += round_up(4 * (size_t)npx, 256);
    int *present = (int *)s;
    s += round_up(4 * ((size_t)npx + 2), 256);
    int *k_dev = (int *)s;

    const int nb = cdiv(npx, 256);
    hipLaunchKernelGGL(weak_mask_kernel, dim3(nb), dim3(256), 0, stream, dp_dev, npx, thres, mask);
    IRN_LAUNCH_CHECK("weak_mask_kernel");
    int rc = run_label4(mask, 1, h, w, labels, nullptr, ccl, stream);
    if (rc) return rc;
    IRN_HIP_TRY(hipMemsetAsync(present, 0, sizeof(int) * ((size_t)npx + 2), stream));
    hipLaunchKernelGGL(pick_kernel, dim3(nb), dim3(256), 0, stream, centroids_dev, labels, npx, w, picked, present);
    IRN_LAUNCH_CHECK("pick_kernel");
    // label values lie in [0, npx/2+1]; +1 shifts them to [1, npx/2+2] -> npx+2 slots are plenty
    hipLaunchKernelGGL(compress_kernel, dim3(1), dim3(1024), 0, stream, present, npx + 2, k_dev);
    IRN_LAUNCH_CHECK("compress_kernel");
    hipLaunchKernelGGL(remap_kernel, dim3(nb), dim3(256), 0, stream, picked, present, npx, cluster_map_dev);
    IRN_LAUNCH_CHECK("remap_kernel");
    IRN_HIP_TRY(hipMemcpyAsync(k_out, k_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    IRN_HIP_TRY(hipStreamSynchronize(stream));
    return IRN_OK;
}

// scratch layout of irn_detect_instance_*: [parent npx][prov npx][newid npx][area npx][score_bits npx] int32,
//                                          [keys npx] int64, [counter]
extern "C" size_t irn_detect_scratch_bytes(int n_channels, int h, int w) {
    if (n_channels < 1 || h < 1 || w < 1) return 0;
    const size_t npx = (size_t)h * w;
    return round_up(4 * npx, 256) * 5 + round_up(8 * npx, 256) + 256;
}

namespace {
struct DetScratch {
    int *parent, *prov, *newid, *area, *score_bits, *counter;
    long long *keys;
};
DetScratch det_carve(void *scratch, size_t npx) {
    char *b = (char *)scratch;
    DetScratch s;
    const size_t a = round_up(4 * npx, 256);
    s.parent = (int *)b;
    s.prov = (int *)(b + a);
    s.newid = (int *)(b + 2 * a);
    s.area = (int *)(b + 3 * a);
    s.score_bits = (int *)(b + 4 * a);
    s.keys = (long long *)(b + 5 * a);
    s.counter = (int *)(b + 5 * a + round_up(8 * npx, 256));
    return s;
}
}  // namespace

extern "C" int irn_detect_instance_count(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h,
                                         int w, int *n_det_out, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!rw_up_dev || !argmax_dev || !n_det_out || !scratch_dev || n_channels < 1 || n_channels > 65535 || h < 1 || w < 1)
        return fail(IRN_ERR_ARG, "irn_detect_instance_count: bad argument");
    if ((long)h * w > (1L << 30)) return fail(IRN_ERR_ARG, "irn_detect_instance_count: image too large");
    const int npx = h * w;
    const DetScratch s = det_carve(scratch_dev, (size_t)npx);
    const int nb = cdiv(npx, 256);
    IRN_HIP_TRY(hipMemsetAsync(s.counter, 0, sizeof(int), stream));
    hipLaunchKernelGGL(det_init_kernel, dim3(nb), dim3(256), 0, stream, argmax_dev, s.parent, npx);
    IRN_LAUNCH_CHECK("det_init_kernel");
    hipLaunchKernelGGL(det_merge_kernel, dim3(nb), dim3(256), 0, stream, argmax_dev, s.parent, h, w);
    IRN_LAUNCH_CHECK("det_merge_kernel");
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3(nb), dim3(256), 0, stream, s.parent, npx, (long)npx);
    IRN_LAUNCH_CHECK("ccl_flatten_kernel");
    hipLaunchKernelGGL(det_roots_kernel, dim3(nb), dim3(256), 0, stream, argmax_dev, s.parent, s.prov, s.keys, s.counter, npx);
    IRN_LAUNCH_CHECK("det_roots_kernel");
    int n_det = 0;
    IRN_HIP_TRY(hipMemcpyAsync(&n_det, s.counter, sizeof(int), hipMemcpyDeviceToHost, stream));
    IRN_HIP_TRY(hipStreamSynchronize(stream));
    *n_det_out = n_det;
    return IRN_OK;
}

extern "C" int irn_detect_instance_emit(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h, int w,
                                        int n_det, double min_area, float *score_dev, int32_t *channel_dev,
                                        uint8_t *mask_dev, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!rw_up_dev || !argmax_dev || !score_dev || !channel_dev || !mask_dev || !scratch_dev || n_channels < 1 || h < 1 ||
        w < 1 || n_det < 1)
        return fail(IRN_ERR_ARG, "irn_detect_instance_emit: bad argument");
    const int npx = h * w;
    const DetScratch s = det_carve(scratch_dev, (size_t)npx);
    IRN_HIP_TRY(hipMemsetAsync(mask_dev, 0, (size_t)n_det * npx, stream));
    hipLaunchKernelGGL(det_order_kernel, dim3(1), dim3(1024), 0, stream, s.keys, s.newid, s.area, s.score_bits, channel_dev,
                       n_det, npx);
    IRN_LAUNCH_CHECK("det_order_kernel");
    hipLaunchKernelGGL(det_stats_kernel, dim3(cdiv(npx, 256)), dim3(256), 0, stream, argmax_dev, s.parent, s.prov, s.newid,
                       rw_up_dev, s.area, s.score_bits, mask_dev, npx);
    IRN_LAUNCH_CHECK("det_stats_kernel");
    hipLaunchKernelGGL(det_final_kernel, dim3(cdiv(n_det, 256)), dim3(256), 0, stream, s.area, s.score_bits, min_area,
                       score_dev, n_det);
    IRN_LAUNCH_CHECK("det_final_kernel");
    return IRN_OK;
}
