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

struct CenJob {
    const float *dp;     // [2,h,w]
    int32_t *out;        // [2,h,w]
    int h, w;
};

// grid.y = image of the batch (every pixel's trajectory is independent; a batch fills the chip where one 128x128
// image is 64 workgroups of latency-bound gathers)
__global__ __launch_bounds__(256) void centroid_kernel(const CenJob *__restrict__ jobs, int iters) {
    const CenJob J = jobs[blockIdx.y];
    const float *__restrict__ dp = J.dp;
    int32_t *__restrict__ out = J.out;
    const int h = J.h, w = J.w;
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
// cluster_centroids, batched (grid.y or grid.x = image): K of every image stays on the device, the caller reads
// all of them with ONE transfer per batch (step/make_ins_seg_labels.py:58-75 returns K to the host per image).
// ---------------------------------------------------------------------------------------------
struct ClusterJob {
    const int32_t *centroids;   // [2,h,w]
    const float *dp;            // [2,h,w]
    int32_t *cluster_map;       // [h,w] out
    int *parent, *rank;         // [npx] union-find forest of the weak-displacement mask / rank of its roots
    int32_t *picked;            // [npx]
    int *present;               // [npx + 2]
    int32_t *k_out;             // -> k_dev[i]
    int h, w;
};

// weak = |dp| < thres, np.sqrt(dp[1]**2 + dp[0]**2) in float32 with every operation rounded
// (step/make_ins_seg_labels.py:61); forest initialised on the mask; `present` cleared
__global__ __launch_bounds__(256) void cluster_init_kernel(const ClusterJob *__restrict__ jobs, float thres) {
    const ClusterJob J = jobs[blockIdx.y];
    const int n = J.h * J.w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < 2) J.present[n + p] = 0;
    if (p >= n) return;
    const float a = J.dp[n + p], b = J.dp[p];
    const float s = __fsqrt_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)));
    J.parent[p] = s < thres ? p : -1;
    J.present[p] = 0;
}

__global__ __launch_bounds__(256) void cluster_merge_kernel(const ClusterJob *__restrict__ jobs) {
    const ClusterJob J = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= J.h * J.w) return;
    int *parent = J.parent;
    if (parent[p] < 0) return;
    const int y = p / J.w, x = p - y * J.w;
    if (x > 0 && parent[p - 1] >= 0) uf_union(parent, p, p - 1);
    if (y > 0 && parent[p - J.w] >= 0) uf_union(parent, p, p - J.w);
}

__global__ __launch_bounds__(256) void cluster_flatten_kernel(const ClusterJob *__restrict__ jobs) {
    const ClusterJob J = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= J.h * J.w) return;
    int *parent = J.parent;
    if (parent[p] < 0) return;
    int r = p;
    while (parent[r] != r) r = parent[r];      // all unions are done: roots are final
    parent[p] = r;
}

// one workgroup per image: rank[root] = 1 + number of roots before it in raster order (skimage numbering)
__global__ __launch_bounds__(1024) void cluster_rank_kernel(const ClusterJob *__restrict__ jobs) {
    __shared__ int sums[1024];
    const ClusterJob J = jobs[blockIdx.x];
    const int npx = J.h * J.w;
    const int *parent = J.parent;
    const int chunk = (npx + 1023) / 1024;
    const int lo = min(npx, (int)threadIdx.x * chunk), hi = min(npx, lo + chunk);
    int cnt = 0;
    for (int p = lo; p < hi; ++p) cnt += (parent[p] == p);
    sums[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        const int v = threadIdx.x >= s ? sums[threadIdx.x - s] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sums[threadIdx.x] - cnt;
    for (int p = lo; p < hi; ++p)
        if (parent[p] == p) J.rank[p] = ++run;
}

// picked[p] = label at centroid(p) (+1, as the reference adds before compress_range); mark presence
__global__ __launch_bounds__(256) void cluster_pick_kernel(const ClusterJob *__restrict__ jobs) {
    const ClusterJob J = jobs[blockIdx.y];
    const int n = J.h * J.w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int c = J.centroids[p] * J.w + J.centroids[n + p];
    const int root = J.parent[c];
    const int v = (root < 0 ? 0 : J.rank[root]) + 1;
    J.picked[p] = v;
    J.present[v] = 1;
}

// one workgroup per image: renumber the distinct values of `picked` ascending to 0..K-1 (compress_range,
// misc/imutils.py:182-190; its final "- min" is a no-op because the smallest value maps to 0)
__global__ __launch_bounds__(1024) void cluster_compress_kernel(const ClusterJob *__restrict__ jobs) {
    __shared__ int sums[1024];
    const ClusterJob J = jobs[blockIdx.x];
    int *present = J.present;
    const int n_vals = J.h * J.w + 2;       // label values lie in [0, npx/2+1]; +1 shifts them to [1, npx/2+2]
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
    if (threadIdx.x == 1023) *J.k_out = sums[1023];
}

__global__ __launch_bounds__(256) void cluster_remap_kernel(const ClusterJob *__restrict__ jobs) {
    const ClusterJob J = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= J.h * J.w) return;
    J.cluster_map[p] = J.present[J.picked[p]];
}

// ---------------------------------------------------------------------------------------------
// detect_instance (reference step/make_ins_seg_labels.py:82-105) on the device, batched over images.
//
// The reference labels the 4-connected components of every channel's mask separately; the masks are
// the one-hot planes of ONE argmax map (:145-147), so they are disjoint and all components come
// out of a single labelling pass over the class map (neighbours join when their classes agree).
// Detections are ordered like the reference's: channel ascending, then skimage's label order =
// raster order of each component's first pixel = of its union-find root.
// ---------------------------------------------------------------------------------------------
struct DetJob {
    const float *rw_up;         // [n_channels,h,w]
    const int32_t *cls;         // [h,w] argmax (0 = background, c+1 = channel c)
    int *parent, *prov, *newid, *area, *score_bits;   // [npx] each
    long long *keys;            // [npx]
    int32_t *counter;           // -> n_det_dev[i]
    float *score;               // [n_det]      (emit)
    int32_t *channel;           // [n_det]      (emit)
    uint8_t *mask;              // [n_det,h,w]  (emit)
    double min_area;
    int h, w, n_det;
};

// Labelling, stage 1: every 64 x 16 tile is labelled in LDS.  Row runs come from a segmented scan over the 16 lanes of a
// row (4 pixels per lane) — no atomics; runs of adjacent rows are joined by an LDS union-find ("smaller index wins": local
// raster order agrees with the image's inside a tile, so a local root is the component's first pixel of the tile), one
// union per PAIR of touching runs, not per pixel.  Every pixel leaves with parent = its local root's image index.  Per-pixel
// global atomics over the whole map (round 1-4: 0.056 ms per 512^2 image, the largest item of the instance stage,
// profiles/r05_s1_ins_breakdown_r5.txt) remain only along the tile borders (stage 2).
constexpr int kDetTW = 64, kDetTH = 16;

__device__ __forceinline__ int lds_find(volatile int *parent, int a) {
    int r = a;
    while (true) {
        const int q = parent[r];
        if (q == r) return r;
        r = q;
    }
}

__device__ __forceinline__ void lds_union(int *parent, int a, int b) {
    while (true) {
        a = lds_find(parent, a);
        b = lds_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(parent + b, a);
        if (old == b) return;
        b = old;
    }
}

__global__ __launch_bounds__(256) void det_local_kernel(const DetJob *__restrict__ jobs) {
    __shared__ int s_cls[kDetTH * kDetTW];
    __shared__ int s_parent[kDetTH * kDetTW];
    const DetJob J = jobs[blockIdx.y];
    const int H = J.h, W = J.w;
    const int tiles_x = (W + kDetTW - 1) / kDetTW, tiles_y = (H + kDetTH - 1) / kDetTH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * kDetTH, x0 = tx * kDetTW;
    const int t = threadIdx.x, ly = t >> 4, lane16 = t & 15, lx0 = lane16 * 4;
    const int gy = y0 + ly, gx0 = x0 + lx0;
    const int idx0 = ly * kDetTW + lx0;
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (gy < H && gx0 + j < W) ? J.cls[(long)gy * W + gx0 + j] : 0;
    // links to the left neighbour inside the tile row
    int left = __shfl_up(c[3], 1, 16);
    if (lane16 == 0) left = -1;
    bool link[4];
    link[0] = c[0] > 0 && c[0] == left;
#pragma unroll
    for (int j = 1; j < 4; ++j) link[j] = c[j] > 0 && c[j] == c[j - 1];
    // run start of the lane's LAST pixel: `pass` = the incoming start runs through all four pixels
    bool pass = link[0] && link[1] && link[2] && link[3];
    int val = idx0 + 3;
#pragma unroll
    for (int j = 3; j >= 1; --j) {
        if (!link[j]) break;
        val = idx0 + j - 1;
    }
    // (val is only used when !pass: the start of the run that contains pixel 3)
    // inclusive segmented scan over the row's 16 lanes
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int pv = __shfl_up(val, d, 16);
        const int pp = __shfl_up((int)pass, d, 16);
        if (lane16 >= d && pass) {
            val = pv;
            pass = pp != 0;
        }
    }
    int incoming = __shfl_up(val, 1, 16);     // run start of the left lane's last pixel (unused when !link[0])
    int start[4];
    int cur = link[0] ? incoming : idx0;
    start[0] = cur;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        cur = link[j] ? cur : idx0 + j;
        start[j] = cur;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_cls[idx0 + j] = c[j];
        s_parent[idx0 + j] = c[j] > 0 ? start[j] : -1;
    }
    __syncthreads();
    if (ly > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int up = idx0 + j - kDetTW;
            if (c[j] > 0 && s_cls[up] == c[j]) {
                // one union per pair of touching runs: where either run begins
                const bool here = start[j] == idx0 + j;
                const bool there = lx0 + j == 0 || s_cls[up - 1] != c[j];
                if (here || there) lds_union(s_parent, start[j], up);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (gy < H && gx0 + j < W) {
            int out = -1;
            if (c[j] > 0) {
                const int r = lds_find(s_parent, start[j]);
                out = (y0 + r / kDetTW) * W + x0 + (r % kDetTW);
            }
            J.parent[(long)gy * W + gx0 + j] = out;
        }
    }
}

// Labelling, stage 2: components are joined across tile borders — one global union per pair of touching border runs.
__global__ __launch_bounds__(256) void det_border_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    const int H = J.h, W = J.w;
    const int rows = (H + kDetTH - 1) / kDetTH - 1, cols = (W + kDetTW - 1) / kDetTW - 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int32_t *__restrict__ cls = J.cls;
    if (i < rows * W) {
        const int r = i / W, x = i - r * W;
        const int p = (r + 1) * kDetTH * W + x;
        const int c = cls[p];
        // (a border run restarts at every tile corner: p and p - 1 are only joined inside ONE tile)
        if (c > 0 && cls[p - W] == c && (x % kDetTW == 0 || cls[p - 1] != c || cls[p - W - 1] != c)) uf_union(J.parent, p, p - W);
        return;
    }
    const int k = i - rows * W;
    if (k < cols * H) {
        const int q = k / H, y = k - q * H;
        const int p = y * W + (q + 1) * kDetTW;
        const int c = cls[p];
        if (c > 0 && cls[p - 1] == c && (y % kDetTH == 0 || cls[p - W] != c || cls[p - W - 1] != c)) uf_union(J.parent, p, p - 1);
    }
}

__global__ __launch_bounds__(256) void det_flatten_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= J.h * J.w) return;
    int *parent = J.parent;
    if (parent[p] < 0) return;
    int r = p;
    while (parent[r] != r) r = parent[r];
    parent[p] = r;
}

// Every root (= first raster pixel of its component) claims a provisional id and records its sort key
// channel * npx + pixel: detections are ordered channel ascending, then by first pixel (skimage's
// label order inside a channel).
__global__ __launch_bounds__(256) void det_roots_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    const int npx = J.h * J.w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    if (J.parent[p] == p) {
        const int id = atomicAdd(J.counter, 1);
        J.prov[p] = id;
        J.keys[id] = (long long)(J.cls[p] - 1) * npx + p;
    }
}

// final id of provisional detection i = number of keys below its own (keys are distinct).  A handful to a few
// hundred detections is the normal case; a noisy class map can have tens of thousands of fragments, so the n^2
// comparisons are spread over ceil(n / 256) workgroups per image with the keys staged through LDS (the reference's
// labelling is linear in the fragment count; one 1024-thread workgroup would stall for seconds).  Also zeroes the
// statistics.
__global__ __launch_bounds__(256) void det_order_kernel(const DetJob *__restrict__ jobs) {
    __shared__ long long tile[1024];
    const DetJob J = jobs[blockIdx.y];
    const int n = J.n_det;
    if ((int)blockIdx.x * 256 >= n) return;
    const int npx = J.h * J.w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const long long k = i < n ? J.keys[i] : 0;
    int r = 0;
    for (int base = 0; base < n; base += 1024) {
        const int cnt = min(1024, n - base);
        for (int j = threadIdx.x; j < cnt; j += 256) tile[j] = J.keys[base + j];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) r += tile[j] < k;
        __syncthreads();
    }
    if (i < n) {
        J.newid[i] = r;
        J.channel[r] = (int32_t)(k / npx);
        J.area[i] = 0;
        J.score_bits[i] = 0;
    }
}

__global__ __launch_bounds__(256) void det_zero_masks_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    const size_t total = (size_t)J.n_det * J.h * J.w;
    if (total == 0) return;
    uint8_t *m = J.mask;
    // bytes up to the first 16-byte boundary, 16-byte stores for the body, bytes for the tail
    const size_t pre = std::min(total, (size_t)((16 - ((uintptr_t)m & 15)) & 15));
    const size_t n16 = (total - pre) / 16;
    uint4 *m16 = reinterpret_cast<uint4 *>(m + pre);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) m16[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < pre; i += 256) m[i] = 0;
        for (size_t i = pre + n16 * 16 + threadIdx.x; i < total; i += 256) m[i] = 0;
    }
}

// Pixel p belongs to detection newid[prov[root(p)]]: its mask byte, and area / max score per detection.  Scores are
// compared as int bit patterns: the reference takes max(score * mask), which is >= 0 whatever the scores are, and so
// is a maximum that starts from +0.
__global__ __launch_bounds__(256) void det_stats_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    if (J.n_det < 1) return;
    const int npx = J.h * J.w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int c = p < npx ? J.cls[p] : 0;
    const bool fg = c > 0;
    int d = -1, bits = 0;
    if (fg) {
        d = J.newid[J.prov[J.parent[p]]];
        J.mask[(long)d * npx + p] = 1;
        const float sc = J.rw_up[(long)(c - 1) * npx + p];
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
            atomicAdd(J.area + d0, __popcll(act));
            if (mx > 0) atomicMax(J.score_bits + d0, mx);
        }
    } else if (fg) {
        atomicAdd(J.area + d, 1);
        if (bits > 0) atomicMax(J.score_bits + d, bits);
    }
}

__global__ __launch_bounds__(256) void det_final_kernel(const DetJob *__restrict__ jobs) {
    const DetJob J = jobs[blockIdx.y];
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= J.n_det) return;
    J.score[d] = ((double)J.area[d] < J.min_area) ? 0.f : __int_as_float(J.score_bits[d]);   // statistics are kept by FINAL id
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

extern "C" int irn_find_centroids_batch(int n_images, const float *const *dp_dev, const int32_t *h, const int32_t *w,
                                        int iterations, int32_t *const *centroids_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_images < 1 || !dp_dev || !h || !w || !centroids_dev || iterations < 0)
        return fail(IRN_ERR_ARG, "irn_find_centroids_batch: bad argument");
    std::vector<CenJob> jobs(n_images);
    int max_n = 0;
    for (int i = 0; i < n_images; ++i) {
        if (!dp_dev[i] || !centroids_dev[i] || h[i] < 1 || w[i] < 1)
            return fail(IRN_ERR_ARG, "irn_find_centroids_batch: image %d: bad argument", i);
        jobs[i] = CenJob{dp_dev[i], centroids_dev[i], h[i], w[i]};
        max_n = std::max(max_n, h[i] * w[i]);
    }
    CenJob *jobs_dev = nullptr;
    int rc = scratch_upload(jobs.data(), sizeof(CenJob) * n_images, (void **)&jobs_dev, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(centroid_kernel, dim3(cdiv(max_n, 256), n_images), dim3(256), 0, stream, jobs_dev, iterations);
    IRN_LAUNCH_CHECK("centroid_kernel");
    return scratch_release(stream);
}

extern "C" int irn_find_centroids(const float *dp_dev, int h, int w, int iterations, int32_t *centroids_dev,
                                  void *stream_) {
    if (!dp_dev || !centroids_dev || h < 1 || w < 1 || iterations < 0)
        return fail(IRN_ERR_ARG, "irn_find_centroids: bad argument");
    const int32_t hh = h, ww = w;
    return irn_find_centroids_batch(1, &dp_dev, &hh, &ww, iterations, &centroids_dev, stream_);
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

// scratch of one image in irn_cluster_centroids[_batch]:
//   [parent npx][rank npx] int32, [picked npx] int32, [present npx+2] int32, [K slot of the single-image form]
//   (each 256-byte aligned)
extern "C" size_t irn_cluster_scratch_bytes(int h, int w) {
    if (h < 1 || w < 1) return 0;
    const size_t npx = (size_t)h * w;
    return round_up(4 * npx, 256) * 3 + round_up(4 * (npx + 2), 256) + 256;
}

extern "C" size_t irn_cluster_batch_scratch_bytes(int n_images, const int32_t *h, const int32_t *w) {
    if (n_images < 1 || !h || !w) return 0;
    size_t total = 0;
    for (int i = 0; i < n_images; ++i) total += irn_cluster_scratch_bytes(h[i], w[i]);
    return total;
}

extern "C" int irn_cluster_centroids_batch(int n_images, const int32_t *const *centroids_dev, const float *const *dp_dev,
                                           const int32_t *h, const int32_t *w, float thres,
                                           int32_t *const *cluster_map_dev, int32_t *k_dev, void *scratch_dev,
                                           void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_images < 1 || !centroids_dev || !dp_dev || !h || !w || !cluster_map_dev || !k_dev || !scratch_dev)
        return fail(IRN_ERR_ARG, "irn_cluster_centroids_batch: bad argument");
    std::vector<ClusterJob> jobs(n_images);
    char *s = (char *)scratch_dev;
    int max_n = 0;
    for (int i = 0; i < n_images; ++i) {
        if (!centroids_dev[i] || !dp_dev[i] || !cluster_map_dev[i] || h[i] < 1 || w[i] < 1)
            return fail(IRN_ERR_ARG, "irn_cluster_centroids_batch: image %d: bad argument", i);
        if ((long)h[i] * w[i] > (1L << 30)) return fail(IRN_ERR_ARG, "irn_cluster_centroids_batch: image too large");
        const size_t npx = (size_t)h[i] * w[i];
        const size_t a = round_up(4 * npx, 256);
        ClusterJob &J = jobs[i];
        J.centroids = centroids_dev[i];
        J.dp = dp_dev[i];
        J.cluster_map = cluster_map_dev[i];
        J.parent = (int *)s;
        J.rank = (int *)(s + a);
        J.picked = (int32_t *)(s + 2 * a);
        J.present = (int *)(s + 3 * a);
        J.k_out = k_dev + i;
        J.h = h[i];
        J.w = w[i];
        s += irn_cluster_scratch_bytes(h[i], w[i]);
        max_n = std::max(max_n, (int)npx);
    }
    ClusterJob *jd = nullptr;
    int rc = scratch_upload(jobs.data(), sizeof(ClusterJob) * n_images, (void **)&jd, stream);
    if (rc) return rc;
    const dim3 px(cdiv(max_n, 256), n_images);
    hipLaunchKernelGGL(cluster_init_kernel, px, dim3(256), 0, stream, jd, thres);
    IRN_LAUNCH_CHECK("cluster_init_kernel");
    hipLaunchKernelGGL(cluster_merge_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_merge_kernel");
    hipLaunchKernelGGL(cluster_flatten_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_flatten_kernel");
    hipLaunchKernelGGL(cluster_rank_kernel, dim3(n_images), dim3(1024), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_rank_kernel");
    hipLaunchKernelGGL(cluster_pick_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_pick_kernel");
    hipLaunchKernelGGL(cluster_compress_kernel, dim3(n_images), dim3(1024), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_compress_kernel");
    hipLaunchKernelGGL(cluster_remap_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("cluster_remap_kernel");
    return scratch_release(stream);
}

extern "C" int irn_cluster_centroids(const int32_t *centroids_dev, const float *dp_dev, int h, int w, float thres,
                                     int32_t *cluster_map_dev, int *k_out, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!centroids_dev || !dp_dev || !cluster_map_dev || !k_out || !scratch_dev || h < 1 || w < 1)
        return fail(IRN_ERR_ARG, "irn_cluster_centroids: bad argument");
    int32_t *k_dev = (int32_t *)((char *)scratch_dev + irn_cluster_scratch_bytes(h, w) - 256);   // the K slot
    const int32_t hh = h, ww = w;
    int rc = irn_cluster_centroids_batch(1, &centroids_dev, &dp_dev, &hh, &ww, thres, &cluster_map_dev, k_dev, scratch_dev,
                                         stream_);
    if (rc) return rc;
    int32_t k = 0;
    IRN_HIP_TRY(hipMemcpyAsync(&k, k_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    IRN_HIP_TRY(hipStreamSynchronize(stream));
    *k_out = k;
    return IRN_OK;
}

// scratch of one image in irn_detect_instance_*: [parent npx][prov npx][newid npx][area npx][score_bits npx] int32,
//                                                 [keys npx] int64, [counter]
extern "C" size_t irn_detect_scratch_bytes(int n_channels, int h, int w) {
    if (n_channels < 1 || h < 1 || w < 1) return 0;
    const size_t npx = (size_t)h * w;
    return round_up(4 * npx, 256) * 5 + round_up(8 * npx, 256) + 256;
}

extern "C" size_t irn_detect_batch_scratch_bytes(int n_images, const int32_t *n_channels, const int32_t *h,
                                                 const int32_t *w) {
    if (n_images < 1 || !n_channels || !h || !w) return 0;
    size_t total = 0;
    for (int i = 0; i < n_images; ++i) total += irn_detect_scratch_bytes(n_channels[i], h[i], w[i]);
    return total;
}

namespace {
// carve the per-image scratch; `counter` = the last 256-byte block (single-image API) unless the caller supplies one
void det_carve(DetJob &J, char *b, size_t npx) {
    const size_t a = round_up(4 * npx, 256);
    J.parent = (int *)b;
    J.prov = (int *)(b + a);
    J.newid = (int *)(b + 2 * a);
    J.area = (int *)(b + 3 * a);
    J.score_bits = (int *)(b + 4 * a);
    J.keys = (long long *)(b + 5 * a);
    J.counter = (int32_t *)(b + 5 * a + round_up(8 * npx, 256));
}

int det_fill_jobs(const char *who, int n_images, const float *const *rw_up_dev, const int32_t *const *argmax_dev,
                  const int32_t *n_channels, const int32_t *h, const int32_t *w, void *scratch_dev,
                  std::vector<DetJob> &jobs, int *max_n) {
    if (n_images < 1 || !rw_up_dev || !argmax_dev || !n_channels || !h || !w || !scratch_dev)
        return fail(IRN_ERR_ARG, "%s: bad argument", who);
    jobs.assign(n_images, DetJob{});
    char *s = (char *)scratch_dev;
    *max_n = 0;
    for (int i = 0; i < n_images; ++i) {
        if (!rw_up_dev[i] || !argmax_dev[i] || n_channels[i] < 1 || n_channels[i] > 65535 || h[i] < 1 || w[i] < 1)
            return fail(IRN_ERR_ARG, "%s: image %d: bad argument", who, i);
        if ((long)h[i] * w[i] > (1L << 30)) return fail(IRN_ERR_ARG, "%s: image too large", who);
        DetJob &J = jobs[i];
        J.rw_up = rw_up_dev[i];
        J.cls = argmax_dev[i];
        J.h = h[i];
        J.w = w[i];
        det_carve(J, s, (size_t)h[i] * w[i]);
        s += irn_detect_scratch_bytes(n_channels[i], h[i], w[i]);
        *max_n = std::max(*max_n, h[i] * w[i]);
    }
    return IRN_OK;
}
}  // namespace

extern "C" int irn_detect_instance_batch_count(int n_images, const float *const *rw_up_dev,
                                               const int32_t *const *argmax_dev, const int32_t *n_channels,
                                               const int32_t *h, const int32_t *w, int32_t *n_det_dev,
                                               void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<DetJob> jobs;
    int max_n = 0;
    int rc = det_fill_jobs("irn_detect_instance_batch_count", n_images, rw_up_dev, argmax_dev, n_channels, h, w,
                           scratch_dev, jobs, &max_n);
    if (rc) return rc;
    if (!n_det_dev) return fail(IRN_ERR_ARG, "irn_detect_instance_batch_count: null n_det_dev");
    for (int i = 0; i < n_images; ++i) jobs[i].counter = n_det_dev + i;
    DetJob *jd = nullptr;
    rc = scratch_upload(jobs.data(), sizeof(DetJob) * n_images, (void **)&jd, stream);
    if (rc) return rc;
    IRN_HIP_TRY(hipMemsetAsync(n_det_dev, 0, sizeof(int32_t) * n_images, stream));
    const dim3 px(cdiv(max_n, 256), n_images);
    int max_tiles = 0, max_border = 0;
    for (int i = 0; i < n_images; ++i) {
        const int tx = cdiv(w[i], kDetTW), ty = cdiv(h[i], kDetTH);
        max_tiles = std::max(max_tiles, tx * ty);
        max_border = std::max(max_border, (ty - 1) * w[i] + (tx - 1) * h[i]);
    }
    hipLaunchKernelGGL(det_local_kernel, dim3(max_tiles, n_images), dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_local_kernel");
    if (max_border > 0) {
        hipLaunchKernelGGL(det_border_kernel, dim3(cdiv(max_border, 256), n_images), dim3(256), 0, stream, jd);
        IRN_LAUNCH_CHECK("det_border_kernel");
    }
    hipLaunchKernelGGL(det_flatten_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_flatten_kernel");
    hipLaunchKernelGGL(det_roots_kernel, px, dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_roots_kernel");
    return scratch_release(stream);
}

extern "C" int irn_detect_instance_batch_emit(int n_images, const float *const *rw_up_dev,
                                              const int32_t *const *argmax_dev, const int32_t *n_channels,
                                              const int32_t *h, const int32_t *w, const int32_t *n_det,
                                              const double *min_area, float *const *score_dev,
                                              int32_t *const *channel_dev, uint8_t *const *mask_dev, void *scratch_dev,
                                              void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<DetJob> jobs;
    int max_n = 0;
    int rc = det_fill_jobs("irn_detect_instance_batch_emit", n_images, rw_up_dev, argmax_dev, n_channels, h, w,
                           scratch_dev, jobs, &max_n);
    if (rc) return rc;
    if (!n_det || !min_area || !score_dev || !channel_dev || !mask_dev)
        return fail(IRN_ERR_ARG, "irn_detect_instance_batch_emit: null argument");
    int max_det = 0;
    for (int i = 0; i < n_images; ++i) {
        DetJob &J = jobs[i];
        J.n_det = n_det[i];
        if (J.n_det < 0 || J.n_det > h[i] * w[i])
            return fail(IRN_ERR_ARG, "irn_detect_instance_batch_emit: image %d: n_det = %d", i, J.n_det);
        if (J.n_det == 0) continue;                       // nothing to write for this image
        if (!score_dev[i] || !channel_dev[i] || !mask_dev[i])
            return fail(IRN_ERR_ARG, "irn_detect_instance_batch_emit: image %d: null output", i);
        J.score = score_dev[i];
        J.channel = channel_dev[i];
        J.mask = mask_dev[i];
        J.min_area = min_area[i];
        max_det = std::max(max_det, J.n_det);
    }
    if (max_det == 0) return IRN_OK;
    DetJob *jd = nullptr;
    rc = scratch_upload(jobs.data(), sizeof(DetJob) * n_images, (void **)&jd, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(det_zero_masks_kernel, dim3(64, n_images), dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_zero_masks_kernel");
    hipLaunchKernelGGL(det_order_kernel, dim3(cdiv(max_det, 256), n_images), dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_order_kernel");
    hipLaunchKernelGGL(det_stats_kernel, dim3(cdiv(max_n, 256), n_images), dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_stats_kernel");
    hipLaunchKernelGGL(det_final_kernel, dim3(cdiv(max_det, 256), n_images), dim3(256), 0, stream, jd);
    IRN_LAUNCH_CHECK("det_final_kernel");
    return scratch_release(stream);
}

extern "C" int irn_detect_instance_count(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h,
                                         int w, int *n_det_out, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!rw_up_dev || !argmax_dev || !n_det_out || !scratch_dev || n_channels < 1 || n_channels > 65535 || h < 1 || w < 1)
        return fail(IRN_ERR_ARG, "irn_detect_instance_count: bad argument");
    const int32_t cc = n_channels, hh = h, ww = w;
    // the counter of the single-image form is the last block of its own scratch
    int32_t *counter = (int32_t *)((char *)scratch_dev + irn_detect_scratch_bytes(n_channels, h, w) - 256);
    int rc = irn_detect_instance_batch_count(1, &rw_up_dev, &argmax_dev, &cc, &hh, &ww, counter, scratch_dev, stream_);
    if (rc) return rc;
    int32_t n_det = 0;
    IRN_HIP_TRY(hipMemcpyAsync(&n_det, counter, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    IRN_HIP_TRY(hipStreamSynchronize(stream));
    *n_det_out = n_det;
    return IRN_OK;
}

extern "C" int irn_detect_instance_emit(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h, int w,
                                        int n_det, double min_area, float *score_dev, int32_t *channel_dev,
                                        uint8_t *mask_dev, void *scratch_dev, void *stream_) {
    if (!rw_up_dev || !argmax_dev || !score_dev || !channel_dev || !mask_dev || !scratch_dev || n_channels < 1 || h < 1 ||
        w < 1 || n_det < 1)
        return fail(IRN_ERR_ARG, "irn_detect_instance_emit: bad argument");
    const int32_t cc = n_channels, hh = h, ww = w, nd = n_det;
    return irn_detect_instance_batch_emit(1, &rw_up_dev, &argmax_dev, &cc, &hh, &ww, &nd, &min_area, &score_dev,
                                          &channel_dev, &mask_dev, scratch_dev, stream_);
}
