// Path-max affinity kernel (gfx950).
//
// Replaces reference misc/indexing.py:91-109 (`edge_to_affinity`: 9-20 index_select + max_pool2d
// launches over int64 index tensors) by one kernel that stages an edge tile with its radial halo in
// LDS and walks the (dy,dx) path table itself.  Two instantiations:
//   POW=false : aff = 1 - max(edge over path)          (API form, reference channel order)
//   POW=true  : w   = fp32(aff ** beta)                (walk weight table, raster plane order;
//               fuses the Hadamard power of misc/indexing.py:133)
//
// Roofline: per source pixel the kernel reads n_cells LDS words (242 at r=5, 2134 at r=10) and writes
// n_dirs floats (34 / 152) — one pass, LDS-issue bound; it runs once per image against 2^exp_times
// sweeps of the walk, so it is <2 % of the path.
#include <utility>

#include "kernels.hpp"

namespace irn {

namespace {

constexpr int AFF_TH = kAffTileH;   // source rows per workgroup
constexpr int AFF_TW = kAffTileW;   // source cols per workgroup (one wave covers two rows)

__device__ __forceinline__ float pow_beta(float a, float beta, int beta_int) {
    // torch.pow(fp32, beta) is a <=1-ulp powf; the fp64 power rounded once to fp32 is the
    // correctly rounded value and so agrees with it to 1 ulp.
    double b = (double)a;
    if (beta_int > 0) {
        double r = 1.0;
        int e = beta_int;
        while (e) {
            if (e & 1) r *= b;
            b *= b;
            e >>= 1;
        }
        return (float)r;
    }
    return (float)pow(b, (double)beta);
}

__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <bool POW>
__global__ __launch_bounds__(256) void affinity_kernel(const AffJob *__restrict__ jobs,
                                                       const int *__restrict__ dir_start8,
                                                       const int *__restrict__ cell_off8, int n_dirs,
                                                       int radius, float beta, int beta_int) {
    extern __shared__ float tile[];
    const AffJob J = jobs[blockIdx.y];
    const int tiles_x = (J.sw + AFF_TW - 1) / AFF_TW;
    const int tiles_y = (J.sh + AFF_TH - 1) / AFF_TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int halo = radius - 1;
    const int LW = AFF_TW + 2 * halo;
    const int LH = AFF_TH + halo;
    const int ty0 = ((int)blockIdx.x / tiles_x) * AFF_TH;
    const int tx0 = ((int)blockIdx.x % tiles_x) * AFF_TW;

    // stage the edge tile: rows [ty0, ty0+TH+halo), cols [tx0-halo, tx0+TW+halo) of the source
    // rectangle, translated into the grid; anything outside the grid is a boundary (edge = 1),
    // which is what the reference's F.pad(..., value=1.0) supplies (misc/indexing.py:150).
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = J.oy + ty0 + ly;
        const int gx = J.ox + tx0 + lx - halo;
        float v = 1.0f;
        if (gy < J.gh && gx >= 0 && gx < J.gw) v = J.edge[(long)gy * J.gw + gx];
        tile[i] = v;
    }
    __syncthreads();

    const int ly = threadIdx.x / AFF_TW, lx = threadIdx.x % AFF_TW;
    const int sy = ty0 + ly, sx = tx0 + lx;
    const bool valid = sy < J.sh && sx < J.sw;
    const int base = ly * LW + lx + halo;
    float *out = J.out + (long)sy * J.sw + sx;

    // Eight path cells per scalar load of their (wave-uniform) LDS offsets.  The first version
    // fetched dy and dx of one cell at a time: two scalar loads, a multiply and a wait that also
    // drained the LDS reads per cell — latency-bound, ~10x off the LDS/VALU bound of the kernel.
    const float *tb = tile + base;
    for (int d = 0; d < n_dirs; ++d) {
        const int k0 = dir_start8[d], k1 = dir_start8[d + 1];
        float m0 = -INFINITY, m1 = -INFINITY;
        for (int k = k0; k < k1; k += 8) {
            const int4 oa = *reinterpret_cast<const int4 *>(cell_off8 + k);
            const int4 ob = *reinterpret_cast<const int4 *>(cell_off8 + k + 4);
            const float a0 = tb[oa.x], a1 = tb[oa.y], a2 = tb[oa.z], a3 = tb[oa.w];
            const float b0 = tb[ob.x], b1 = tb[ob.y], b2 = tb[ob.z], b3 = tb[ob.w];
            // v_max3_f32 directly: fmaxf() on values fresh from memory is preceded by a canonicalising v_max_f32 v, v, v
            // each (8 extra vector instructions per 8 cells in a loop the vector pipe bounds)
            m0 = max3(max3(m0, a0, a1), a2, a3);
            m1 = max3(max3(m1, b0, b1), b2, b3);
        }
        const float m = max3(m0, m1, m1);
        float a = 1.0f - m;
        if (POW) a = pow_beta(a, beta, beta_int);
        if (valid) out[(long)d * J.plane_stride] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// The same kernel with the path table as a compile-time constant (radius 5 and 10, raster plane order: the walk's
// weight table).  In the table-driven loop above every LDS read costs a vector add for its address (the cell's offset
// arrives in a scalar register) and the loop is bound by the vector pipe (profiles/r02_s27_affinity_counters.txt: 11.5 k
// vector instructions per wave for 2.6 k LDS reads).  Unrolled, the cell offsets are the immediate offsets of the
// ds_read instructions: 2134 reads + 1067 v_max3 + 152 powers and stores of straight-line code at radius 10 (~40 KB,
// inside the instruction cache, every wave running the same stream).
// ------------------------------------------------------------------------------------------------
template <int R>
struct Paths {
    static constexpr int kMaxDirs = 2 * R * R, kMaxCells = 32 * R * R;
    int n_dirs = 0, n_cells = 0;
    signed char dy[kMaxDirs] = {}, dx[kMaxDirs] = {};
    short start[kMaxDirs + 1] = {};
    signed char cy[kMaxCells] = {}, cx[kMaxCells] = {};
    constexpr void add(int y, int x) {
        // thick segment (0,0) -> (y,x): lattice points of the bounding box with (y*px - x*py)^2 < y^2 + x^2
        // (misc/indexing.py:37-46); the max over a path does not depend on the order of its cells
        dy[n_dirs] = (signed char)y;
        dx[n_dirs] = (signed char)x;
        const int lsq = y * y + x * x;
        const int x_lo = x < 0 ? x : 0, x_hi = x < 0 ? 0 : x;
        for (int py = 0; py <= y; ++py)
            for (int px = x_lo; px <= x_hi; ++px) {
                const int cross = y * px - x * py;
                if (cross * cross < lsq) {
                    cy[n_cells] = (signed char)py;
                    cx[n_cells] = (signed char)px;
                    ++n_cells;
                }
            }
        start[++n_dirs] = (short)n_cells;
    }
    constexpr Paths() {
        // raster order of the directions = discovery order of misc/indexing.py:24-30 (path table order 1)
        for (int x = 1; x < R; ++x) add(0, x);
        for (int y = 1; y < R; ++y)
            for (int x = -R + 1; x < R; ++x)
                if (x * x + y * y < R * R) add(y, x);
    }
};
template <int R>
inline constexpr Paths<R> kPaths{};

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int R, int D>
__device__ __forceinline__ float path_max(const float *tb) {
    constexpr int LW = AFF_TW + 2 * (R - 1), HALO = R - 1;
    constexpr int k0 = kPaths<R>.start[D], n = kPaths<R>.start[D + 1] - k0;
    constexpr auto off = [](int k) constexpr { return kPaths<R>.cy[k] * LW + kPaths<R>.cx[k] + HALO; };
    float m = tb[off(k0)];
    static_for<(n - 1) / 2>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = k0 + 1 + 2 * decltype(ik)::value;
        m = max3(m, tb[off(k)], tb[off(k + 1)]);
    });
    if constexpr ((n - 1) % 2 == 1) {
        const float v = tb[off(k0 + n - 1)];
        m = max3(m, v, v);
    }
    return m;
}

template <int R>
__global__ __launch_bounds__(256) void affinity_unrolled_kernel(const AffJob *__restrict__ jobs, int beta_int) {
    constexpr int HALO = R - 1, LW = AFF_TW + 2 * HALO, LH = AFF_TH + HALO, ND = kPaths<R>.n_dirs;
    __shared__ float tile[LH * LW];
    const AffJob J = jobs[blockIdx.y];
    const int tiles_x = (J.sw + AFF_TW - 1) / AFF_TW;
    const int tiles_y = (J.sh + AFF_TH - 1) / AFF_TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int ty0 = ((int)blockIdx.x / tiles_x) * AFF_TH;
    const int tx0 = ((int)blockIdx.x % tiles_x) * AFF_TW;
    for (int i = threadIdx.x; i < LH * LW; i += 256) {          // staging as in affinity_kernel
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = J.oy + ty0 + ly;
        const int gx = J.ox + tx0 + lx - HALO;
        float v = 1.0f;
        if (gy < J.gh && gx >= 0 && gx < J.gw) v = J.edge[(long)gy * J.gw + gx];
        tile[i] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x / AFF_TW, lx = threadIdx.x % AFF_TW;
    const int sy = ty0 + ly, sx = tx0 + lx;
    const bool valid = sy < J.sh && sx < J.sw;
    const float *tb = tile + ly * LW + lx;              // cell (cy, cx) of this pixel sits at tb[cy * LW + cx + HALO]
    float *out = J.out + (long)sy * J.sw + sx;
    static_for<ND>([&](auto id) __attribute__((always_inline)) {
        constexpr int d = decltype(id)::value;
        // integer beta only (the launch sends any other beta to the table-driven kernel): the general fp64 pow() inlined
        // once per direction would be 300 KB of code
        double b = (double)(1.0f - path_max<R, d>(tb)), r = 1.0;
        for (int e = beta_int; e; e >>= 1) {
            if (e & 1) r *= b;
            b *= b;
        }
        if (valid) *out = (float)r;
        out += J.plane_stride;
    });
}

// Backward of edge_to_affinity for the training seam (reference net/resnet50_irn.py:162-175: index_select
// + max_pool2d over the path axis; autograd sends the gradient of aff[d,s] to the FIRST path cell that
// attains the maximum, negated because aff = 1 - max).  The forward's edge tile is rebuilt in LDS and the
// arg-max is recomputed (nothing is saved by the forward); gradients of a workgroup are gathered in an
// LDS tile with ds_add_f32 and flushed with one global atomic per touched cell, so overlapping halos of
// neighbouring workgroups meet in HBM only once per cell instead of once per (pixel, direction).
__global__ __launch_bounds__(256) void affinity_backward_kernel(const AffJob *__restrict__ jobs,
                                                                const float *const *__restrict__ grad_aff,
                                                                float *const *__restrict__ grad_edge,
                                                                const int *__restrict__ dir_start,
                                                                const int *__restrict__ cell_dy,
                                                                const int *__restrict__ cell_dx, int n_dirs, int radius) {
    extern __shared__ float tile[];
    const AffJob J = jobs[blockIdx.y];
    const int tiles_x = (J.sw + AFF_TW - 1) / AFF_TW;
    const int tiles_y = (J.sh + AFF_TH - 1) / AFF_TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int halo = radius - 1;
    const int LW = AFF_TW + 2 * halo;
    const int LH = AFF_TH + halo;
    float *gtile = tile + LH * LW;
    const int ty0 = ((int)blockIdx.x / tiles_x) * AFF_TH;
    const int tx0 = ((int)blockIdx.x % tiles_x) * AFF_TW;
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = J.oy + ty0 + ly;
        const int gx = J.ox + tx0 + lx - halo;
        float v = 1.0f;
        if (gy < J.gh && gx >= 0 && gx < J.gw) v = J.edge[(long)gy * J.gw + gx];
        tile[i] = v;
        gtile[i] = 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / AFF_TW, lx = threadIdx.x % AFF_TW;
    const int sy = ty0 + ly, sx = tx0 + lx;
    const bool valid = sy < J.sh && sx < J.sw;
    const int base = ly * LW + lx + halo;
    const float *ga = grad_aff[blockIdx.y] + (long)sy * J.sw + sx;
    if (valid) {
        for (int d = 0; d < n_dirs; ++d) {
            const int k0 = dir_start[d], k1 = dir_start[d + 1];
            float m = -INFINITY;
            int arg = 0;
            for (int k = k0; k < k1; ++k) {
                const int off = cell_dy[k] * LW + cell_dx[k];
                const float v = tile[base + off];
                if (v > m) {     // strict: the first maximum keeps the gradient, like max_pool2d
                    m = v;
                    arg = off;
                }
            }
            atomicAdd(&gtile[base + arg], -ga[(long)d * J.plane_stride]);
        }
    }
    __syncthreads();
    float *ge = grad_edge[blockIdx.y];
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const float g = gtile[i];
        if (g == 0.f) continue;
        const int py = i / LW, px = i - py * LW;
        const int gy = J.oy + ty0 + py;
        const int gx = J.ox + tx0 + px - halo;
        if (gy < J.gh && gx >= 0 && gx < J.gw) unsafeAtomicAdd(ge + (long)gy * J.gw + gx, g);
    }
}

}  // namespace

int launch_affinity(const AffJob *jobs_dev, int n_jobs, int max_sh, int max_sw, const DeviceTable &tab,
                    bool with_pow, float beta, hipStream_t stream) {
    const int tiles = cdiv(max_sh, AFF_TH) * cdiv(max_sw, AFF_TW);
    const int halo = tab.radius - 1;
    const size_t lds = sizeof(float) * (AFF_TH + halo) * (AFF_TW + 2 * halo);
    dim3 grid(tiles, n_jobs);
    int beta_int = 0;
    if (with_pow && beta == (float)(int)beta && beta >= 1.0f && beta <= 64.0f) beta_int = (int)beta;
    if (with_pow && beta_int > 0 && tab.order == 1 && tab.radius == 10)
        hipLaunchKernelGGL(affinity_unrolled_kernel<10>, grid, dim3(256), 0, stream, jobs_dev, beta_int);
    else if (with_pow && beta_int > 0 && tab.order == 1 && tab.radius == 5)
        hipLaunchKernelGGL(affinity_unrolled_kernel<5>, grid, dim3(256), 0, stream, jobs_dev, beta_int);
    else if (with_pow)
        hipLaunchKernelGGL(affinity_kernel<true>, grid, dim3(256), lds, stream, jobs_dev, tab.dir_start8,
                           tab.cell_off8, tab.n_dirs, tab.radius, beta, beta_int);
    else
        hipLaunchKernelGGL(affinity_kernel<false>, grid, dim3(256), lds, stream, jobs_dev, tab.dir_start8,
                           tab.cell_off8, tab.n_dirs, tab.radius, beta, beta_int);
    IRN_LAUNCH_CHECK("affinity_kernel");
    return IRN_OK;
}

}  // namespace irn

// ------------------------------------------------------------------------------------------------
// C ABI: path table + edge_to_affinity
// ------------------------------------------------------------------------------------------------
using namespace irn;

extern "C" int irn_version(void) { return 100; }

extern "C" const char *irn_last_error(void) { return last_error_slot().c_str(); }

extern "C" int irn_path_count(int radius, int *n_dirs, int *n_cells) {
    if (radius < 2 || radius > IRN_MAX_RADIUS || !n_dirs || !n_cells)
        return fail(IRN_ERR_ARG, "irn_path_count: radius must be in [2,%d]", IRN_MAX_RADIUS);
    PathTable t = build_path_table(radius, 0);
    *n_dirs = t.n_dirs();
    *n_cells = t.n_cells();
    return IRN_OK;
}

extern "C" int irn_path_table(int radius, int order, int32_t *dst_dydx, int32_t *path_start,
                              int32_t *cells_dydx) {
    if (radius < 2 || radius > IRN_MAX_RADIUS || (order != 0 && order != 1) || !dst_dydx || !path_start ||
        !cells_dydx)
        return fail(IRN_ERR_ARG, "irn_path_table: bad argument");
    PathTable t = build_path_table(radius, order);
    for (int i = 0; i < t.n_dirs(); ++i) {
        dst_dydx[2 * i] = t.dy[i];
        dst_dydx[2 * i + 1] = t.dx[i];
    }
    for (int i = 0; i <= t.n_dirs(); ++i) path_start[i] = t.start[i];
    for (int i = 0; i < t.n_cells(); ++i) {
        cells_dydx[2 * i] = t.cy[i];
        cells_dydx[2 * i + 1] = t.cx[i];
    }
    return IRN_OK;
}

extern "C" int irn_edge_to_affinity(const float *edge_dev, int batch, int hp, int wp, int radius,
                                    float *aff_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!edge_dev || !aff_dev || batch < 1 || radius < 2 || radius > IRN_MAX_RADIUS)
        return fail(IRN_ERR_ARG, "irn_edge_to_affinity: bad argument");
    const int rf = radius - 1;
    const int sh = hp - rf, sw = wp - 2 * rf;
    if (sh < 1 || sw < 1)
        return fail(IRN_ERR_ARG, "irn_edge_to_affinity: grid %dx%d too small for radius %d", hp, wp, radius);
    const DeviceTable *tab = nullptr;
    int rc = get_device_table(radius, 0, &tab);
    if (rc) return rc;
    // job descriptors live in a small device buffer owned by the table cache (library-private)
    std::vector<AffJob> jobs(batch);
    const long ns = (long)sh * sw;
    for (int b = 0; b < batch; ++b) {
        AffJob &j = jobs[b];
        j.edge = edge_dev + (long)b * hp * wp;
        j.out = aff_dev + (long)b * tab->n_dirs * ns;
        j.gh = hp; j.gw = wp; j.oy = 0; j.ox = rf; j.sh = sh; j.sw = sw;
        j.plane_stride = ns;
    }
    AffJob *jobs_dev = nullptr;
    rc = scratch_upload(jobs.data(), sizeof(AffJob) * batch, (void **)&jobs_dev, stream);
    if (rc) return rc;
    rc = launch_affinity(jobs_dev, batch, sh, sw, *tab, false, 0.f, stream);
    if (rc) return rc;
    return scratch_release(stream);
}

extern "C" int irn_edge_to_affinity_backward(const float *edge_dev, const float *grad_aff_dev, int batch, int hp, int wp,
                                             int radius, float *grad_edge_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!edge_dev || !grad_aff_dev || !grad_edge_dev || batch < 1 || radius < 2 || radius > IRN_MAX_RADIUS)
        return fail(IRN_ERR_ARG, "irn_edge_to_affinity_backward: bad argument");
    const int rf = radius - 1;
    const int sh = hp - rf, sw = wp - 2 * rf;
    if (sh < 1 || sw < 1)
        return fail(IRN_ERR_ARG, "irn_edge_to_affinity_backward: grid %dx%d too small for radius %d", hp, wp, radius);
    const DeviceTable *tab = nullptr;
    int rc = get_device_table(radius, 0, &tab);
    if (rc) return rc;
    const long ns = (long)sh * sw;
    // descriptors + the two pointer arrays in one upload
    std::vector<char> host(sizeof(AffJob) * batch + 2 * sizeof(void *) * batch);
    AffJob *jobs = (AffJob *)host.data();
    const float **gas = (const float **)(host.data() + sizeof(AffJob) * batch);
    float **ges = (float **)(host.data() + sizeof(AffJob) * batch + sizeof(void *) * batch);
    for (int b = 0; b < batch; ++b) {
        AffJob &j = jobs[b];
        j.edge = edge_dev + (long)b * hp * wp;
        j.out = nullptr;
        j.gh = hp; j.gw = wp; j.oy = 0; j.ox = rf; j.sh = sh; j.sw = sw;
        j.plane_stride = ns;
        gas[b] = grad_aff_dev + (long)b * tab->n_dirs * ns;
        ges[b] = grad_edge_dev + (long)b * hp * wp;
    }
    char *dev = nullptr;
    rc = scratch_upload(host.data(), host.size(), (void **)&dev, stream);
    if (rc) return rc;
    IRN_HIP_TRY(hipMemsetAsync(grad_edge_dev, 0, sizeof(float) * (size_t)batch * hp * wp, stream));
    const int tiles = cdiv(sh, AFF_TH) * cdiv(sw, AFF_TW);
    const size_t lds = 2 * sizeof(float) * (AFF_TH + rf) * (AFF_TW + 2 * rf);
    hipLaunchKernelGGL(affinity_backward_kernel, dim3(tiles, batch), dim3(256), lds, stream, (const AffJob *)dev,
                       (const float *const *)(dev + sizeof(AffJob) * batch),
                       (float *const *)(dev + sizeof(AffJob) * batch + sizeof(void *) * batch), tab->dir_start, tab->cell_dy,
                       tab->cell_dx, tab->n_dirs, radius);
    IRN_LAUNCH_CHECK("affinity_backward_kernel");
    return scratch_release(stream);
}

namespace {
// ---------------------------------------------------------------------------------------------
// Pair displacement of the training seam (AffinityDisplacementLoss.to_pair_displacement,
// net/resnet50_irn.py:177-193): out[b,c,d,y,x] = disp[b,c,y,rf+x] - disp[b,c,y+dy_d,rf+x+dx_d] over the
// cropped grid — |S| shifted slices, a stack and a subtraction in the reference; one pass here, exact.
// Backward as a gather (no atomics): the gradient of a cell is the sum over d of what it received as a
// source minus what it received as the destination of the cell at (-dy_d, -dx_d).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_disp_kernel(const float *__restrict__ disp, float *__restrict__ out, int hp,
                                                        int wp, int rf, int n_dirs, const int *__restrict__ dir_dy,
                                                        const int *__restrict__ dir_dx) {
    const int ch = hp - rf, cw = wp - 2 * rf;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= ch * cw) return;
    const int y = s / cw, x = s - y * cw;
    const float *plane = disp + (size_t)blockIdx.y * hp * wp;
    float *o = out + (size_t)blockIdx.y * n_dirs * ch * cw + s;
    const float src = plane[y * wp + rf + x];
    for (int d = 0; d < n_dirs; ++d)
        o[(size_t)d * ch * cw] = src - plane[(y + dir_dy[d]) * wp + rf + x + dir_dx[d]];
}

__global__ __launch_bounds__(256) void pair_disp_backward_kernel(const float *__restrict__ gout, float *__restrict__ gdisp,
                                                                 int hp, int wp, int rf, int n_dirs,
                                                                 const int *__restrict__ dir_dy, const int *__restrict__ dir_dx) {
    const int ch = hp - rf, cw = wp - 2 * rf;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hp * wp) return;
    const int Y = p / wp, X = p - Y * wp;
    const float *g = gout + (size_t)blockIdx.y * n_dirs * ch * cw;
    const int xs = X - rf;
    float acc = 0.f;
    if (Y < ch && xs >= 0 && xs < cw)
        for (int d = 0; d < n_dirs; ++d) acc += g[(size_t)d * ch * cw + Y * cw + xs];
    for (int d = 0; d < n_dirs; ++d) {
        const int yy = Y - dir_dy[d], xx = xs - dir_dx[d];
        if (yy >= 0 && yy < ch && xx >= 0 && xx < cw) acc -= g[(size_t)d * ch * cw + yy * cw + xx];
    }
    gdisp[(size_t)blockIdx.y * hp * wp + p] = acc;
}

int pair_disp_args(const char *who, const void *a, const void *b, int batch, int channels, int hp, int wp, int radius,
                   const DeviceTable **tab) {
    if (!a || !b || batch < 1 || channels < 1 || radius < 2 || radius > IRN_MAX_RADIUS)
        return fail(IRN_ERR_ARG, "%s: bad argument", who);
    if (hp - (radius - 1) < 1 || wp - 2 * (radius - 1) < 1)
        return fail(IRN_ERR_ARG, "%s: grid %dx%d too small for radius %d", who, hp, wp, radius);
    if ((long)batch * channels > 65535) return fail(IRN_ERR_ARG, "%s: batch * channels must be <= 65535", who);
    return get_device_table(radius, 0, tab);
}
}  // namespace

extern "C" int irn_pair_displacement(const float *disp_dev, int batch, int channels, int hp, int wp, int radius,
                                     float *out_dev, void *stream) {
    const DeviceTable *tab = nullptr;
    if (int rc = pair_disp_args("irn_pair_displacement", disp_dev, out_dev, batch, channels, hp, wp, radius, &tab)) return rc;
    const int rf = radius - 1;
    hipLaunchKernelGGL(pair_disp_kernel, dim3(cdiv((hp - rf) * (wp - 2 * rf), 256), batch * channels), dim3(256), 0,
                       (hipStream_t)stream, disp_dev, out_dev, hp, wp, rf, tab->n_dirs, tab->dir_dy, tab->dir_dx);
    IRN_LAUNCH_CHECK("pair_disp_kernel");
    return IRN_OK;
}

extern "C" int irn_pair_displacement_backward(const float *grad_out_dev, int batch, int channels, int hp, int wp,
                                              int radius, float *grad_disp_dev, void *stream) {
    const DeviceTable *tab = nullptr;
    if (int rc = pair_disp_args("irn_pair_displacement_backward", grad_out_dev, grad_disp_dev, batch, channels, hp, wp,
                                radius, &tab))
        return rc;
    hipLaunchKernelGGL(pair_disp_backward_kernel, dim3(cdiv(hp * wp, 256), batch * channels), dim3(256), 0,
                       (hipStream_t)stream, grad_out_dev, grad_disp_dev, hp, wp, radius - 1, tab->n_dirs, tab->dir_dy,
                       tab->dir_dx);
    IRN_LAUNCH_CHECK("pair_disp_backward_kernel");
    return IRN_OK;
}
