// Inter-pixel affinity random walk as sparse stencil sweeps (gfx950).
//
// Replaces reference misc/indexing.py:112-165: the reference densifies the affinity into an
// (hw x hw) matrix on the host (:117-127), powers and column-normalises it (:133-135), squares it
// exp_times times with dense sgemm (:136-137, 70 TFLOP per 512^2 image) and multiplies the CAMs in
// (:164).  The operator has only 2|S|+1 non-zeros per row (|S| = 34 at radius 5, 152 at radius 10),
// so the same product is 2^exp_times applications of
//
//     x'[c,p] = ( x[c,p] + sum_{d in S} w_d(p) x[c,p+d] + w_d(p-d) x[c,p-d] ) / deg(p)
//
// with w_d(p) = (1 - max_{path(d)} edge)^beta stored ONCE per unordered pixel pair (the matrix is
// symmetric) as |S| planes of the image, and deg(p) = 1 + sum of the 2|S| weights touching p.
//
// Numerics (tests/test_precision_model.py): fp32 FMAs inside one neighbour row (<= 2R-1 terms),
// row partials combined and normalised in fp64, state stored fp32.  This is as close to the exact
// operator as full fp64 accumulation (1.5e-6) while the bulk arithmetic stays fp32; plain fp32
// accumulation drifts 1.8e-4 over 256 sweeps and misses the 1e-4 parity bar.
//
// Schedule (round 3): x . T^n is a polynomial of degree n in the operator; T is similar to a symmetric matrix
// (T = A D^-1 with A symmetric, non-negative, unit diagonal), so its spectrum is real and inside [-1, 1], and
//     lambda^n = sum_k c_k T_k(lambda),   c_k = 2^(1-n) C(n, (n-k)/2)  (k = n mod 2; c_0 halved)
// has coefficients that fall off like exp(-k^2 / 2n): for n = 256 the terms beyond k = 84 sum to < 1e-7.  With
// |T_k| <= 1 on the spectrum, dropping them changes the result by less than that (in the D^1/2-weighted norm) —
// below the 1.5e-6 that storing the state in fp32 costs the plain iteration.  So the walk runs the three-term
// recurrence   y_0 = x_0,  y_1 = T y_0,  y_{t+1} = 2 T y_t - y_{t-1},   s = sum_k c_k y_k   for K ~ sqrt(2 n ln(1/tol))
// operator applications instead of n (84 instead of 256); measured against the fp64 oracle it is as close as the
// plain iteration or closer (fewer roundings of the state).  Option "accel" = 0 restores the plain powers (the
// recurrence with a = 1, b = 0 and the unit series, bit-identical to rounds 1-2); schedules that would not be
// shorter (small n) are plain powers anyway.  chebyshev_power_series below.
//
// HBM layout per image (all in the caller's workspace):
//   weights  |S| planes, plane d = [front_pad zeros][h*w floats][tail], plane_stride floats apart.
//            front_pad >= (R-1)*w + (R-1) makes every "-d" read w_d(p-d) in range: rows above the
//            image land in the zero pad, and a column that leaves the image wraps onto a pixel
//            whose own +d neighbour is outside the image, i.e. onto a stored zero.
//   inv_deg  fp64 [h*w]
//   xa, xb   fp32 [C, h*w] ping-pong state (y_t)
//   xc       fp32 [C, h*w] the series sum s_t (accelerated schedule only)
//
// Roofline of one sweep launch: streams the weight planes once from HBM (4*|S|*N bytes) plus
// 8*C*N of state -> HBM-bound for C <= ~27 at radius 10.  The "-d" reads and tile
// halos re-read bytes another workgroup already pulled; the block->XCD mapping keeps all tiles of
// an image on one XCD so those hit its L2 instead of HBM.
#include <algorithm>
#include <cmath>
#include <mutex>

#include "walk_ctx.hpp"

namespace irn {


namespace {

// ---------------------------------------------------------------------------------------------
// small per-pixel kernels (grid.y = image)
// ---------------------------------------------------------------------------------------------

// Zeroes what the affinity kernel does not write: every plane's front pad and tail.
__global__ __launch_bounds__(256) void zero_pad_kernel(const WalkImg *__restrict__ imgs, int n_dirs) {
    const WalkImg I = imgs[blockIdx.y];
    const int d = blockIdx.x;
    if (d >= n_dirs) return;
    float *plane = I.wts + (long)d * I.plane_stride;
    for (int i = threadIdx.x; i < I.front_pad; i += 256) plane[-1 - i] = 0.f;
    const long n = (long)I.h * I.w;
    const long end = I.plane_stride - I.front_pad;
    for (long i = n + threadIdx.x; i < end; i += 256) plane[i] = 0.f;
}

// deg(p) = 1 + sum_d [ w_d(p) + w_d(p-d) ] in fp64; stores 1/deg.  (Column sum of
// misc/indexing.py:135; the unit diagonal of :123-126 keeps it >= 1.)
__global__ __launch_bounds__(256) void degree_kernel(const WalkImg *__restrict__ imgs,
                                                     const int *__restrict__ dir_dy,
                                                     const int *__restrict__ dir_dx, int n_dirs) {
    const WalkImg I = imgs[blockIdx.y];
    const long n = (long)I.h * I.w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int y = (int)(p / I.w), x = (int)(p - (long)y * I.w);
    double deg = 1.0;
    for (int d = 0; d < n_dirs; ++d) {
        const int dy = dir_dy[d], dx = dir_dx[d];
        const float *plane = I.wts + (long)d * I.plane_stride;
        deg += (double)plane[p];
        const int yy = y - dy, xx = x - dx;
        if (yy >= 0 && xx >= 0 && xx < I.w) deg += (double)plane[p - ((long)dy * I.w + dx)];
    }
    I.inv_deg[p] = 1.0 / deg;
}

// x0 = cam * (1 - edge)  (misc/indexing.py:162), optionally split by instance
// (step/make_ins_seg_labels.py:77-80: channel cls*K+k = cam[cls] * (inst == k)).
__global__ __launch_bounds__(256) void x0_kernel(const WalkImg *__restrict__ imgs, int to_out, int cheb, float c0) {
    const WalkImg I = imgs[blockIdx.y];
    const long n = (long)I.h * I.w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float one_minus = 1.0f - I.edge[p];
    const int k = I.inst ? I.k_inst : 1;
    const int id = I.inst ? I.inst[p] : 0;
    float *dst = to_out ? I.out : I.xa;
    for (int c = 0; c < I.C; ++c) {
        const int cls = c / k, kk = c - cls * k;
        float v = I.cam[(long)cls * n + p];
        if (I.inst) v = v * (id == kk ? 1.0f : 0.0f);
        dst[(long)c * n + p] = v * one_minus;
        if (cheb) I.xc[(long)c * n + p] = c0 * (v * one_minus);
    }
}

// ---------------------------------------------------------------------------------------------
// generic sweep: any radius, one pixel per thread, table-driven, full fp64 accumulation.
// Correctness fallback (radius other than 5/10, images narrower than the radius) and the
// on-device cross-check of the blocked kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_generic_kernel(const WalkImg *__restrict__ imgs,
                                                            const int *__restrict__ dir_dy,
                                                            const int *__restrict__ dir_dx, int n_dirs,
                                                            int phase, int last, int cheb, float ck) {
    const WalkImg I = imgs[blockIdx.y];
    const long n = (long)I.h * I.w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float *src = (phase & 1) ? I.xb : I.xa;
    float *dst = (last && !cheb) ? I.out : ((phase & 1) ? I.xa : I.xb);
    const int y = (int)(p / I.w), x = (int)(p - (long)y * I.w);
    const double inv = I.inv_deg[p];
    for (int c = 0; c < I.C; ++c) {
        const float *xc = src + (long)c * n;
        double acc = (double)xc[p];
        for (int d = 0; d < n_dirs; ++d) {
            const int dy = dir_dy[d], dx = dir_dx[d];
            const long off = (long)dy * I.w + dx;
            const float *plane = I.wts + (long)d * I.plane_stride;
            if (y + dy < I.h && x + dx >= 0 && x + dx < I.w) acc += (double)plane[p] * (double)xc[p + off];
            if (y - dy >= 0 && x - dx >= 0 && x - dx < I.w) acc += (double)plane[p - off] * (double)xc[p - off];
        }
        if (!cheb) {
            dst[(long)c * n + p] = (float)(acc * inv);
        } else {
            // cheb 1: y_1 = T y_0; cheb 2: y_{t+1} = 2 T y_t - y_{t-1} (y_{t-1} is what the ping-pong target still holds)
            const float y = cheb == 1 ? (float)(acc * inv) : (float)(2.0 * (acc * inv) - (double)dst[(long)c * n + p]);
            const float sn = fmaf(ck, y, I.xc[(long)c * n + p]);
            I.xc[(long)c * n + p] = sn;
            if (last) I.out[(long)c * n + p] = sn;
            else dst[(long)c * n + p] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// blocked sweep for radius R in {5, 10}
//
// One workgroup (256 threads) owns a TH x TW tile of one image for CH channels; a thread owns P
// consecutive pixels of a row.  The state tile + halo is staged in LDS.  For every stored direction
// d = (dy,dx) the thread issues the FORWARD load w_d(p) and the BACKWARD load w_d(p-d) back to
// back: both come from the same plane, shifted by d, so all but dy rows / |dx| columns of the
// second load hit in L1/L2 instead of HBM (visiting neighbour rows in raster order instead puts
// ~one block lifetime between the two touches of a plane and doubles the HBM traffic).
// ---------------------------------------------------------------------------------------------
typedef float f4a __attribute__((ext_vector_type(4)));
typedef float f2a __attribute__((ext_vector_type(2)));

typedef const float IRN_GLOBAL *gcf_t;
typedef float IRN_GLOBAL *gf_t;
typedef const double IRN_GLOBAL *gcd_t;

template <int R, int P, int TH, int TW>
struct Geo {
    static_assert((TW / P) * TH == 256, "tile must map onto 256 threads");
    static constexpr int H = R - 1;                              // halo
    static constexpr int LH = TH + 2 * H;
    static constexpr int WIN = ((P + 2 * H + P - 1) / P) * P;    // state window per thread per row
    static constexpr int LW = ((TW - P + WIN + 3) / 4) * 4;      // LDS row length
};

struct BlockEnt {
    int img, ty0, tx0, c0;   // img < 0 : idle block
};

// P consecutive weights of one plane through the buffer resource: wave-uniform base in the
// descriptor, plane/row offset in an SGPR, lane offset in one VGPR -> no per-load VALU address math.
template <int P>
__device__ __forceinline__ void load_w(float (&w)[P], __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    if constexpr (P == 4) {
        const f4a v = __builtin_bit_cast(f4a, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else if constexpr (P == 2) {
        const f2a v = __builtin_bit_cast(f2a, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
        w[0] = v.x; w[1] = v.y;
    } else {
        w[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
}

template <int P, int WIN>
__device__ __forceinline__ void load_window(float (&xw)[WIN], const float *row) {
    if constexpr (P == 4) {
#pragma unroll
        for (int q = 0; q < WIN / 4; ++q) {
            const f4a v = reinterpret_cast<const f4a *>(row)[q];
            xw[4 * q] = v.x; xw[4 * q + 1] = v.y; xw[4 * q + 2] = v.z; xw[4 * q + 3] = v.w;
        }
    } else if constexpr (P == 2) {
#pragma unroll
        for (int q = 0; q < WIN / 2; ++q) {
            const f2a v = reinterpret_cast<const f2a *>(row)[q];
            xw[2 * q] = v.x; xw[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < WIN; ++q) xw[q] = row[q];
    }
}

template <int R, int CH, int P, int TH, int TW>
__device__ __forceinline__ void sweep_body(const WalkImg &I, const BlockEnt &B, const float *src_,
                                           float *dst_, const int *__restrict__ plane_tab, float *xs, int last, int cheb,
                                           float ck) {
    using G = Geo<R, P, TH, TW>;
    constexpr int H = G::H, LH = G::LH, LW = G::LW, WIN = G::WIN;
    const int tid = threadIdx.x;
    const int h = I.h, w = I.w;
    const unsigned n = (unsigned)(h * w);
    const gcf_t src = (gcf_t)src_ + (size_t)B.c0 * n;
    const gf_t dst = (gf_t)dst_ + (size_t)B.c0 * n;

    // ---- stage state tile + halo for CH channels (zero outside the image) ----
    // Loads are gathered eight at a time into registers before the LDS writes so that they are in
    // flight together (one load -> one ds_write per iteration serialises on the L2 latency).
    {
        constexpr int TOT = CH * LH * LW;
        constexpr int NIT = (TOT + 255) / 256;
        constexpr int GRP = 8;
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += GRP) {
            float v[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int i = tid + (it0 + k) * 256;
                const int c = i / (LH * LW);
                const int r = i - c * (LH * LW);
                const int ly = r / LW, lx = r - ly * LW;
                const int gy = B.ty0 - H + ly, gx = B.tx0 - H + lx;
                const bool ok = i < TOT && gy >= 0 && gy < h && gx >= 0 && gx < w;
                v[k] = ok ? src[(unsigned)c * n + (unsigned)(gy * w + gx)] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int i = tid + (it0 + k) * 256;
                if (i < TOT) xs[i] = v[k];
            }
        }
    }
    __syncthreads();

    const int trow = tid / (TW / P);
    const int tcol = (tid % (TW / P)) * P;
    const int y = B.ty0 + trow, x = B.tx0 + tcol;
    const bool live = y < h && x < w;
    const unsigned p0 = live ? (unsigned)(y * w + x) : 0u;   // lane offset (elements) into a plane
    // record 0 of the resource is the start of plane 0's front pad
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(I.wts - I.front_pad), 0, (int)(I.n_dirs * I.plane_stride * 4), 0x00020000);
    const int voff = (int)(p0 * 4u);
    const int ps4 = (int)(I.plane_stride * 4);
    const int fp4 = I.front_pad * 4;
    const int w4 = w * 4;

    double acc[CH][P];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < P; ++j) acc[c][j] = (double)xs[(c * LH + trow + H) * LW + tcol + H + j];

    // Rows of stored directions are a RUNTIME loop (the 2H+1 column offsets inside are expanded):
    // the body stays a few KB and lives in the instruction cache.  The fully expanded form (10-20k
    // instructions of straight-line code) was instruction-fetch bound: 50-100 us per workgroup
    // regardless of how many workgroups were resident (profiles/r01_s2_*).  Offsets (dy,dx) that
    // are not in the direction set do not branch: they load the nearest in-set plane of the row (hot
    // in cache) and scale the weight by 0, so the 2(2H+1) loads of a row are issued together.  (A
    // dedicated all-zero plane for those slots cost +40 % fabric traffic, profiles/r01_s3_*.)
    const float *xrow = &xs[(trow + H) * LW + tcol];
#pragma unroll 1
    for (int dy = 0; dy <= H; ++dy) {
        float pf[CH][P], pb[CH][P];                        // fp32 partials: forward / backward neighbours
        float xf[CH][WIN], xb[CH][WIN];                    // state windows of rows y+dy and y-dy
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            load_window<P, WIN>(xf[c], xrow + (c * LH + dy) * LW);
            load_window<P, WIN>(xb[c], xrow + (c * LH - dy) * LW);
#pragma unroll
            for (int j = 0; j < P; ++j) pf[c][j] = pb[c][j] = 0.f;
        }
        const int rowoff4 = dy * w4;                        // wave-uniform, bytes
        const int *prow = plane_tab + dy * (2 * H + 1);
        static_for<2 * H + 1>([&](auto ix) __attribute__((always_inline)) {
            constexpr int dx = decltype(ix)::value - H;
            const int ent = prow[dx + H];                                 // scalar table entry
            const int soff = fp4 + (ent < 0 ? ~ent : ent) * ps4;
            const float keep = ent < 0 ? 0.f : 1.f;
            float wf[P], wb[P];
            load_w<P>(wf, wrsrc, voff, soff);                             // w_d(p)    pairs p with p+d
            load_w<P>(wb, wrsrc, voff, soff - rowoff4 - dx * 4);          // w_d(p-d)  pairs p with p-d
#pragma unroll
            for (int j = 0; j < P; ++j) {
                wf[j] *= keep;
                wb[j] *= keep;
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    pf[c][j] = fmaf(wf[j], xf[c][H + dx + j], pf[c][j]);
                    pb[c][j] = fmaf(wb[j], xb[c][H - dx + j], pb[c][j]);
                }
        });
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                acc[c][j] += (double)pf[c][j];
                acc[c][j] += (double)pb[c][j];
            }
    }

    if (!live) return;
    const gcd_t inv_deg = (gcd_t)I.inv_deg;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if (x + j < w) {
            const double inv = inv_deg[p0 + j];
            if (!cheb) {
#pragma unroll
                for (int c = 0; c < CH; ++c) dst[(unsigned)c * n + p0 + j] = (float)(acc[c][j] * inv);
            } else {
                // accelerated schedule: y_{t+1} = 2 T y_t - y_{t-1} (the ping-pong target still holds y_{t-1};
                // cheb == 1 is the first step, y_1 = T y_0), s += c_{t+1} y_{t+1}; the last step hands out s
                const gf_t sacc = (gf_t)I.xc + (size_t)B.c0 * n;
                const gf_t out = (gf_t)I.out + (size_t)B.c0 * n;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned o = (unsigned)c * n + p0 + j;
                    const float y = cheb == 1 ? (float)(acc[c][j] * inv)
                                              : (float)(2.0 * (acc[c][j] * inv) - (double)dst[o]);
                    const float sn = fmaf(ck, y, sacc[o]);
                    sacc[o] = sn;
                    if (last) out[o] = sn;
                    else dst[o] = y;
                }
            }
        }
    }
}

// MAXW caps the occupancy the register allocator aims for: with a low cap it spends registers on
// keeping 10+ weight loads in flight per wave instead of squeezing into 64 VGPRs with 2-3 in flight.
template <int R, int CH, int P, int TH, int TW, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void sweep_blocked_kernel(const WalkImg *__restrict__ imgs,
                                                            const int4 *__restrict__ block_map,
                                                            const int *__restrict__ plane_tab, int phase,
                                                            int last, int cheb, float ck) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int4 e = block_map[blockIdx.x];
    if (e.x < 0) return;
    const WalkImg I = imgs[e.x];
    const BlockEnt B{e.x, e.y, e.z, e.w & 0xffff};
    const float *src = (phase & 1) ? I.xb : I.xa;
    float *dst = (last && !cheb) ? I.out : ((phase & 1) ? I.xa : I.xb);
    sweep_body<R, CH, P, TH, TW>(I, B, src, dst, plane_tab, xs, last, cheb, ck);
}

// The one tile shape of the blocked sweep: 8 x 128 pixels per 256-thread workgroup, 4 pixels per thread, at most 2 waves per
// SIMD — the best of the twelve shapes measured in rounds 1-3 (profiles/README.md); the others went with round 6's prune.
constexpr int kTileP = 4, kTileH = 8, kTileW = 128, kTileMaxW = 2;
constexpr int kMaxChunk = 4;      // channels of an image a workgroup carries at once (its LDS state window per channel)

template <int R, int CH>
int launch_sweep(const WalkImg *imgs, const int4 *map, const int *ptab, int nb, int phase, int last, int cheb, float ck,
                 hipStream_t stream) {
    using G = Geo<R, kTileP, kTileH, kTileW>;
    const size_t lds = sizeof(float) * CH * G::LH * G::LW;
    hipLaunchKernelGGL((sweep_blocked_kernel<R, CH, kTileP, kTileH, kTileW, kTileMaxW>), dim3(nb), dim3(256), lds, stream, imgs, map, ptab,
                       phase, last, cheb, ck);
    IRN_LAUNCH_CHECK("sweep_blocked_kernel");
    return IRN_OK;
}

}  // namespace
}  // namespace irn

// ------------------------------------------------------------------------------------------------
// context + C ABI
// ------------------------------------------------------------------------------------------------
using namespace irn;


// lambda^n in the Chebyshev basis (header comment, "Schedule"): c_k = 2^(1-n) C(n, (n-k)/2) for k = n (mod 2), c_0
// halved; the coefficients are positive and sum to 1 (lambda = 1), so the dropped tail is 1 - (partial sum).
int irn::chebyshev_power_series(int n, double tol, std::vector<double> *coef, bool *cheb) {
    std::vector<double> c((size_t)n + 1, 0.0);
    for (int k = n & 1; k <= n; k += 2) {
        const int j = (n - k) / 2;
        const double lc = std::lgamma((double)n + 1.0) - std::lgamma((double)j + 1.0) - std::lgamma((double)(n - j) + 1.0) +
                          (1.0 - (double)n) * 0.6931471805599453094;
        c[k] = std::exp(lc) * (k == 0 ? 0.5 : 1.0);
    }
    int K = n;
    double head = 0.0;
    for (int k = 0; k <= n; ++k) {
        head += c[k];
        if (1.0 - head <= tol) {
            K = k;
            break;
        }
    }
    if (n < 8 || K + 2 >= n) {               // not shorter: plain powers, "series" = y_n alone
        coef->assign((size_t)n + 1, 0.0);
        (*coef)[n] = 1.0;
        *cheb = false;
        return n;
    }
    c.resize((size_t)K + 1);
    // the kept coefficients are renormalised to sum to 1: a constant field (the stationary direction of a uniform
    // operator, lambda = 1) stays exactly constant, and the truncation error is spread instead of one-sided
    for (double &v : c) v /= head;
    *coef = c;
    *cheb = true;
    return K;
}

int irn::walk_schedule(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream) {
    if (ctx->sched_n == n_sweeps && ctx->sched_accel == ctx->accel && ctx->sched_tol == ctx->accel_tol_exp) return IRN_OK;
    bool cheb = false;
    std::vector<double> coef;
    int steps = n_sweeps;
    if (ctx->accel) {
        steps = chebyshev_power_series(n_sweeps, std::pow(10.0, -(double)ctx->accel_tol_exp), &coef, &cheb);
    } else {
        coef.assign((size_t)n_sweeps + 1, 0.0);
        coef[n_sweeps] = 1.0;
    }
    if (steps + 1 > ctx->coef_cap) {
        // the table of the previous schedule may still be read by a run in flight
        if (ctx->run_done_ev && ctx->last_valid) IRN_HIP_TRY(hipEventSynchronize(ctx->run_done_ev));
        if (ctx->coef_dev) (void)hipFree(ctx->coef_dev);
        ctx->coef_dev = nullptr;
        IRN_HIP_TRY(hipMalloc((void **)&ctx->coef_dev, sizeof(float) * (size_t)(steps + 1)));
        ctx->coef_cap = steps + 1;
    } else if (ctx->run_done_ev && ctx->last_valid) {
        IRN_HIP_TRY(hipEventSynchronize(ctx->run_done_ev));
    }
    ctx->sched_coef_f.resize(coef.size());
    for (size_t i = 0; i < coef.size(); ++i) ctx->sched_coef_f[i] = (float)coef[i];
    IRN_HIP_TRY(hipMemcpyAsync(ctx->coef_dev, ctx->sched_coef_f.data(), sizeof(float) * coef.size(), hipMemcpyHostToDevice, stream));
    IRN_HIP_TRY(hipStreamSynchronize(stream));      // pageable source; a schedule changes once per (n_sweeps, options), not per run
    ctx->sched_coef = coef;
    ctx->sched_steps = steps;
    ctx->sched_cheb = cheb;
    ctx->sched_n = n_sweeps;
    ctx->sched_accel = ctx->accel;
    ctx->sched_tol = ctx->accel_tol_exp;
    return IRN_OK;
}

extern "C" int irn_power_series(int n, int tol_exp, double *coef_out, int coef_cap, int *n_steps, int *recurrence) {
    if (n < 0 || n > (1 << 20) || tol_exp < 4 || tol_exp > 12 || !n_steps || !recurrence)
        return fail(IRN_ERR_ARG, "irn_power_series: bad argument");
    bool cheb = false;
    std::vector<double> coef;
    *n_steps = chebyshev_power_series(n, std::pow(10.0, -(double)tol_exp), &coef, &cheb);
    *recurrence = cheb ? 1 : 0;
    if (coef_out) {
        if (coef_cap < (int)coef.size()) return fail(IRN_ERR_ARG, "irn_power_series: %zu coefficients, room for %d", coef.size(), coef_cap);
        std::copy(coef.begin(), coef.end(), coef_out);
    }
    return IRN_OK;
}

extern "C" int irn_walk_steps(irn_walk_ctx *ctx, int n_sweeps, int *n_steps) {
    if (!ctx || !n_steps || n_sweeps < 0) return fail(IRN_ERR_ARG, "irn_walk_steps: bad argument");
    bool cheb = false;
    std::vector<double> coef;
    *n_steps = ctx->accel ? chebyshev_power_series(n_sweeps, std::pow(10.0, -(double)ctx->accel_tol_exp), &coef, &cheb) : n_sweeps;
    return IRN_OK;
}

extern "C" int irn_walk_create(int radius, irn_walk_ctx **ctx_out) {
    if (!ctx_out || radius < 2 || radius > IRN_MAX_RADIUS)
        return fail(IRN_ERR_ARG, "irn_walk_create: radius must be in [2,%d]", IRN_MAX_RADIUS);
    const DeviceTable *tab = nullptr;
    int rc = get_device_table(radius, 1, &tab);
    if (rc) return rc;
    irn_walk_ctx *c = new irn_walk_ctx();
    c->radius = radius;
    c->tab = tab;
    // radius 5 / 10: weights-stationary persistent walk (falls back to the streaming sweeps per batch
    // when an image does not fit one round); other radii: generic table-driven sweep
    c->variant = (radius == 5 || radius == 10) ? 2 : 0;
    // radius 5: an image is 16 tiles and normally sits inside one XCD; tiles verify that per image inside the
    // kernel and then exchange through the XCD's L2 (plain stores) instead of the fabric
    c->res_plain_store = radius == 5;
    *ctx_out = c;
    return IRN_OK;
}

extern "C" int irn_walk_destroy(irn_walk_ctx *ctx) {
    if (!ctx) return IRN_OK;
    if (ctx->imgs_dev) (void)hipFree(ctx->imgs_dev);
    if (ctx->jobs_dev) (void)hipFree(ctx->jobs_dev);
    if (ctx->map_dev) (void)hipFree(ctx->map_dev);
    if (ctx->coef_dev) (void)hipFree(ctx->coef_dev);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 3; ++k) {
        if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
        if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->run_done_ev) (void)hipEventDestroy(ctx->run_done_ev);
    resident_destroy(ctx);
    for (int k = 0; k < 2; ++k) {
        if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    delete ctx;
    return IRN_OK;
}

extern "C" int irn_walk_set_option(irn_walk_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return fail(IRN_ERR_ARG, "irn_walk_set_option: null argument");
    if (!strcmp(name, "variant")) {
        if (value == 1 && !(ctx->radius == 5 || ctx->radius == 10))
            return fail(IRN_ERR_ARG, "blocked sweep exists for radius 5 and 10 only");
        if (value == 2 && !resident_supported(ctx))
            return fail(IRN_ERR_ARG, "resident walk exists for radius 5 and 10 only");
        if (value < 0 || value > 2) return fail(IRN_ERR_ARG, "variant must be 0, 1 or 2");
        ctx->variant = value;
    } else if (!strcmp(name, "sweeps_per_launch")) {
        if (value < 0) return fail(IRN_ERR_ARG, "sweeps_per_launch must be >= 0");
        ctx->res_sweeps_per_launch = value;
    } else if (!strcmp(name, "plain_store")) {
        if (value && ctx->radius != 5)
            return fail(IRN_ERR_ARG, "plain_store needs radius 5 (all tiles of an image inside one XCD)");
        ctx->res_plain_store = value != 0;
    } else if (!strcmp(name, "poll_delay_plain")) {
        if (value < 0 || value > 1000) return fail(IRN_ERR_ARG, "poll_delay_plain must be in [0,1000]");
        ctx->res_poll_delay_plain = value;
    } else if (!strcmp(name, "poll_delay")) {
        if (value < 0 || value > 1000) return fail(IRN_ERR_ARG, "poll_delay must be in [0,1000]");
        ctx->res_poll_delay = value;
        ctx->res_poll_auto = 0;                         // pinned by the caller: no start-up probe
        return IRN_OK;
    } else if (!strcmp(name, "poll_delay_auto")) {
        ctx->res_poll_auto = value ? 1 : 0;
        return IRN_OK;
    } else if (!strcmp(name, "accel")) {
        ctx->accel = value ? 1 : 0;
        return IRN_OK;                                  // the schedule is rebuilt by the next run; no re-configure
    } else if (!strcmp(name, "accel_tol_exp")) {
        if (value < 4 || value > 12) return fail(IRN_ERR_ARG, "accel_tol_exp must be in [4,12] (truncation bound 10^-value)");
        ctx->accel_tol_exp = value;
        return IRN_OK;
    } else if (!strcmp(name, "cooperative")) {
        ctx->res_cooperative = value ? 1 : 0;
        ctx->res_coop_refused = false;
    } else if (!strcmp(name, "inject_timeout")) {      // test hook: the next persistent launch gives up at once
        ctx->res_inject_timeout = value ? 1 : 0;
        return IRN_OK;                                  // no re-configure
    } else if (!strcmp(name, "profile")) {
        if (value && !ctx->res_prof_dev) {
            IRN_HIP_TRY(hipMalloc((void **)&ctx->res_prof_dev, 2 * 256 * 4 * sizeof(long long)));
            IRN_HIP_TRY(hipMemset(ctx->res_prof_dev, 0, 2 * 256 * 4 * sizeof(long long)));
        } else if (!value && ctx->res_prof_dev) {
            (void)hipFree(ctx->res_prof_dev);
            ctx->res_prof_dev = nullptr;
        }
    } else {
        return fail(IRN_ERR_ARG, "irn_walk_set_option: unknown option '%s'", name);
    }
    ctx->n = 0;   // force re-configure
    return IRN_OK;
}

extern "C" int irn_walk_enable_timing(irn_walk_ctx *ctx, int enable) {
    if (!ctx) return fail(IRN_ERR_ARG, "null ctx");
    ctx->timing = enable ? 1 : 0;
    return IRN_OK;
}

extern "C" int irn_walk_last_sweep_ms(irn_walk_ctx *ctx, float *ms, int *n_launches) {
    if (!ctx || !ms || !n_launches) return fail(IRN_ERR_ARG, "null argument");
    if (ctx->ev_used == 0) return fail(IRN_ERR_STATE, "no timed run recorded");
    float total = 0.f;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float t = 0.f;
        IRN_HIP_TRY(hipEventSynchronize(ctx->ev_pool[i + 1]));
        IRN_HIP_TRY(hipEventElapsedTime(&t, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        total += t;
    }
    *ms = total;
    *n_launches = ctx->pending_launches;
    ctx->ev_used = 0;
    ctx->pending_launches = 0;
    return IRN_OK;
}

extern "C" int irn_walk_configure(irn_walk_ctx *ctx, int n_images, const int32_t *h, const int32_t *w,
                                  const int32_t *c, size_t *workspace_bytes) {
    if (!ctx || !h || !w || !c || !workspace_bytes || n_images < 1)
        return fail(IRN_ERR_ARG, "irn_walk_configure: bad argument");
    const int R = ctx->radius;
    // the tables rewritten below (block map, resident job list, descriptors) may still be read by the previous run
    // when the caller's stream is not the legacy default stream
    if (ctx->run_done_ev && ctx->last_valid) IRN_HIP_TRY(hipEventSynchronize(ctx->run_done_ev));
    ctx->last_valid = false;
    ctx->n = 0;
    ctx->h.assign(h, h + n_images);
    ctx->w.assign(w, w + n_images);
    ctx->c.assign(c, c + n_images);
    ctx->off_wts.resize(n_images);
    ctx->off_deg.resize(n_images);
    ctx->off_xa.resize(n_images);
    ctx->off_xb.resize(n_images);
    ctx->off_xc.resize(n_images);
    ctx->plane_stride.resize(n_images);
    ctx->front_pad.resize(n_images);
    size_t off = 0;
    ctx->max_h = ctx->max_w = ctx->max_n = 0;
    bool blocked_ok = ctx->variant >= 1;
    for (int i = 0; i < n_images; ++i) {
        if (h[i] < 1 || w[i] < 1 || c[i] < 1 || c[i] > 0xffff)
            return fail(IRN_ERR_ARG, "irn_walk_configure: image %d has invalid size %dx%d c=%d", i, h[i], w[i], c[i]);
        const size_t npx = (size_t)h[i] * w[i];
        const int fp = (int)round_up((size_t)(R - 1) * w[i] + (R - 1), 64);
        const long ps = (long)round_up(fp + npx + 8, 64);
        ctx->front_pad[i] = fp;
        ctx->plane_stride[i] = ps;
        ctx->off_wts[i] = off;
        off += round_up(sizeof(float) * ps * ctx->tab->n_dirs, 256);
        ctx->off_deg[i] = off;
        off += round_up(sizeof(double) * npx, 256);
        ctx->off_xa[i] = off;
        // 8 bytes per pixel and channel: the resident kernel keeps {tag, value} granules here
        off += round_up(8 * npx * c[i] + 64, 256);
        ctx->off_xb[i] = off;
        off += round_up(8 * npx * c[i] + 64, 256);
        ctx->off_xc[i] = off;
        off += round_up(8 * npx * c[i] + 64, 256);
        ctx->max_h = std::max(ctx->max_h, (int)h[i]);
        ctx->max_w = std::max(ctx->max_w, (int)w[i]);
        ctx->max_n = std::max(ctx->max_n, (int)npx);
        if (w[i] < R) blocked_ok = false;   // the wrap-onto-a-zero argument needs w >= R
    }
    ctx->ws_bytes = off;
    ctx->all_blocked_ok = blocked_ok;

    // ---- block maps of the blocked sweep, one slice per channel-chunk width (1..4) ----
    // Entry = (image, tile row0, tile col0, first channel).  Inside a slice, block b is meant for
    // XCD b % 8 (observed dispatch order; speed only): every image is pinned to one XCD so that the
    // halo / "-d" re-reads of its weight planes hit that XCD's L2.
    std::vector<int4> map;
    for (int k = 0; k < 5; ++k) ctx->cls_begin[k] = ctx->cls_count[k] = 0;
    ctx->max_nch = 1;
    if (blocked_ok) {
        const int NX = 8;
        for (int cls = 4; cls >= 1; --cls) {          // widest first: its workgroups run longest
            while (map.size() % NX) map.push_back(make_int4(-1, 0, 0, 0));   // keep block b on XCD b % 8
            ctx->cls_begin[cls] = (int)map.size();
            std::vector<std::vector<int4>> q(NX);
            std::vector<size_t> load(NX, 0);
            for (int i = 0; i < n_images; ++i) {
                int best = 0;
                for (int k = 1; k < NX; ++k)
                    if (load[k] < load[best]) best = k;
                for (int c0 = 0; c0 < c[i]; c0 += kMaxChunk) {
                    const int nch = std::min(kMaxChunk, c[i] - c0);
                    if (nch != cls) continue;
                    ctx->max_nch = std::max(ctx->max_nch, nch);
                    for (int ty = 0; ty < h[i]; ty += kTileH)
                        for (int tx = 0; tx < w[i]; tx += kTileW) {
                            q[best].push_back(make_int4(i, ty, tx, c0 | (nch << 16)));
                            ++load[best];
                        }
                }
            }
            size_t depth = 0;
            for (auto &v : q) depth = std::max(depth, v.size());
            const size_t begin = map.size();
            for (size_t sl = 0; sl < depth; ++sl)
                for (int k = 0; k < NX; ++k) map.push_back(sl < q[k].size() ? q[k][sl] : make_int4(-1, 0, 0, 0));
            while (map.size() > begin && map.back().x < 0) map.pop_back();
            ctx->cls_count[cls] = (int)(map.size() - begin);
        }
    }
    ctx->map_len = (int)map.size();

    if (n_images > ctx->cap_imgs) {
        if (ctx->imgs_dev) (void)hipFree(ctx->imgs_dev);
        if (ctx->jobs_dev) (void)hipFree(ctx->jobs_dev);
        ctx->imgs_dev = nullptr;
        ctx->jobs_dev = nullptr;
        IRN_HIP_TRY(hipMalloc((void **)&ctx->imgs_dev, sizeof(WalkImg) * n_images));
        IRN_HIP_TRY(hipMalloc((void **)&ctx->jobs_dev, sizeof(AffJob) * n_images));
        ctx->cap_imgs = n_images;
    }
    const size_t stage_need = (sizeof(WalkImg) + sizeof(AffJob)) * (size_t)n_images;
    if (stage_need > ctx->stage_cap) {
        for (int k = 0; k < 2; ++k) {
            if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
            ctx->stage[k] = nullptr;
            IRN_HIP_TRY(hipHostMalloc(&ctx->stage[k], stage_need, hipHostMallocDefault));
            if (!ctx->stage_ev[k]) IRN_HIP_TRY(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
        }
        ctx->stage_cap = stage_need;
    }
    if ((int)map.size() > ctx->cap_map) {
        if (ctx->map_dev) (void)hipFree(ctx->map_dev);
        ctx->map_dev = nullptr;
        IRN_HIP_TRY(hipMalloc((void **)&ctx->map_dev, sizeof(int4) * map.size()));
        ctx->cap_map = (int)map.size();
    }
    if (!map.empty())
        IRN_HIP_TRY(hipMemcpy(ctx->map_dev, map.data(), sizeof(int4) * map.size(), hipMemcpyHostToDevice));
    if (ctx->variant == 2) {
        const int rc = resident_configure(ctx);
        if (rc) return rc;
    } else {
        ctx->res_ok = false;
    }
    ctx->n = n_images;
    *workspace_bytes = ctx->ws_bytes;
    return IRN_OK;
}

template <int R, int CH>
static int launch_blocked_cls(irn_walk_ctx *ctx, int phase, int last, int cheb, float ck, hipStream_t stream) {
    const int nb = ctx->cls_count[CH];
    if (nb <= 0) return IRN_OK;
    return launch_sweep<R, CH>(ctx->imgs_dev, ctx->map_dev + ctx->cls_begin[CH], ctx->tab->plane_tab, nb,
                               phase, last, cheb, ck, stream);
}

// One sweep = one launch per channel-chunk width present in the batch.  The widths touch disjoint
// images, so each runs its whole chain of sweeps on its own stream (st[CH-1]); a width with a handful
// of workgroups then overlaps the big one instead of costing a full kernel latency per sweep.
template <int R>
static int launch_blocked(irn_walk_ctx *ctx, int phase, int last, int cheb, float ck, hipStream_t const *st) {
    int rc = launch_blocked_cls<R, 1>(ctx, phase, last, cheb, ck, st[0]);
    if (!rc) rc = launch_blocked_cls<R, 2>(ctx, phase, last, cheb, ck, st[1]);
    if (!rc) rc = launch_blocked_cls<R, 3>(ctx, phase, last, cheb, ck, st[2]);
    if (!rc) rc = launch_blocked_cls<R, 4>(ctx, phase, last, cheb, ck, st[3]);
    return rc;
}

static int timing_events(irn_walk_ctx *ctx, hipEvent_t *ev0, hipEvent_t *ev1) {
    while (ctx->ev_pool.size() < ctx->ev_used + 2) {
        hipEvent_t e;
        IRN_HIP_TRY(hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    *ev0 = ctx->ev_pool[ctx->ev_used];
    *ev1 = ctx->ev_pool[ctx->ev_used + 1];
    return IRN_OK;
}

// degree + x_0 + n_sweeps streaming sweeps of the configured batch; the weight planes are already in the workspace
// and the descriptors in imgs_dev.  The path of variants 0/1, of batches the persistent kernel cannot take, and the
// re-run of irn_walk_sync after a persistent launch gave up.
int irn::streaming_run(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream, bool timed) {
    const DeviceTable &tab = *ctx->tab;
    const int n = ctx->n;
    const int pix_blocks = cdiv(ctx->max_n, 256);
    {
        const int rc_s = walk_schedule(ctx, n_sweeps, stream);
        if (rc_s) return rc_s;
    }
    const bool cheb = ctx->sched_cheb;
    const int n_steps = ctx->sched_steps;
    hipLaunchKernelGGL(degree_kernel, dim3(pix_blocks, n), dim3(256), 0, stream, ctx->imgs_dev, tab.dir_dy, tab.dir_dx,
                       tab.n_dirs);
    IRN_LAUNCH_CHECK("degree_kernel");
    ctx->deg_stale = false;
    hipLaunchKernelGGL(x0_kernel, dim3(pix_blocks, n), dim3(256), 0, stream, ctx->imgs_dev, n_sweeps == 0 ? 1 : 0,
                       cheb ? 1 : 0, cheb ? (float)ctx->sched_coef[0] : 0.f);
    IRN_LAUNCH_CHECK("x0_kernel");

    const bool blocked = ctx->variant >= 1 && ctx->all_blocked_ok;
    // stream of every channel-chunk class: the class with most workgroups stays on the caller's stream
    hipStream_t st[4] = {stream, stream, stream, stream};
    int n_side = 0;
    if (blocked && n_steps > 0) {
        int big = 1;
        for (int k = 2; k <= 4; ++k)
            if (ctx->cls_count[k] > ctx->cls_count[big]) big = k;
        for (int k = 1; k <= 4; ++k) {
            if (k == big || ctx->cls_count[k] == 0) continue;
            if (!ctx->side[n_side]) {
                IRN_HIP_TRY(hipStreamCreateWithFlags(&ctx->side[n_side], hipStreamNonBlocking));
                IRN_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join[n_side], hipEventDisableTiming));
            }
            st[k - 1] = ctx->side[n_side++];
        }
        if (n_side && !ctx->ev_fork) IRN_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int rc = IRN_OK;
    if (timed) {
        rc = timing_events(ctx, &ev0, &ev1);
        if (rc) return rc;
        IRN_HIP_TRY(hipEventRecord(ev0, stream));
    }
    if (n_side) {
        IRN_HIP_TRY(hipEventRecord(ctx->ev_fork, stream));
        for (int k = 0; k < n_side; ++k) IRN_HIP_TRY(hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    }
    for (int t = 0; t < n_steps; ++t) {
        const int last = (t == n_steps - 1) ? 1 : 0;
        const int mode = cheb ? (t == 0 ? 1 : 2) : 0;                   // plain power / first / later recurrence step
        const float ck = cheb ? (float)ctx->sched_coef[t + 1] : 0.f;    // coefficient of y_{t+1} in the series
        if (blocked) {
            rc = ctx->radius == 5 ? launch_blocked<5>(ctx, t, last, mode, ck, st) : launch_blocked<10>(ctx, t, last, mode, ck, st);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(sweep_generic_kernel, dim3(pix_blocks, n), dim3(256), 0, stream, ctx->imgs_dev,
                               tab.dir_dy, tab.dir_dx, tab.n_dirs, t, last, mode, ck);
            IRN_LAUNCH_CHECK("sweep_generic_kernel");
        }
    }
    for (int k = 0; k < n_side; ++k) {
        IRN_HIP_TRY(hipEventRecord(ctx->ev_join[k], ctx->side[k]));
        IRN_HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_join[k], 0));
    }
    if (timed) {
        IRN_HIP_TRY(hipEventRecord(ev1, stream));
        ctx->ev_used += 2;
        ctx->pending_launches += n_steps;
    }
    return IRN_OK;
}

extern "C" int irn_walk_run(irn_walk_ctx *ctx, const float *const *edge_dev, const float *const *cam_dev,
                            const int32_t *const *inst_map_dev, const int32_t *k_inst,
                            float *const *out_dev, float beta, int n_sweeps, void *workspace_dev,
                            size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx || !edge_dev || !cam_dev || !out_dev || !workspace_dev)
        return fail(IRN_ERR_ARG, "irn_walk_run: null argument");
    if (ctx->n < 1) return fail(IRN_ERR_STATE, "irn_walk_run: irn_walk_configure has not been called");
    if (workspace_bytes < ctx->ws_bytes)
        return fail(IRN_ERR_STATE, "irn_walk_run: workspace of %zu bytes, %zu needed", workspace_bytes, ctx->ws_bytes);
    if (!(beta > 0.f)) return fail(IRN_ERR_ARG, "irn_walk_run: beta must be > 0 (got %g)", (double)beta);
    if (n_sweeps < 0) return fail(IRN_ERR_ARG, "irn_walk_run: n_sweeps must be >= 0");
    if (((uintptr_t)workspace_dev & 255) != 0)
        return fail(IRN_ERR_ARG, "irn_walk_run: workspace must be 256-byte aligned");

    const int n = ctx->n;
    const int slot = ctx->stage_next;
    ctx->stage_next ^= 1;
    IRN_HIP_TRY(hipEventSynchronize(ctx->stage_ev[slot]));   // previous upload from this slot has landed
    if (ctx->res_err_host && ctx->res_err_host[0] != 0) {    // an earlier resident launch gave up: do not go on silently
        const int rc_prev = irn_walk_check(ctx);
        if (rc_prev) return rc_prev;
    }
    WalkImg *imgs = (WalkImg *)ctx->stage[slot];
    AffJob *jobs = (AffJob *)(imgs + n);
    char *ws = (char *)workspace_dev;
    for (int i = 0; i < n; ++i) {
        if (!edge_dev[i] || !cam_dev[i] || !out_dev[i])
            return fail(IRN_ERR_ARG, "irn_walk_run: null pointer for image %d", i);
        WalkImg &I = imgs[i];
        I.edge = edge_dev[i];
        I.cam = cam_dev[i];
        I.inst = (inst_map_dev && inst_map_dev[i]) ? inst_map_dev[i] : nullptr;
        I.k_inst = (I.inst && k_inst) ? k_inst[i] : 1;
        if (I.inst && (I.k_inst < 1 || ctx->c[i] % I.k_inst != 0))
            return fail(IRN_ERR_ARG, "irn_walk_run: image %d: c=%d is not a multiple of k_inst=%d", i, ctx->c[i], I.k_inst);
        I.out = out_dev[i];
        I.wts = (float *)(ws + ctx->off_wts[i]) + ctx->front_pad[i];
        I.inv_deg = (double *)(ws + ctx->off_deg[i]);
        I.xa = (float *)(ws + ctx->off_xa[i]);
        I.xb = (float *)(ws + ctx->off_xb[i]);
        I.xc = (float *)(ws + ctx->off_xc[i]);
        I.h = ctx->h[i];
        I.w = ctx->w[i];
        I.C = ctx->c[i];
        I.plane_stride = ctx->plane_stride[i];
        I.front_pad = ctx->front_pad[i];
        I.n_dirs = ctx->tab->n_dirs;
        AffJob &J = jobs[i];
        J.edge = I.edge;
        J.out = I.wts;
        J.gh = I.h; J.gw = I.w; J.oy = 0; J.ox = 0; J.sh = I.h; J.sw = I.w;
        J.plane_stride = I.plane_stride;
    }
    IRN_HIP_TRY(hipMemcpyAsync(ctx->imgs_dev, imgs, sizeof(WalkImg) * n, hipMemcpyHostToDevice, stream));
    IRN_HIP_TRY(hipMemcpyAsync(ctx->jobs_dev, jobs, sizeof(AffJob) * n, hipMemcpyHostToDevice, stream));
    IRN_HIP_TRY(hipEventRecord(ctx->stage_ev[slot], stream));

    const DeviceTable &tab = *ctx->tab;
    hipLaunchKernelGGL(zero_pad_kernel, dim3(tab.n_dirs, n), dim3(256), 0, stream, ctx->imgs_dev, tab.n_dirs);
    IRN_LAUNCH_CHECK("zero_pad_kernel");
    int rc = launch_affinity(ctx->jobs_dev, n, ctx->max_h, ctx->max_w, tab, true, beta, stream);
    if (rc) return rc;
    const bool resident = ctx->variant == 2 && ctx->res_ok && n_sweeps > 0;
    // the resident kernel sums the degree from the weights it holds in registers; the inv_deg array is then
    // only filled on demand (irn_walk_export_weights, or the streaming re-run of irn_walk_sync)
    ctx->deg_stale = resident;
    ctx->last_stream = stream;
    ctx->last_n_sweeps = n_sweeps;
    ctx->last_resident = resident;
    ctx->last_valid = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing && resident) {
        rc = timing_events(ctx, &ev0, &ev1);
        if (rc) return rc;
        IRN_HIP_TRY(hipEventRecord(ev0, stream));
    }
    if (resident) {
        rc = resident_run(ctx, n_sweeps, stream);
        if (rc) return rc;
        if (ctx->timing) {
            IRN_HIP_TRY(hipEventRecord(ev1, stream));
            ctx->ev_used += 2;
            ctx->pending_launches += ctx->sched_steps;
        }
    } else {
        rc = streaming_run(ctx, n_sweeps, stream, ctx->timing != 0);
        if (rc) return rc;
    }
    if (!ctx->run_done_ev) IRN_HIP_TRY(hipEventCreateWithFlags(&ctx->run_done_ev, hipEventDisableTiming));
    IRN_HIP_TRY(hipEventRecord(ctx->run_done_ev, stream));
    return IRN_OK;
}

extern "C" int irn_walk_check(irn_walk_ctx *ctx) {
    if (!ctx) return fail(IRN_ERR_ARG, "null ctx");
    if (!ctx->res_err_host) return IRN_OK;
    if (ctx->res_err_host[0] != 0) {
        const unsigned img = ctx->res_err_host[1], t = ctx->res_err_host[2], wg = ctx->res_err_host[3];
        ctx->res_err_host[0] = 0;
        return fail(IRN_ERR_STATE, "resident walk timed out: workgroup %u waiting for state of sweep %u of image %u "
                                   "(a neighbouring tile never published it)", wg, t, img);
    }
    return IRN_OK;
}

// Wait for the last irn_walk_run.  If it ran on the persistent kernel and a tile gave up its bounded wait (the grid
// was not co-resident in time: another process or stream held compute units), the batch is run again on the
// streaming sweeps — same operator, kernel boundaries instead of in-launch hand-offs — so the caller's outputs are
// valid when this returns IRN_OK.
extern "C" int irn_walk_sync(irn_walk_ctx *ctx, int *fell_back) {
    if (!ctx) return fail(IRN_ERR_ARG, "null ctx");
    if (fell_back) *fell_back = 0;
    if (!ctx->last_valid || !ctx->run_done_ev) return IRN_OK;
    IRN_HIP_TRY(hipEventSynchronize(ctx->run_done_ev));
    if (!ctx->last_resident || !ctx->res_err_host || ctx->res_err_host[0] == 0) return IRN_OK;
    ctx->res_err_host[0] = 0;
    ctx->last_resident = false;
    ++ctx->fallback_runs;
    if (fell_back) *fell_back = 1;
    int rc = streaming_run(ctx, ctx->last_n_sweeps, ctx->last_stream, false);
    if (rc) return rc;
    IRN_HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    return IRN_OK;
}

extern "C" int irn_walk_fallback_runs(irn_walk_ctx *ctx) { return ctx ? ctx->fallback_runs : 0; }

extern "C" int irn_walk_tuning(irn_walk_ctx *ctx, int *poll_delay, int *placement, float *probe_ms4) {
    if (!ctx) return fail(IRN_ERR_ARG, "irn_walk_tuning: null ctx");
    if (poll_delay) *poll_delay = ctx->res_poll_delay;
    if (placement) *placement = ctx->res_placement;
    if (probe_ms4)
        for (int k = 0; k < 4; ++k) probe_ms4[k] = ctx->res_poll_probe_ms[k];
    return IRN_OK;
}

extern "C" int irn_walk_plan_rounds(int radius, int n_images, const int32_t *h, const int32_t *w, const int32_t *channels,
                                    int n_workgroups, int placement, int32_t *jobs_out, int cap_rounds, int *n_rounds) {
    if (!h || !w || !channels || !n_rounds || n_images < 1 || n_workgroups < 1 || cap_rounds < 0 || (cap_rounds > 0 && !jobs_out))
        return fail(IRN_ERR_ARG, "irn_walk_plan_rounds: bad argument");
    if (radius != 5 && radius != 10) return fail(IRN_ERR_ARG, "irn_walk_plan_rounds: the persistent walk exists for radius 5 and 10");
    std::vector<int> jobs;
    int nr = 0;
    if (!irn::resident_plan_rounds(radius, n_images, h, w, channels, n_workgroups, placement, jobs, &nr)) {
        *n_rounds = 0;                       // not an error: such a batch runs on the streaming sweeps
        return IRN_OK;
    }
    *n_rounds = nr;
    if (nr > cap_rounds) return cap_rounds == 0 ? IRN_OK : fail(IRN_ERR_ARG, "irn_walk_plan_rounds: jobs_out too small");
    std::copy(jobs.begin(), jobs.end(), jobs_out);
    return IRN_OK;
}

extern "C" int irn_walk_read_profile(irn_walk_ctx *ctx, long long *host_out) {
    if (!ctx || !host_out) return fail(IRN_ERR_ARG, "null argument");
    if (!ctx->res_prof_dev) return fail(IRN_ERR_STATE, "option 'profile' is off");
    IRN_HIP_TRY(hipMemcpy(host_out, ctx->res_prof_dev, 2 * 256 * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    return IRN_OK;
}

extern "C" int irn_walk_export_weights(irn_walk_ctx *ctx, int image, float *w_dev, double *inv_deg_dev,
                                       void *workspace_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!ctx || ctx->n < 1 || image < 0 || image >= ctx->n || !workspace_dev)
        return fail(IRN_ERR_ARG, "irn_walk_export_weights: bad argument");
    char *ws = (char *)workspace_dev;
    const size_t npx = (size_t)ctx->h[image] * ctx->w[image];
    if (w_dev) {
        const float *src = (const float *)(ws + ctx->off_wts[image]) + ctx->front_pad[image];
        IRN_HIP_TRY(hipMemcpy2DAsync(w_dev, sizeof(float) * npx, src, sizeof(float) * ctx->plane_stride[image],
                                     sizeof(float) * npx, ctx->tab->n_dirs, hipMemcpyDeviceToDevice, stream));
    }
    if (inv_deg_dev && ctx->deg_stale) {
        hipLaunchKernelGGL(degree_kernel, dim3(cdiv(ctx->max_n, 256), ctx->n), dim3(256), 0, stream, ctx->imgs_dev,
                           ctx->tab->dir_dy, ctx->tab->dir_dx, ctx->tab->n_dirs);
        IRN_LAUNCH_CHECK("degree_kernel");
        ctx->deg_stale = false;
    }
    if (inv_deg_dev)
        IRN_HIP_TRY(hipMemcpyAsync(inv_deg_dev, ws + ctx->off_deg[image], sizeof(double) * npx,
                                   hipMemcpyDeviceToDevice, stream));
    return IRN_OK;
}
