// Inference batch norm (+ residual) (+ ReLU) of the ResNet-50 trunk as ONE pass over the convolution's output, in place:
//     x[n, c, :, :] = act(x[n, c, :, :] * scale[c] + shift[c] (+ res[n, c, :, :] (* res_scale[c] + res_shift[c])))
// with scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale folded by the caller; the residual may
// carry a batch norm of its own (the projection shortcut of a stage's first unit, net/resnet50.py:48-49, whose output then
// never exists as a tensor of its own).
//
// Replaces the elementwise tail of reference net/resnet50.py:34-54 (Bottleneck.forward: FixedBatchNorm :11-14 ->
// `out += residual` -> ReLU) and of the stem (:94-97): three kernels and seven tensor transfers at the end of a
// bottleneck (batch norm read + write, add two reads + write, ReLU read + write) become one kernel and three; after a
// plain convolution four transfers become two.  The convolutions stay on MIOpen / rocBLAS; measured on the CAM leg
// (profiles/r02_s13_cam_kernel_stats_composed.csv) the elementwise kernels were 28 % of the backbone's time.
//
// HBM-bound (12 or 8 bytes per element).  The tensor is walked flat in 16-byte pieces, so every load and store is a
// full-width coalesced access whatever H x W is (VOC planes are rarely a multiple of 4 wide); a piece that straddles
// a plane boundary takes the next channel's constants for its upper elements.  Plane and channel of a piece come from
// multiply-shift divisions by invariant divisors prepared on the host.
#include "common.hpp"

#pragma clang fp contract(off)

namespace irn {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

// n / d for n < 2^31 as mulhi + shift (Granlund-Montgomery round-up multiplier)
struct Div {
    unsigned mul = 0, shr = 0, d = 1;
};

Div make_div(unsigned d) {
    Div r;
    r.d = d;
    if (d == 1) return r;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                 // ceil(log2 d)
    const unsigned p = 31 + l;
    r.mul = (unsigned)(((1ull << p) + d - 1) / d);
    r.shr = p - 32;
    return r;
}

__device__ __forceinline__ unsigned div_by(unsigned n, const Div dv) {
    return dv.d == 1 ? n : (__umulhi(n, dv.mul) >> dv.shr);
}

constexpr int kThreads = 256, kPieces = 4;       // 16-byte pieces per thread, a block-width apart

// (non-temporal loads of x / res were measured: faster for the one shape that exceeds the caches, 268 MB per tensor — 5.9 vs 5.3
// TB/s with a residual — and 10-20 % slower for every smaller one, whose lines the convolution has just left in L2 / MALL;
// profiles/r02_s19_epilogue_bench.txt)
__device__ __forceinline__ f4v load_piece(const float *p, unsigned i) { return reinterpret_cast<const f4v *>(p)[i]; }

// RES: 0 no residual, 1 residual added as it is, 2 residual through its own scale / shift first
template <int RES, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_act_kernel(float *__restrict__ x, const float *__restrict__ res,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          const float *__restrict__ res_scale, const float *__restrict__ res_shift,
                                                          unsigned n_pieces, unsigned hw, unsigned n_ch, Div by_hw, Div by_ch) {
    constexpr bool HAS_RES = RES != 0;
    const unsigned base = blockIdx.x * (unsigned)(kThreads * kPieces) + threadIdx.x;
    f4v v[kPieces], r[kPieces];
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const unsigned p = base + j * kThreads;
        if (p < n_pieces) {
            v[j] = load_piece(x, p);
            if (HAS_RES) r[j] = load_piece(res, p);
        }
    }
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const unsigned p = base + j * kThreads;
        if (p >= n_pieces) continue;
        const unsigned e = p * 4u;
        const unsigned plane = div_by(e, by_hw);
        const unsigned in_plane = e - plane * hw;
        const unsigned c0 = plane - div_by(plane, by_ch) * n_ch;
        const unsigned c1 = c0 + 1 == n_ch ? 0u : c0 + 1;
        const float s0 = scale[c0], b0 = shift[c0];
        float s1 = s0, b1 = b0;
        const unsigned left = hw - in_plane;      // elements of this piece that still belong to plane `plane` (if < 4)
        if (left < 4u) {
            s1 = scale[c1];
            b1 = shift[c1];
        }
        float rs0 = 1.f, rb0 = 0.f, rs1 = 1.f, rb1 = 0.f;
        if (RES == 2) {
            rs0 = rs1 = res_scale[c0];
            rb0 = rb1 = res_shift[c0];
            if (left < 4u) {
                rs1 = res_scale[c1];
                rb1 = res_shift[c1];
            }
        }
        f4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool upper = (unsigned)k >= left;
            float y = fmaf(v[j][k], upper ? s1 : s0, upper ? b1 : b0);
            if (RES == 1) y += r[j][k];
            if (RES == 2) y += fmaf(r[j][k], upper ? rs1 : rs0, upper ? rb1 : rb0);
            if (RELU) y = y < 0.f ? 0.f : y;      // NaN stays NaN, like torch.relu
            o[k] = y;
        }
        reinterpret_cast<f4v *>(x)[p] = o;
    }
}

// one element per thread from `first` on: the last numel % 4 elements, and tensors whose planes are shorter than a piece
template <int RES, bool RELU>
__global__ void bn_act_tail_kernel(float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ scale,
                                   const float *__restrict__ shift, const float *__restrict__ res_scale,
                                   const float *__restrict__ res_shift, unsigned first, unsigned numel, unsigned hw, unsigned n_ch) {
    const unsigned e = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= numel) return;
    const unsigned c = (e / hw) % n_ch;
    float y = fmaf(x[e], scale[c], shift[c]);
    if (RES == 1) y += res[e];
    if (RES == 2) y += fmaf(res[e], res_scale[c], res_shift[c]);
    if (RELU) y = y < 0.f ? 0.f : y;
    x[e] = y;
}

// The same pass over a channels-last tensor ([n_pixels, n_ch] in memory, n_ch a multiple of 4): the four elements of a
// 16-byte piece are four consecutive channels of one pixel, so the constants come as 16-byte loads too and no piece
// straddles anything.  (For the trunk run with `IRN_CHANNELS_LAST=1`: MIOpen's NHWC solvers.)
template <int RES, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_act_nhwc_kernel(float *__restrict__ x, const float *__restrict__ res,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               const float *__restrict__ res_scale, const float *__restrict__ res_shift,
                                                               unsigned n_pieces, unsigned n_ch, Div by_ch) {
    constexpr bool HAS_RES = RES != 0;
    const unsigned base = blockIdx.x * (unsigned)(kThreads * kPieces) + threadIdx.x;
    f4v v[kPieces], r[kPieces];
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const unsigned p = base + j * kThreads;
        if (p < n_pieces) {
            v[j] = load_piece(x, p);
            if (HAS_RES) r[j] = load_piece(res, p);
        }
    }
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const unsigned p = base + j * kThreads;
        if (p >= n_pieces) continue;
        const unsigned e = p * 4u;
        const unsigned c0 = e - div_by(e, by_ch) * n_ch;          // multiple of 4
        const f4v s = load_piece(scale, c0 >> 2), b = load_piece(shift, c0 >> 2);
        f4v rs, rb;
        if (RES == 2) {
            rs = load_piece(res_scale, c0 >> 2);
            rb = load_piece(res_shift, c0 >> 2);
        }
        f4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y = fmaf(v[j][k], s[k], b[k]);
            if (RES == 1) y += r[j][k];
            if (RES == 2) y += fmaf(r[j][k], rs[k], rb[k]);
            if (RELU) y = y < 0.f ? 0.f : y;
            o[k] = y;
        }
        reinterpret_cast<f4v *>(x)[p] = o;
    }
}

template <int RES, bool RELU>
int launch_nhwc(float *x, const float *res, const float *scale, const float *shift, const float *res_scale, const float *res_shift,
                unsigned numel, unsigned n_ch, hipStream_t stream) {
    const unsigned n_pieces = numel / 4u;
    const unsigned blocks = (n_pieces + kThreads * kPieces - 1) / (kThreads * kPieces);
    hipLaunchKernelGGL((bn_act_nhwc_kernel<RES, RELU>), dim3(blocks), dim3(kThreads), 0, stream, x, res, scale, shift, res_scale,
                       res_shift, n_pieces, n_ch, make_div(n_ch));
    IRN_LAUNCH_CHECK("bn_act_nhwc_kernel");
    return IRN_OK;
}

template <int RES, bool RELU>
int launch(float *x, const float *res, const float *scale, const float *shift, const float *res_scale, const float *res_shift,
           unsigned numel, unsigned hw, unsigned n_ch, hipStream_t stream) {
    // planes shorter than a piece (a piece would span more than two of them) go element by element; such maps carry no time
    const unsigned n_pieces = hw >= 4u ? numel / 4u : 0u;
    if (n_pieces) {
        const unsigned blocks = (n_pieces + kThreads * kPieces - 1) / (kThreads * kPieces);
        hipLaunchKernelGGL((bn_act_kernel<RES, RELU>), dim3(blocks), dim3(kThreads), 0, stream, x, res, scale, shift, res_scale,
                           res_shift, n_pieces, hw, n_ch, make_div(hw), make_div(n_ch));
        IRN_LAUNCH_CHECK("bn_act_kernel");
    }
    const unsigned rest = numel - n_pieces * 4u;
    if (rest) {
        hipLaunchKernelGGL((bn_act_tail_kernel<RES, RELU>), dim3((rest + 63u) / 64u), dim3(64), 0, stream, x, res, scale, shift,
                           res_scale, res_shift, n_pieces * 4u, numel, hw, n_ch);
        IRN_LAUNCH_CHECK("bn_act_tail_kernel");
    }
    return IRN_OK;
}

}  // namespace
}  // namespace irn

extern "C" int irn_bn_act(float *x_dev, const float *res_dev, const float *scale_dev, const float *shift_dev,
                          const float *res_scale_dev, const float *res_shift_dev, int64_t n_images, int n_channels,
                          int64_t plane_elems, int relu, void *stream) {
    using namespace irn;
    if (!x_dev || !scale_dev || !shift_dev) return fail(IRN_ERR_ARG, "irn_bn_act: null pointer");
    if ((res_scale_dev != nullptr) != (res_shift_dev != nullptr) || (res_scale_dev && !res_dev))
        return fail(IRN_ERR_ARG, "irn_bn_act: res_scale and res_shift come together, and only with a residual");
    if (n_images < 0 || n_channels <= 0 || plane_elems < 0) return fail(IRN_ERR_ARG, "irn_bn_act: negative size");
    if (((uintptr_t)x_dev | (uintptr_t)res_dev) & 15u) return fail(IRN_ERR_ARG, "irn_bn_act: tensors must be 16-byte aligned");
    const int64_t numel = n_images * n_channels * plane_elems;
    if (numel == 0) return IRN_OK;
    if (numel >= (1ll << 31)) return fail(IRN_ERR_ARG, "irn_bn_act: %lld elements; at most 2^31 - 1 per call", (long long)numel);
    const unsigned hw = (unsigned)plane_elems, n_ch = (unsigned)n_channels, n = (unsigned)numel;
    hipStream_t s = (hipStream_t)stream;
    const int mode = !res_dev ? 0 : (res_scale_dev ? 2 : 1);
#define IRN_BN_ACT_CASE(RES_MODE)                                                                                                   \
    return relu ? launch<RES_MODE, true>(x_dev, res_dev, scale_dev, shift_dev, res_scale_dev, res_shift_dev, n, hw, n_ch, s)       \
                : launch<RES_MODE, false>(x_dev, res_dev, scale_dev, shift_dev, res_scale_dev, res_shift_dev, n, hw, n_ch, s)
    if (mode == 0) IRN_BN_ACT_CASE(0);
    if (mode == 1) IRN_BN_ACT_CASE(1);
    IRN_BN_ACT_CASE(2);
#undef IRN_BN_ACT_CASE
}

extern "C" int irn_bn_act_nhwc(float *x_dev, const float *res_dev, const float *scale_dev, const float *shift_dev,
                               const float *res_scale_dev, const float *res_shift_dev, int64_t n_pixels, int n_channels, int relu,
                               void *stream) {
    using namespace irn;
    if (!x_dev || !scale_dev || !shift_dev) return fail(IRN_ERR_ARG, "irn_bn_act_nhwc: null pointer");
    if ((res_scale_dev != nullptr) != (res_shift_dev != nullptr) || (res_scale_dev && !res_dev))
        return fail(IRN_ERR_ARG, "irn_bn_act_nhwc: res_scale and res_shift come together, and only with a residual");
    if (n_pixels < 0 || n_channels <= 0 || (n_channels & 3)) return fail(IRN_ERR_ARG, "irn_bn_act_nhwc: n_channels must be a positive multiple of 4");
    if (((uintptr_t)x_dev | (uintptr_t)res_dev | (uintptr_t)scale_dev | (uintptr_t)shift_dev | (uintptr_t)res_scale_dev |
         (uintptr_t)res_shift_dev) & 15u)
        return fail(IRN_ERR_ARG, "irn_bn_act_nhwc: tensors and constants must be 16-byte aligned");
    const int64_t numel = n_pixels * n_channels;
    if (numel == 0) return IRN_OK;
    if (numel >= (1ll << 31)) return fail(IRN_ERR_ARG, "irn_bn_act_nhwc: %lld elements; at most 2^31 - 1 per call", (long long)numel);
    const unsigned n_ch = (unsigned)n_channels, n = (unsigned)numel;
    hipStream_t s = (hipStream_t)stream;
    const int mode = !res_dev ? 0 : (res_scale_dev ? 2 : 1);
#define IRN_BN_ACT_CASE(RES_MODE)                                                                                                   \
    return relu ? launch_nhwc<RES_MODE, true>(x_dev, res_dev, scale_dev, shift_dev, res_scale_dev, res_shift_dev, n, n_ch, s)      \
                : launch_nhwc<RES_MODE, false>(x_dev, res_dev, scale_dev, shift_dev, res_scale_dev, res_shift_dev, n, n_ch, s)
    if (mode == 0) IRN_BN_ACT_CASE(0);
    if (mode == 1) IRN_BN_ACT_CASE(1);
    IRN_BN_ACT_CASE(2);
#undef IRN_BN_ACT_CASE
}

// ------------------------------------------------------------------------------------------------
// Stem: batch norm + ReLU + 3x3 / stride 2 / pad 1 max pool of reference net/resnet50.py:94-97 (the nets' stage1,
// net/resnet50_cam.py:14, net/resnet50_irn.py:15) in one pass: the largest activation of the trunk (64 x H/2 x W/2) is
// read once and a quarter of it written, instead of read + written by the epilogue and read again by the pool.
// ------------------------------------------------------------------------------------------------
namespace irn {
namespace {

// A workgroup produces a 4 x 64 tile of outputs from a 9 x 129 tile of inputs: the inputs are read once, row-contiguous
// (consecutive lanes = consecutive floats), pass through batch norm + ReLU on their way into LDS (out-of-image taps
// become 0, the ReLU's floor), and every thread takes its 3 x 3 window from there as 8-byte reads (one output = columns
// 2x-1 .. 2x+1 of the tile; lanes a pair of columns apart never share a bank).
constexpr int kPoolW = 64, kPoolH = 4, kPoolInW = 2 * kPoolW + 1, kPoolInH = 2 * kPoolH + 1, kPoolLdsW = 132;

__global__ __launch_bounds__(256) void stem_pool_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, float *__restrict__ out, int n_ch, int h,
                                                        int w, int ho, int wo) {
    __shared__ __attribute__((aligned(16))) float tile[kPoolInH][kPoolLdsW];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int xo0 = blockIdx.x * kPoolW, yo0 = blockIdx.y * kPoolH;
    const unsigned plane = blockIdx.z;
    const float s = scale[plane % (unsigned)n_ch], b = shift[plane % (unsigned)n_ch];
    const float *src = x + (size_t)plane * h * w;
    const int x_in0 = 2 * xo0 - 1, y_in0 = 2 * yo0 - 1;
    // tile[r][c + 1] holds input (y_in0 + r, x_in0 + c); column 0 is padding so that a thread's window starts 8-byte aligned
#pragma unroll
    for (int i = threadIdx.x; i < kPoolInH * kPoolInW; i += 256) {
        const int r = i / kPoolInW, c = i - r * kPoolInW;
        const int yy = y_in0 + r, xx = x_in0 + c;
        float v = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            v = fmaf(src[(size_t)yy * w + xx], s, b);
            v = v < 0.f ? 0.f : v;                                    // NaN stays NaN
        }
        tile[r][c + 1] = v;
    }
    __syncthreads();
    const int xo = xo0 + tx, yo = yo0 + ty;
    if (xo >= wo || yo >= ho) return;
    float m = 0.f;
    bool nan = false;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        // columns 2tx .. 2tx+3 of the LDS row = inputs x_in0 + 2tx - 1 .. + 2: the window is the last three
        const float2 lo = *reinterpret_cast<const float2 *>(&tile[2 * ty + ky][2 * tx]);
        const float2 hi = *reinterpret_cast<const float2 *>(&tile[2 * ty + ky][2 * tx + 2]);
        const float v0 = lo.y, v1 = hi.x, v2 = hi.y;
        nan |= (v0 != v0) | (v1 != v1) | (v2 != v2);
        m = v0 > m ? v0 : m;
        m = v1 > m ? v1 : m;
        m = v2 > m ? v2 : m;
    }
    out[((size_t)plane * ho + yo) * wo + xo] = nan ? __builtin_nanf("") : m;      // torch's max pool propagates NaN
}

// Bilinear x`factor` upsampling (align_corners = False, the scale factor handed to the kernel as nn.Upsample does) + ReLU of
// the IRNet heads, reference net/resnet50_irn.py:36,42,48,72,78,84 followed by nn.ReLU.  Source index and weights as in
// ATen's upsample_bilinear2d: src = (dst + 0.5) / factor - 0.5 clamped at 0, lambda = src - floor(src), and
//     out = (1 - ly) * ((1 - lx) * a + lx * b) + ly * ((1 - lx) * c + lx * d)      in fp32, no contraction.
// Bound by the output write (4 or 16 times the input); V consecutive outputs of a row per thread leave as one store.
template <int V, bool RELU>
__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ x, float *__restrict__ out, int h, int w, int ho,
                                                       int wo, float rscale) {
    const int xg = blockIdx.x * 64 + (threadIdx.x & 63);         // group of V outputs
    const int yo = blockIdx.y * 4 + (threadIdx.x >> 6);
    const unsigned plane = blockIdx.z;
    if (xg * V >= wo || yo >= ho) return;
    float sy = rscale * ((float)yo + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = sy - (float)y0, ly0 = 1.f - ly;
    const float *r0 = x + ((size_t)plane * h + y0) * w, *r1 = x + ((size_t)plane * h + y1) * w;
    float o[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int xo = xg * V + v;
        float sx = rscale * ((float)xo + 0.5f) - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx = sx - (float)x0, lx0 = 1.f - lx;
        float val = ly0 * (lx0 * r0[x0] + lx * r0[x1]) + ly * (lx0 * r1[x0] + lx * r1[x1]);
        if (RELU) val = val < 0.f ? 0.f : val;
        o[v] = val;
    }
    float *dst = out + ((size_t)plane * ho + yo) * wo + (size_t)xg * V;
    if constexpr (V == 4) *reinterpret_cast<f4v *>(dst) = f4v{o[0], o[1], o[2], o[3]};
    else if constexpr (V == 2) *reinterpret_cast<float2 *>(dst) = float2{o[0], o[1]};
    else dst[0] = o[0];
}

template <int V>
int launch_upsample(const float *x, float *out, unsigned n_planes, int h, int w, int ho, int wo, float rscale, int relu,
                    hipStream_t s) {
    const dim3 grid((unsigned)cdiv(cdiv(wo, V), 64), (unsigned)cdiv(ho, 4), n_planes);
    if (relu) hipLaunchKernelGGL((upsample_kernel<V, true>), grid, dim3(256), 0, s, x, out, h, w, ho, wo, rscale);
    else hipLaunchKernelGGL((upsample_kernel<V, false>), grid, dim3(256), 0, s, x, out, h, w, ho, wo, rscale);
    IRN_LAUNCH_CHECK("upsample_kernel");
    return IRN_OK;
}

}  // namespace
}  // namespace irn

extern "C" int irn_stem_pool(const float *x_dev, const float *scale_dev, const float *shift_dev, int64_t n_images, int n_channels,
                             int h, int w, float *out_dev, void *stream) {
    using namespace irn;
    if (!x_dev || !scale_dev || !shift_dev || !out_dev) return fail(IRN_ERR_ARG, "irn_stem_pool: null pointer");
    if (n_images < 0 || n_channels <= 0 || h <= 0 || w <= 0) return fail(IRN_ERR_ARG, "irn_stem_pool: non-positive size");
    const int64_t n_planes = n_images * n_channels;
    if (n_planes == 0) return IRN_OK;
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    if (cdiv(ho, 4) > 65535) return fail(IRN_ERR_ARG, "irn_stem_pool: %d rows; at most 524280", h);
    if (n_channels > 65535) return fail(IRN_ERR_ARG, "irn_stem_pool: %d channels; at most 65535", n_channels);
    // the plane index rides on grid.z (at most 65535 per launch); launches start at whole images so that plane % n_channels holds
    const int64_t step = (int64_t)(65535 / n_channels) * n_channels;
    for (int64_t p0 = 0; p0 < n_planes; p0 += step) {
        const unsigned np = (unsigned)((n_planes - p0) < step ? (n_planes - p0) : step);
        hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)cdiv(wo, 64), (unsigned)cdiv(ho, 4), np), dim3(256), 0, (hipStream_t)stream,
                           x_dev + (size_t)p0 * h * w, scale_dev, shift_dev, out_dev + (size_t)p0 * ho * wo, n_channels, h, w, ho, wo);
        IRN_LAUNCH_CHECK("stem_pool_kernel");
    }
    return IRN_OK;
}

extern "C" int irn_upsample_bilinear(const float *x_dev, int64_t n_planes, int h, int w, int factor, int relu, float *out_dev,
                                     void *stream) {
    using namespace irn;
    if (!x_dev || !out_dev) return fail(IRN_ERR_ARG, "irn_upsample_bilinear: null pointer");
    if (n_planes < 0 || h <= 0 || w <= 0 || factor < 1 || factor > 64)
        return fail(IRN_ERR_ARG, "irn_upsample_bilinear: non-positive size or factor outside 1..64");
    if ((uintptr_t)out_dev & 15u) return fail(IRN_ERR_ARG, "irn_upsample_bilinear: output must be 16-byte aligned");
    if (n_planes == 0) return IRN_OK;
    const int ho = h * factor, wo = w * factor;
    if (cdiv(ho, 4) > 65535) return fail(IRN_ERR_ARG, "irn_upsample_bilinear: %d output rows; at most 262140", ho);
    const float rscale = (float)(1.0 / (double)factor);
    for (int64_t p0 = 0; p0 < n_planes; p0 += 65535) {
        const unsigned np = (unsigned)((n_planes - p0) < 65535 ? (n_planes - p0) : 65535);
        const float *xs = x_dev + (size_t)p0 * h * w;
        float *os = out_dev + (size_t)p0 * ho * wo;
        int rc;
        // rows start at multiples of wo elements: a 16-byte store needs wo % 4 == 0 (plane and row starts then stay aligned)
        if (wo % 4 == 0) rc = launch_upsample<4>(xs, os, np, h, w, ho, wo, rscale, relu, (hipStream_t)stream);
        else if (wo % 2 == 0) rc = launch_upsample<2>(xs, os, np, h, w, ho, wo, rscale, relu, (hipStream_t)stream);
        else rc = launch_upsample<1>(xs, os, np, h, w, ho, wo, rscale, relu, (hipStream_t)stream);
        if (rc != IRN_OK) return rc;
    }
    return IRN_OK;
}
