// Inference batch norm (+ residual) (+ ReLU) of the ResNet-50 trunk as ONE pass over the convolution's output, in place:
//     x[n, c, :, :] = act(x[n, c, :, :] * scale[c] + shift[c] (+ res[n, c, :, :]))
// with scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale folded by the caller.
//
// Replaces the elementwise tail of reference net/resnet50.py:35-55 (Bottleneck.forward: FixedBatchNorm :11-14 ->
// `out += residual` -> ReLU) and of the stem (:87-89): three kernels and seven tensor transfers at the end of a
// bottleneck (batch norm read + write, add two reads + write, ReLU read + write) become one kernel and three; after a
// plain convolution four transfers become two.  The convolutions stay on MIOpen / rocBLAS; measured on the CAM leg
// (profiles/r02_s13_cam_kernel_stats.csv) the elementwise kernels were 28 % of the backbone's time.
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

template <bool HAS_RES, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_act_kernel(float *__restrict__ x, const float *__restrict__ res,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          unsigned n_pieces, unsigned hw, unsigned n_ch, Div by_hw, Div by_ch) {
    const unsigned base = blockIdx.x * (unsigned)(kThreads * kPieces) + threadIdx.x;
    f4v v[kPieces], r[kPieces];
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const unsigned p = base + j * kThreads;
        if (p < n_pieces) {
            v[j] = reinterpret_cast<const f4v *>(x)[p];
            if (HAS_RES) r[j] = reinterpret_cast<const f4v *>(res)[p];
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
        f4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool upper = (unsigned)k >= left;
            float y = fmaf(v[j][k], upper ? s1 : s0, upper ? b1 : b0);
            if (HAS_RES) y += r[j][k];
            if (RELU) y = y < 0.f ? 0.f : y;      // NaN stays NaN, like torch.relu
            o[k] = y;
        }
        reinterpret_cast<f4v *>(x)[p] = o;
    }
}

// one element per thread from `first` on: the last numel % 4 elements, and tensors whose planes are shorter than a piece
template <bool HAS_RES, bool RELU>
__global__ void bn_act_tail_kernel(float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ scale,
                                   const float *__restrict__ shift, unsigned first, unsigned numel, unsigned hw, unsigned n_ch) {
    const unsigned e = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= numel) return;
    const unsigned c = (e / hw) % n_ch;
    float y = fmaf(x[e], scale[c], shift[c]);
    if (HAS_RES) y += res[e];
    if (RELU) y = y < 0.f ? 0.f : y;
    x[e] = y;
}

template <bool HAS_RES, bool RELU>
int launch(float *x, const float *res, const float *scale, const float *shift, unsigned numel, unsigned hw, unsigned n_ch,
           hipStream_t stream) {
    // planes shorter than a piece (a piece would span more than two of them) go element by element; such maps carry no time
    const unsigned n_pieces = hw >= 4u ? numel / 4u : 0u;
    if (n_pieces) {
        const unsigned blocks = (n_pieces + kThreads * kPieces - 1) / (kThreads * kPieces);
        hipLaunchKernelGGL((bn_act_kernel<HAS_RES, RELU>), dim3(blocks), dim3(kThreads), 0, stream, x, res, scale, shift, n_pieces,
                           hw, n_ch, make_div(hw), make_div(n_ch));
        IRN_LAUNCH_CHECK("bn_act_kernel");
    }
    const unsigned rest = numel - n_pieces * 4u;
    if (rest) {
        hipLaunchKernelGGL((bn_act_tail_kernel<HAS_RES, RELU>), dim3((rest + 63u) / 64u), dim3(64), 0, stream, x, res, scale, shift,
                           n_pieces * 4u, numel, hw, n_ch);
        IRN_LAUNCH_CHECK("bn_act_tail_kernel");
    }
    return IRN_OK;
}

}  // namespace
}  // namespace irn

extern "C" int irn_bn_act(float *x_dev, const float *res_dev, const float *scale_dev, const float *shift_dev, int64_t n_images,
                          int n_channels, int64_t plane_elems, int relu, void *stream) {
    using namespace irn;
    if (!x_dev || !scale_dev || !shift_dev) return fail(IRN_ERR_ARG, "irn_bn_act: null pointer");
    if (n_images < 0 || n_channels <= 0 || plane_elems < 0) return fail(IRN_ERR_ARG, "irn_bn_act: negative size");
    if (((uintptr_t)x_dev | (uintptr_t)res_dev) & 15u) return fail(IRN_ERR_ARG, "irn_bn_act: tensors must be 16-byte aligned");
    const int64_t numel = n_images * n_channels * plane_elems;
    if (numel == 0) return IRN_OK;
    if (numel >= (1ll << 31)) return fail(IRN_ERR_ARG, "irn_bn_act: %lld elements; at most 2^31 - 1 per call", (long long)numel);
    const unsigned hw = (unsigned)plane_elems, n_ch = (unsigned)n_channels;
    hipStream_t s = (hipStream_t)stream;
    if (res_dev && relu) return launch<true, true>(x_dev, res_dev, scale_dev, shift_dev, (unsigned)numel, hw, n_ch, s);
    if (res_dev) return launch<true, false>(x_dev, res_dev, scale_dev, shift_dev, (unsigned)numel, hw, n_ch, s);
    if (relu) return launch<false, true>(x_dev, res_dev, scale_dev, shift_dev, (unsigned)numel, hw, n_ch, s);
    return launch<false, false>(x_dev, res_dev, scale_dev, shift_dev, (unsigned)numel, hw, n_ch, s);
}
