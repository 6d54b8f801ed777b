// fp16 hi/lo split of a channels-last fp32 activation: the A operand of the split-precision 1x1 convolutions
// (irn_gemm16_nhwc, conv1x1.cpp; reference net/resnet50.py:34-54).
//
//     x = hi + 2^-11 lo',   hi = fp16(x),   lo' = fp16((x - hi) 2^11)            (11 + 11 mantissa bits)
//     out[p, 0:c] = hi,  out[p, c:2c] = hi,  out[p, 2c:3c] = lo'                 (fp16 [n_pixels, 3c])
//
// The GEMM multiplies this row with [w_hi | w_lo | w_hi 2^-11] (prepared once per layer on the host, in fp64), i.e.
// x_hi w_hi + x_hi w_lo + x_lo w_hi in one fp16 MFMA product over 3c with fp32 accumulation; the dropped term is
// x_lo w_lo ~ 2^-22 |x w|.  Scaling lo' by 2^11 keeps it a NORMAL fp16 number for every |x| above 1.2e-4 (unscaled it
// would be subnormal below |x| ~ 0.12).  With `scale` / `shift` the pass is the inference batch norm + ReLU of the 3x3
// convolution in front (net/resnet50.py:40-42) at the same time: that convolution's output is read once as fp32 and
// never written back.  HBM-bound: 4 bytes read, 6 written per element.
#include "common.hpp"
#include "kernels.hpp"

namespace irn {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr int kThreads = 256;
constexpr float kFp16Max = 65504.f;

// one thread = 8 consecutive channels of one pixel: two 16-byte loads, three 16-byte stores
// Row maps of the padded form (irn_split16_pad): pixel (n, y, x) of an [N, h, w] map lives at row n (h+2)(w+2) + (y+1)(w+2) + x+1
// of the zero-bordered layout a 3x3 / pad 1 convolution reads as nine row-shifted GEMM operands (conv1x1.cpp).
struct PadMap {
    unsigned h, w;          // 0, 0: dense on both sides
    int in_pad, out_pad;
};
__device__ __forceinline__ unsigned padded_row(unsigned row, unsigned h, unsigned w) {
    const unsigned n = row / (h * w), r = row - n * (h * w), y = r / w, x = r - y * w;
    return n * (h + 2u) * (w + 2u) + (y + 1u) * (w + 2u) + x + 1u;
}

template <bool BN, bool RELU>
__global__ __launch_bounds__(kThreads) void split16_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, _Float16 *__restrict__ out,
                                                           unsigned n_pieces, unsigned n_ch, unsigned pieces_per_row,
                                                           unsigned *__restrict__ overflow, PadMap pm) {
    const unsigned p = blockIdx.x * (unsigned)kThreads + threadIdx.x;
    if (p >= n_pieces) return;
    unsigned row = p / pieces_per_row;
    const unsigned c0 = (p - row * pieces_per_row) * 8u;
    const f4v *src;
    if (pm.out_pad) {
        // threads enumerate the rows of the BORDERED form: interior rows are split from their pixel, border rows are written as
        // zeros here (the operand needs no zeroing by the caller and may come fresh from an allocator every call)
        const unsigned hp = pm.h + 2u, wp = pm.w + 2u;
        const unsigned n = row / (hp * wp), r = row - n * (hp * wp), yp = r / wp, xp = r - yp * wp;
        if (yp == 0u || yp > pm.h || xp == 0u || xp > pm.w) {
            _Float16 *dst = out + (size_t)row * (3u * n_ch) + c0;
            const h8v z = {0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<h8v *>(dst) = z;
            *reinterpret_cast<h8v *>(dst + n_ch) = z;
            *reinterpret_cast<h8v *>(dst + 2u * n_ch) = z;
            return;
        }
        const unsigned dense = n * (pm.h * pm.w) + (yp - 1u) * pm.w + (xp - 1u);
        src = reinterpret_cast<const f4v *>(x + (size_t)(pm.in_pad ? row : dense) * n_ch + c0);
    } else {
        const unsigned prow = pm.h ? padded_row(row, pm.h, pm.w) : row;
        src = reinterpret_cast<const f4v *>(x + (size_t)(pm.in_pad ? prow : row) * n_ch + c0);
    }
    f4v a = __builtin_nontemporal_load(src), b = __builtin_nontemporal_load(src + 1);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (BN) {
        const f4v s0 = *reinterpret_cast<const f4v *>(scale + c0), s1 = *reinterpret_cast<const f4v *>(scale + c0 + 4);
        const f4v t0 = *reinterpret_cast<const f4v *>(shift + c0), t1 = *reinterpret_cast<const f4v *>(shift + c0 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float y = fmaf(v[k], sc[k], sh[k]);          // the same fmaf as irn_bn_act_nhwc
            if (RELU) y = y < 0.f ? 0.f : y;
            v[k] = y;
        }
    }
    h8v hi, lo;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const _Float16 h = (_Float16)v[k];
        hi[k] = h;
        lo[k] = (_Float16)((v[k] - (float)h) * 2048.f);
        bad |= !(fabsf(v[k]) <= kFp16Max);               // beyond fp16's range, or NaN
    }
    _Float16 *dst = out + (size_t)row * (3u * n_ch) + c0;
    *reinterpret_cast<h8v *>(dst) = hi;
    *reinterpret_cast<h8v *>(dst + n_ch) = hi;
    *reinterpret_cast<h8v *>(dst + 2u * n_ch) = lo;
    if (bad && overflow) atomicOr(overflow, 1u);
}

}  // namespace
}  // namespace irn

namespace irn {
namespace {
int split16_launch(const float *x_dev, const float *scale_dev, const float *shift_dev, int relu, void *out_dev, int64_t n_pixels,
                   int n_channels, unsigned *overflow_dev, void *stream, PadMap pm) {
    if (!x_dev || !out_dev) return fail(IRN_ERR_ARG, "irn_split16: null pointer");
    if ((scale_dev != nullptr) != (shift_dev != nullptr)) return fail(IRN_ERR_ARG, "irn_split16: scale and shift come together");
    if (relu && !scale_dev) return fail(IRN_ERR_ARG, "irn_split16: relu only with the batch norm in front of it");
    if (n_pixels < 0 || n_channels <= 0 || (n_channels & 7)) return fail(IRN_ERR_ARG, "irn_split16: n_channels must be a positive multiple of 8");
    if (((uintptr_t)x_dev | (uintptr_t)out_dev | (uintptr_t)scale_dev | (uintptr_t)shift_dev) & 15u)
        return fail(IRN_ERR_ARG, "irn_split16: tensors and constants must be 16-byte aligned");
    int64_t rows = n_pixels;                       // rows the threads enumerate: the bordered form's when it is the output
    if (pm.out_pad) rows = n_pixels / ((int64_t)pm.h * pm.w) * (pm.h + 2) * (pm.w + 2);
    const int64_t numel = rows * n_channels;
    if (numel == 0) return IRN_OK;
    if (numel >= (1ll << 31)) return fail(IRN_ERR_ARG, "irn_split16: %lld elements; at most 2^31 - 1 per call", (long long)numel);
    const unsigned n_pieces = (unsigned)(numel / 8), ppr = (unsigned)n_channels / 8u;
    const dim3 grid((n_pieces + kThreads - 1) / kThreads), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    _Float16 *out = (_Float16 *)out_dev;
    if (!scale_dev) hipLaunchKernelGGL((split16_kernel<false, false>), grid, block, 0, s, x_dev, scale_dev, shift_dev, out, n_pieces, (unsigned)n_channels, ppr, overflow_dev, pm);
    else if (relu) hipLaunchKernelGGL((split16_kernel<true, true>), grid, block, 0, s, x_dev, scale_dev, shift_dev, out, n_pieces, (unsigned)n_channels, ppr, overflow_dev, pm);
    else hipLaunchKernelGGL((split16_kernel<true, false>), grid, block, 0, s, x_dev, scale_dev, shift_dev, out, n_pieces, (unsigned)n_channels, ppr, overflow_dev, pm);
    IRN_LAUNCH_CHECK("split16_kernel");
    return IRN_OK;
}
}  // namespace
}  // namespace irn

extern "C" int irn_split16(const float *x_dev, const float *scale_dev, const float *shift_dev, int relu, void *out_dev,
                           int64_t n_pixels, int n_channels, unsigned *overflow_dev, void *stream) {
    return irn::split16_launch(x_dev, scale_dev, shift_dev, relu, out_dev, n_pixels, n_channels, overflow_dev, stream, irn::PadMap{0u, 0u, 0, 0});
}

extern "C" int irn_split16_pad(const float *x_dev, const float *scale_dev, const float *shift_dev, int relu, void *out_dev,
                               int64_t n_images, int h, int w, int n_channels, int in_padded, int out_padded, unsigned *overflow_dev,
                               void *stream) {
    using namespace irn;
    if (n_images < 0 || h < 1 || w < 1) return fail(IRN_ERR_ARG, "irn_split16_pad: n_images >= 0, h, w >= 1");
    if ((int64_t)n_images * (h + 2) * (w + 2) * n_channels >= (1ll << 31)) return fail(IRN_ERR_ARG, "irn_split16_pad: at most 2^31 - 1 padded elements per call");
    return split16_launch(x_dev, scale_dev, shift_dev, relu, out_dev, n_images * h * w, n_channels, overflow_dev, stream,
                          PadMap{(unsigned)h, (unsigned)w, in_padded ? 1 : 0, out_padded ? 1 : 0});
}
