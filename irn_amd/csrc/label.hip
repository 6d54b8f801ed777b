// Label epilogue (gfx950): bilinear x4 upsample + crop + /global-max + background plane + argmax +
// keys look-up in two passes over the OUTPUT pixels; the fp32 [C,H,W] upsampled tensor is never
// materialised unless the caller asks for it (instance scoring).
//
// Replaces reference step/make_sem_seg_labels.py:43-49 and step/make_ins_seg_labels.py:137-145
// (F.interpolate -> slice -> torch.max -> divide -> F.pad -> argmax -> .cpu() -> numpy LUT).
//
// Bit-exactness: labels must equal the reference's, so the interpolation replays ATen's CPU
// arithmetic (torch 2.10; pinned by tests/golden/semseg.npz through the oracle):
//     src = max((dst + 0.5) * 0.25 - 0.5, 0);  i0 = floor(src);  i1 = min(i0+1, n-1)
//     l1 = src - i0;  l0 = 1 - l1
//     top = fma(v00, lx0, v01*lx1);  bot = fma(v10, lx0, v11*lx1);  out = fma(ly0, top, ly1*bot)
// followed by an IEEE division by the global maximum; argmax keeps the FIRST maximum
// (strict > while scanning channels after the background plane).
//
// Roofline: reads C*h*w floats (L2-resident), writes H*W bytes; ~0.3 MB per 512^2 image against
// 2.7 GB streamed by the walk — not a bottleneck.
#include "kernels.hpp"

namespace irn {

struct LabelJob {
    const float *rw;        // [c,h,w]
    const int64_t *keys;    // [c] or null
    uint8_t *labels;        // [oh,ow] or null
    int32_t *argmax;        // [oh,ow] or null
    float *rw_up;           // [c,oh,ow] or null
    unsigned *max_slot;     // order-preserving encoding of the running max
    int c, h, w, oh, ow;
};

namespace {

__device__ __forceinline__ unsigned enc_ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct Taps {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Taps taps_x4(int dst, int n_in) {
    // area_pixel_compute_source_index(scale = 1/4, align_corners = false), clamped at 0
    float src = __fsub_rn(__fmul_rn(__fadd_rn((float)dst, 0.5f), 0.25f), 0.5f);
    src = src < 0.f ? 0.f : src;
    Taps t;
    t.i0 = min((int)src, n_in - 1);
    t.i1 = min(t.i0 + 1, n_in - 1);
    t.l1 = __fsub_rn(src, (float)t.i0);
    t.l0 = __fsub_rn(1.0f, t.l1);
    return t;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ plane, int w, const Taps &ty, const Taps &tx) {
    const float v00 = plane[ty.i0 * w + tx.i0], v01 = plane[ty.i0 * w + tx.i1];
    const float v10 = plane[ty.i1 * w + tx.i0], v11 = plane[ty.i1 * w + tx.i1];
    const float top = __fmaf_rn(v00, tx.l0, __fmul_rn(v01, tx.l1));
    const float bot = __fmaf_rn(v10, tx.l0, __fmul_rn(v11, tx.l1));
    return __fmaf_rn(ty.l0, top, __fmul_rn(ty.l1, bot));
}

__global__ __launch_bounds__(256) void upsample_max_kernel(const LabelJob *__restrict__ jobs) {
    const LabelJob J = jobs[blockIdx.y];
    const long npx = (long)J.oh * J.ow;
    float m = -INFINITY;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < npx; o += (long)gridDim.x * 256) {
        const int oy = (int)(o / J.ow), ox = (int)(o - (long)oy * J.ow);
        const Taps ty = taps_x4(oy, J.h), tx = taps_x4(ox, J.w);
        for (int c = 0; c < J.c; ++c) m = fmaxf(m, bilerp(J.rw + (long)c * J.h * J.w, J.w, ty, tx));
    }
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > -INFINITY) atomicMax(J.max_slot, enc_ordered(m));
    }
}

__global__ __launch_bounds__(256) void label_argmax_kernel(const LabelJob *__restrict__ jobs, float bg) {
    const LabelJob J = jobs[blockIdx.y];
    const long npx = (long)J.oh * J.ow;
    const float gmax = dec_ordered(*J.max_slot);
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < npx; o += (long)gridDim.x * 256) {
        const int oy = (int)(o / J.ow), ox = (int)(o - (long)oy * J.ow);
        const Taps ty = taps_x4(oy, J.h), tx = taps_x4(ox, J.w);
        float best = bg;
        int idx = 0;
        for (int c = 0; c < J.c; ++c) {
            const float v = __fdiv_rn(bilerp(J.rw + (long)c * J.h * J.w, J.w, ty, tx), gmax);
            if (J.rw_up) J.rw_up[(long)c * npx + o] = v;
            if (v > best) {
                best = v;
                idx = c + 1;
            }
        }
        if (J.argmax) J.argmax[o] = idx;
        if (J.labels) J.labels[o] = idx == 0 ? (uint8_t)0 : (uint8_t)(J.keys[idx - 1] + 1);
    }
}

}  // namespace
}  // namespace irn

using namespace irn;

extern "C" int irn_label_epilogue(int n_images, const float *const *rw_dev, const int32_t *c, const int32_t *h,
                                  const int32_t *w, const int32_t *out_h, const int32_t *out_w, float bg_thres,
                                  const int64_t *const *keys_dev, uint8_t *const *labels_dev,
                                  int32_t *const *argmax_dev, float *const *rw_up_dev, void *scratch_dev,
                                  void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_images < 1 || !rw_dev || !c || !h || !w || !out_h || !out_w || !scratch_dev)
        return fail(IRN_ERR_ARG, "irn_label_epilogue: bad argument");
    std::vector<LabelJob> jobs(n_images);
    long max_px = 0;
    for (int i = 0; i < n_images; ++i) {
        LabelJob &J = jobs[i];
        if (!rw_dev[i] || c[i] < 1 || h[i] < 1 || w[i] < 1 || out_h[i] < 1 || out_w[i] < 1 ||
            out_h[i] > 4 * h[i] || out_w[i] > 4 * w[i])
            return fail(IRN_ERR_ARG, "irn_label_epilogue: image %d: bad sizes (c=%d %dx%d -> %dx%d)", i, c[i], h[i],
                        w[i], out_h[i], out_w[i]);
        J.rw = rw_dev[i];
        J.keys = keys_dev ? keys_dev[i] : nullptr;
        J.labels = labels_dev ? labels_dev[i] : nullptr;
        J.argmax = argmax_dev ? argmax_dev[i] : nullptr;
        J.rw_up = rw_up_dev ? rw_up_dev[i] : nullptr;
        if (J.labels && !J.keys)
            return fail(IRN_ERR_ARG, "irn_label_epilogue: image %d: labels requested without keys", i);
        J.max_slot = (unsigned *)scratch_dev + i;
        J.c = c[i]; J.h = h[i]; J.w = w[i]; J.oh = out_h[i]; J.ow = out_w[i];
        max_px = std::max(max_px, (long)out_h[i] * out_w[i]);
    }
    LabelJob *jobs_dev = nullptr;
    int rc = scratch_upload(jobs.data(), sizeof(LabelJob) * n_images, (void **)&jobs_dev, stream);
    if (rc) return rc;
    IRN_HIP_TRY(hipMemsetAsync(scratch_dev, 0, sizeof(unsigned) * n_images, stream));
    const int bx = (int)std::min<long>((max_px + 255) / 256, 512);
    hipLaunchKernelGGL(upsample_max_kernel, dim3(bx, n_images), dim3(256), 0, stream, jobs_dev);
    IRN_LAUNCH_CHECK("upsample_max_kernel");
    hipLaunchKernelGGL(label_argmax_kernel, dim3(bx, n_images), dim3(256), 0, stream, jobs_dev, bg_thres);
    IRN_LAUNCH_CHECK("label_argmax_kernel");
    return scratch_release(stream);
}
