// Label epilogue (gfx950): bilinear x4 upsample + crop + /global-max + background plane + argmax +
// keys look-up: the global maximum from two passes over the SOURCE cells (bounded search, below) and one pass
// over the output pixels; the fp32 [C,H,W] upsampled tensor is never materialised unless the caller asks
// for it (instance scoring).
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

// Every rounding in this file is part of the contract (labels and CAMs must reproduce ATen's values).
// HIP's __fmul_rn / __fsub_rn are inline `x * y` / `x - y` compiled under the HEADER's contraction
// mode, so after inlining scale * (dst + 0.5) - 0.5 still becomes an fma — invisible for the dyadic
// x4 scale of the label epilogue, a few ulp off for the arbitrary scales of the CAM merge.  Hence:
// plain operators under this pragma for every separately rounded step, fmaf where ATen fuses.
#pragma clang fp contract(off)

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
    float src = ((float)dst + 0.5f) * 0.25f;
    src = src - 0.5f;
    src = src < 0.f ? 0.f : src;
    Taps t;
    t.i0 = min((int)src, n_in - 1);
    t.i1 = min(t.i0 + 1, n_in - 1);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ plane, int w, const Taps &ty, const Taps &tx) {
    const float v00 = plane[ty.i0 * w + tx.i0], v01 = plane[ty.i0 * w + tx.i1];
    const float v10 = plane[ty.i1 * w + tx.i0], v11 = plane[ty.i1 * w + tx.i1];
    const float p01 = v01 * tx.l1, p11 = v11 * tx.l1;
    const float top = __builtin_fmaf(v00, tx.l0, p01);
    const float bot = __builtin_fmaf(v10, tx.l0, p11);
    const float pb = ty.l1 * bot;
    return __builtin_fmaf(ty.l0, top, pb);
}

// Global maximum of the upsampled scores WITHOUT evaluating every output pixel.  An output pixel with taps
// (i0, i1) x (j0, j1) is a convex combination of the 2x2 source cell (i0, j0), so max(cell) bounds it from above; the
// outputs of cell (k, l) are the rows 4k+2 .. 4k+5 (0 .. 5 for k = 0) and the same in x.  Pass 1 evaluates ONE output
// per cell (a lower bound L of the maximum); pass 2 evaluates all outputs of the cells whose bound reaches L — a handful
// around the peak — with the same `bilerp` as the argmax pass, so the maximum is the identical float.  (The one-pass
// form interpolated every output pixel twice: 0.65 ms of a 31.7 ms step.)  The bound carries a few-ulp margin: the
// rounded fma chain of `bilerp` can exceed max(cell) in the last bit.
__device__ __forceinline__ int cell_lo(int k) { return k == 0 ? 0 : 4 * k + 2; }

template <int PASS>
__global__ __launch_bounds__(256) void upsample_max_kernel(const LabelJob *__restrict__ jobs) {
    const LabelJob J = jobs[blockIdx.y];
    const unsigned plane_px = (unsigned)(J.h * J.w), cells = (unsigned)J.c * plane_px;     // < 2^31: checked by the host
    const float lower = PASS == 2 ? dec_ordered(*J.max_slot) : 0.f;
    float m = -INFINITY;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int c = (int)(i / plane_px);
        const int r = (int)(i - (unsigned)c * plane_px);
        const int k = (int)((unsigned)r / (unsigned)J.w), l = r - k * J.w;
        const int y0 = cell_lo(k), x0 = cell_lo(l);
        if (y0 >= J.oh || x0 >= J.ow) continue;                       // cell lies entirely in the cropped margin
        const float *plane = J.rw + (long)c * J.h * J.w;
        if (PASS == 1) {
            m = fmaxf(m, bilerp(plane, J.w, taps_x4(y0, J.h), taps_x4(x0, J.w)));
        } else {
            const int k1 = min(k + 1, J.h - 1), l1 = min(l + 1, J.w - 1);
            const float u = fmaxf(fmaxf(plane[k * J.w + l], plane[k * J.w + l1]), fmaxf(plane[k1 * J.w + l], plane[k1 * J.w + l1]));
            if (u + fabsf(u) * 1e-6f + 1e-37f < lower) continue;
            const int y1 = min(k == J.h - 1 ? 4 * J.h - 1 : 4 * k + 5, J.oh - 1);
            const int x1 = min(l == J.w - 1 ? 4 * J.w - 1 : 4 * l + 5, J.ow - 1);
            for (int oy = y0; oy <= y1; ++oy) {
                const Taps ty = taps_x4(oy, J.h);
                for (int ox = x0; ox <= x1; ++ox) m = fmaxf(m, bilerp(plane, J.w, ty, taps_x4(ox, J.w)));
            }
        }
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

__device__ __forceinline__ float bilerp_vals(float v00, float v01, float v10, float v11, const Taps &ty, const Taps &tx) {
    const float p01 = v01 * tx.l1, p11 = v11 * tx.l1;              // the arithmetic of `bilerp`, values already loaded
    const float top = __builtin_fmaf(v00, tx.l0, p01);
    const float bot = __builtin_fmaf(v10, tx.l0, p11);
    const float pb = ty.l1 * bot;
    return __builtin_fmaf(ty.l0, top, pb);
}

// One thread = FOUR horizontally adjacent output pixels 4j .. 4j+3 of a row: at x4 the first two share one pair of
// source columns and the last two the next pair, so a channel costs 8 loads instead of 16, the index arithmetic (a
// division by the row length) is paid once per four pixels, and the labels leave as one 4-byte store per thread (256
// contiguous bytes per wave) instead of four single bytes.  The per-pixel arithmetic is `bilerp` unchanged.  (One
// pixel per thread with 64-bit index arithmetic: 0.140 ms per 192 images of 512^2, 0.36 Tpixel/s.)
__global__ __launch_bounds__(256) void label_argmax_kernel(const LabelJob *__restrict__ jobs, float bg) {
    const LabelJob J = jobs[blockIdx.y];
    const unsigned gw = (unsigned)(J.ow + 3) >> 2, groups = gw * (unsigned)J.oh;
    const unsigned npx = (unsigned)J.oh * (unsigned)J.ow;
    const float gmax = dec_ordered(*J.max_slot);
    for (unsigned g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
        const int oy = (int)(g / gw), ox = (int)(g - (unsigned)oy * gw) * 4;
        const int nq = min(4, J.ow - ox);
        const Taps ty = taps_x4(oy, J.h);
        Taps tx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tx[q] = taps_x4(min(ox + q, J.ow - 1), J.w);
        float best[4] = {bg, bg, bg, bg};
        int idx[4] = {0, 0, 0, 0};
        const unsigned o = (unsigned)oy * (unsigned)J.ow + (unsigned)ox;
        for (int c = 0; c < J.c; ++c) {
            const float *r0 = J.rw + (long)c * J.h * J.w + ty.i0 * J.w, *r1 = J.rw + (long)c * J.h * J.w + ty.i1 * J.w;
            // pixels 0, 1 read columns tx[0].{i0, i1}, pixels 2, 3 columns tx[2].{i0, i1} (tx[1] == tx[0]'s, tx[3] == tx[2]'s:
            // src = j - 0.375, j - 0.125, j + 0.125, j + 0.375, clamped at 0)
            const float a00 = r0[tx[0].i0], a01 = r0[tx[0].i1], a10 = r1[tx[0].i0], a11 = r1[tx[0].i1];
            const float b00 = r0[tx[2].i0], b01 = r0[tx[2].i1], b10 = r1[tx[2].i0], b11 = r1[tx[2].i1];
            float v[4];
            v[0] = bilerp_vals(a00, a01, a10, a11, ty, tx[0]) / gmax;
            v[1] = bilerp_vals(a00, a01, a10, a11, ty, tx[1]) / gmax;
            v[2] = bilerp_vals(b00, b01, b10, b11, ty, tx[2]) / gmax;
            v[3] = bilerp_vals(b00, b01, b10, b11, ty, tx[3]) / gmax;
            if (J.rw_up) {
                float *dst = J.rw_up + (long)c * npx + o;
                if (nq == 4 && ((uintptr_t)dst & 15) == 0) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int q = 0; q < nq; ++q) dst[q] = v[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] > best[q]) {
                    best[q] = v[q];
                    idx[q] = c + 1;
                }
        }
        if (J.argmax) {
            int32_t *dst = J.argmax + o;
            if (nq == 4 && ((uintptr_t)dst & 15) == 0) *reinterpret_cast<int4 *>(dst) = make_int4(idx[0], idx[1], idx[2], idx[3]);
            else
                for (int q = 0; q < nq; ++q) dst[q] = idx[q];
        }
        if (J.labels) {
            uint8_t lab[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) lab[q] = idx[q] == 0 ? (uint8_t)0 : (uint8_t)(J.keys[idx[q] - 1] + 1);
            uint8_t *dst = J.labels + o;
            if (nq == 4 && ((uintptr_t)dst & 3) == 0)
                *reinterpret_cast<uint32_t *>(dst) = (uint32_t)lab[0] | ((uint32_t)lab[1] << 8) | ((uint32_t)lab[2] << 16) | ((uint32_t)lab[3] << 24);
            else
                for (int q = 0; q < nq; ++q) dst[q] = lab[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-scale CAM merge (reference step/make_cam.py:38-52): for both target grids — the stride-4
// grid (ceil(H/4), ceil(W/4)) and the stride-16-rounded full size cropped to (H, W) — sum over the
// scales of F.interpolate(size=..., bilinear, align_corners=False), keep the present classes, divide
// every channel by (its max + 1e-5).  The reference issues 2 x n_scales interpolate launches, two
// stacks + sums, two adaptive_max_pool2d (one serial thread per channel on the GPU: 24 ms at 512^2)
// and two divides; here: one pass that writes the sums and the channel maxima, one that divides.
// Arithmetic = ATen's CPU kernel as restated in oracle/irn_oracle.py (resize_bilinear, cam_merge).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxScales = 8;
struct MergeArgs {
    const float *src[kMaxScales];   // [n_cls, hs, ws]
    int hs[kMaxScales], ws[kMaxScales];
    int n_scales, n_cls, k;
    const int64_t *keys;            // [k] present classes
    float *lo, *hi;                 // [k, lh, lw], [k, H, W]
    int lh, lw, uh, uw, H, W;       // (uh, uw): interpolation target of `hi` before the crop to (H, W)
    unsigned *max_slots;            // [2k] ordered encoding: lo maxima, then hi maxima
};

__device__ __forceinline__ Taps taps_resize(int dst, int n_in, int n_out) {
    // area_pixel_compute_source_index(scale = n_in / n_out in fp32, align_corners = false), clamped at 0
    const float scale = (float)n_in / (float)n_out;
    float src = scale * ((float)dst + 0.5f);
    src = src - 0.5f;
    src = src < 0.f ? 0.f : src;
    Taps t;
    t.i0 = min((int)src, n_in - 1);
    t.i1 = min(t.i0 + 1, n_in - 1);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

__global__ __launch_bounds__(256) void cam_merge_sum_kernel(const MergeArgs A) {
    const int kk = blockIdx.y;
    const bool is_hi = blockIdx.z == 1;
    const int oh = is_hi ? A.H : A.lh, ow = is_hi ? A.W : A.lw;      // written (cropped) extent
    const int th = is_hi ? A.uh : A.lh, tw = is_hi ? A.uw : A.lw;    // interpolation target
    float *out = (is_hi ? A.hi : A.lo) + (long)kk * oh * ow;
    const int cls = (int)A.keys[kk];
    float m = -INFINITY;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < (long)oh * ow; o += (long)gridDim.x * 256) {
        const int oy = (int)(o / ow), ox = (int)(o - (long)oy * ow);
        float acc = 0.f;
        for (int s = 0; s < A.n_scales; ++s) {
            const Taps ty = taps_resize(oy, A.hs[s], th), tx = taps_resize(ox, A.ws[s], tw);
            acc = acc + bilerp(A.src[s] + (long)cls * A.hs[s] * A.ws[s], A.ws[s], ty, tx);
        }
        out[o] = acc;
        m = fmaxf(m, acc);
    }
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > -INFINITY) atomicMax(A.max_slots + (is_hi ? A.k : 0) + kk, enc_ordered(m));
    }
}

__global__ __launch_bounds__(256) void cam_merge_norm_kernel(const MergeArgs A) {
    const int kk = blockIdx.y;
    const bool is_hi = blockIdx.z == 1;
    const long n = is_hi ? (long)A.H * A.W : (long)A.lh * A.lw;
    float *out = (is_hi ? A.hi : A.lo) + (long)kk * n;
    const float den = dec_ordered(A.max_slots[(is_hi ? A.k : 0) + kk]) + 1e-5f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long)gridDim.x * 256) out[o] = out[o] / den;
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
    long max_px = 0, max_cells = 0;
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
        max_cells = std::max(max_cells, (long)c[i] * h[i] * w[i]);
    }
    LabelJob *jobs_dev = nullptr;
    int rc = scratch_upload(jobs.data(), sizeof(LabelJob) * n_images, (void **)&jobs_dev, stream);
    if (rc) return rc;
    IRN_HIP_TRY(hipMemsetAsync(scratch_dev, 0, sizeof(unsigned) * n_images, stream));
    if (max_px >= (1L << 31) || max_cells >= (1L << 31)) return fail(IRN_ERR_ARG, "irn_label_epilogue: image too large (32-bit pixel indices)");
    // The two maximum passes: ~4096 workgroups per launch whatever the batch (16 per CU), each thread striding over several
    // source cells — with one cell per thread (15 000 workgroups per 192-image batch) they took 66 + 37 us, now 22 + 19.
    // The argmax pass is the opposite: its iterations are chains of dependent memory operations (score loads -> division ->
    // keys look-up -> store), and one 4-pixel group per thread (49 000 workgroups) hides them better than twelve (92 vs
    // 104 us per 192 images; profiles/r04_s20_label_kernels.txt).
    const long per_image = std::max<long>(1, 4096 / n_images);
    const int bx = (int)std::min<long>((max_px / 4 + 255) / 256 + 1, 512);
    const int bc = (int)std::min<long>((max_cells + 255) / 256, per_image);
    hipLaunchKernelGGL(upsample_max_kernel<1>, dim3(bc, n_images), dim3(256), 0, stream, jobs_dev);
    IRN_LAUNCH_CHECK("upsample_max_kernel<1>");
    hipLaunchKernelGGL(upsample_max_kernel<2>, dim3(bc, n_images), dim3(256), 0, stream, jobs_dev);
    IRN_LAUNCH_CHECK("upsample_max_kernel<2>");
    hipLaunchKernelGGL(label_argmax_kernel, dim3(bx, n_images), dim3(256), 0, stream, jobs_dev, bg_thres);
    IRN_LAUNCH_CHECK("label_argmax_kernel");
    return scratch_release(stream);
}

extern "C" int irn_cam_merge(int n_scales, const float *const *src_dev, const int32_t *hs, const int32_t *ws, int n_classes,
                             const int64_t *keys_dev, int n_keys, int out_h, int out_w, float *cam_dev,
                             float *high_res_dev, void *scratch_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_scales < 1 || n_scales > kMaxScales || !src_dev || !hs || !ws || n_classes < 1 || !keys_dev || n_keys < 1 ||
        out_h < 1 || out_w < 1 || !cam_dev || !high_res_dev || !scratch_dev)
        return fail(IRN_ERR_ARG, "irn_cam_merge: bad argument (1..%d scales, >= 1 key)", kMaxScales);
    MergeArgs A;
    for (int s = 0; s < n_scales; ++s) {
        if (!src_dev[s] || hs[s] < 1 || ws[s] < 1) return fail(IRN_ERR_ARG, "irn_cam_merge: scale %d: bad input", s);
        A.src[s] = src_dev[s];
        A.hs[s] = hs[s];
        A.ws[s] = ws[s];
    }
    A.n_scales = n_scales;
    A.n_cls = n_classes;
    A.k = n_keys;
    A.keys = keys_dev;
    A.lo = cam_dev;
    A.hi = high_res_dev;
    A.H = out_h;
    A.W = out_w;
    A.lh = (out_h - 1) / 4 + 1;              // misc/imutils.py get_strided_size(size, 4)
    A.lw = (out_w - 1) / 4 + 1;
    A.uh = ((out_h - 1) / 16 + 1) * 16;      // get_strided_up_size(size, 16)
    A.uw = ((out_w - 1) / 16 + 1) * 16;
    A.max_slots = (unsigned *)scratch_dev;
    IRN_HIP_TRY(hipMemsetAsync(scratch_dev, 0, sizeof(unsigned) * 2 * n_keys, stream));
    const int bx = (int)std::min<long>(((long)out_h * out_w + 255) / 256, 256);
    hipLaunchKernelGGL(cam_merge_sum_kernel, dim3(bx, n_keys, 2), dim3(256), 0, stream, A);
    IRN_LAUNCH_CHECK("cam_merge_sum_kernel");
    hipLaunchKernelGGL(cam_merge_norm_kernel, dim3(bx, n_keys, 2), dim3(256), 0, stream, A);
    IRN_LAUNCH_CHECK("cam_merge_norm_kernel");
    return IRN_OK;
}
