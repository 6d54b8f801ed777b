// VALU issue rate of v_fma_f32 vs v_pk_fma_f32 at 1 and 2 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
    float acc[8];
    f2 acc2[8];
    for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x * 0.001f + i; acc2[i] = f2{acc[i], acc[i] + 1.f}; }
    f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = fmaf(acc[i], a, b);
                else acc2[i] = __builtin_elementwise_fma(acc2[i], a2, b2);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double instr_per_wave = (double)iters * 16 * 8;
                const double waves_per_simd = threads / 256.0;
                const double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
                if (rep == 1)
                    printf("%s threads/CU=%d: %.3f ms, %.3f ns per wave-instruction per SIMD (= %.2f clk at 2.4 GHz); %.1f TFLOP/s\n",
                           mode ? "v_pk_fma_f32" : "v_fma_f32   ", threads, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4,
                           (mode ? 4.0 : 2.0) * 64 * instr_per_wave * (threads / 64) * 256 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
