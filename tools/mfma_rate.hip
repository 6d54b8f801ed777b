// Issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 from ONE wave per SIMD (the matrix-pipe walk of walk_mfma.hip):
//   mode 0  MFMAs only, 4 rotating accumulators, operands in VGPRs
//   mode 1  + one ds_read_b32 per MFMA through a ring of 8 registers (the B operands of the walk)
//   mode 2  + the A operand copied out of an AGPR first (v_accvgpr_read), as the compiler does for the walk's weights
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f4a __attribute__((ext_vector_type(4)));
constexpr int N = 352;

template <int MODE, int AHEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const float *__restrict__ src, float *out, int iters) {
    __shared__ float xs[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) xs[i] = 1e-3f * (float)(i % 61);
    float A[N];
#pragma unroll
    for (int r = 0; r < N; ++r) A[r] = src[r * 256 + tid];
    __syncthreads();
    f4a acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float *bp = xs + (tid & 63) + (tid >> 6) * 1500;     // consecutive lanes -> consecutive banks: conflict-free
    const long long t0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            const float b = bp[it & 7];
#pragma unroll
            for (int r = 0; r < N; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[r], b, acc[r & 3], 0, 0, 0);
        } else {
            float bq[AHEAD];
#pragma unroll
            for (int r = 0; r < AHEAD; ++r) bq[r] = bp[r * 3 + (it & 1)];
#pragma unroll
            for (int r = 0; r < N; ++r) {
                acc[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[r], bq[r % AHEAD], acc[r & 3], 0, 0, 0);
                if (r + AHEAD < N) bq[r % AHEAD] = bp[(r + AHEAD) * 3 + (it & 1)];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = wall_clock64();
    const f4a d = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    out[blockIdx.x * 256 + tid] = d[0] + d[1] + d[2] + d[3];
    if (tid == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE, int AHEAD>
static void run(const char *name, const float *src, float *out, int n_cu) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, AHEAD>), dim3(n_cu), dim3(256), 0, 0, src, out, 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, AHEAD>), dim3(n_cu), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = 1e6 * ms / ((double)iters * N);      // ns per MFMA of a wave
    printf("%-40s %.2f ns per MFMA = %.1f cycles at 2.4 GHz; chip %.1f TFLOP/s (%s)\n", name, per, per * 2.4,
           512.0 * 4 * n_cu / per / 1e3, hipGetErrorString(hipGetLastError()));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    float *src, *out;
    hipMalloc(&src, (size_t)N * 256 * 4);
    hipMemset(src, 0, (size_t)N * 256 * 4);
    hipMalloc(&out, (size_t)n_cu * 256 * 4);
    run<0, 8>("MFMA only (VGPR/AGPR operands)", src, out, n_cu);
    run<1, 4>("MFMA + ds_read_b32 ring of 4", src, out, n_cu);
    run<1, 8>("MFMA + ds_read_b32 ring of 8", src, out, n_cu);
    run<1, 16>("MFMA + ds_read_b32 ring of 16", src, out, n_cu);
    run<1, 32>("MFMA + ds_read_b32 ring of 32", src, out, n_cu);
    return 0;
}
