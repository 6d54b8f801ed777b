// Read-bandwidth of the memory hierarchy as a function of working-set size (gfx950).
// Answers one design question for the walk: do weight planes that fit the 256 MiB Infinity Cache
// stream faster than from HBM?   hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o /tmp/membw
//   usage: membw [wg_per_cu=8] [loads_in_flight=8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int INF>
__global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ src, size_t n_vec, int reps, float *sink) {
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * INF;
    for (int r = 0; r < reps; ++r) {
        for (size_t base = (size_t)blockIdx.x * 256 * INF + threadIdx.x; base < n_vec; base += stride) {
            f4 v[INF];
#pragma unroll
            for (int k = 0; k < INF; ++k) {
                const size_t i = base + (size_t)k * 256;
                v[k] = i < n_vec ? __builtin_nontemporal_load(&src[i]) : acc;
            }
#pragma unroll
            for (int k = 0; k < INF; ++k) acc += v[k];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int INF>
__global__ __launch_bounds__(256) void read_kernel_plain(const f4 *__restrict__ src, size_t n_vec, int reps, float *sink) {
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * INF;
    for (int r = 0; r < reps; ++r) {
        for (size_t base = (size_t)blockIdx.x * 256 * INF + threadIdx.x; base < n_vec; base += stride) {
            f4 v[INF];
#pragma unroll
            for (int k = 0; k < INF; ++k) {
                const size_t i = base + (size_t)k * 256;
                v[k] = i < n_vec ? src[i] : acc;
            }
#pragma unroll
            for (int k = 0; k < INF; ++k) acc += v[k];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main(int argc, char **argv) {
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t max_bytes = (size_t)4 << 30;
    f4 *buf;
    float *sink;
    hipMalloc(&buf, max_bytes);
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, max_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t sizes_mb[] = {8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048, 4096};
    printf("CUs %d, %d workgroups per CU\n", cus, wg_per_cu);
    printf("%8s %12s %12s\n", "MiB", "nt GB/s", "plain GB/s");
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n_vec = bytes / 16;
        const int reps = (int)(((size_t)16 << 30) / bytes) < 4 ? 4 : (int)(((size_t)16 << 30) / bytes);
        float best[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode)
            for (int trial = 0; trial < 3; ++trial) {
                hipEventRecord(e0);
                if (mode == 0)
                    hipLaunchKernelGGL(read_kernel<8>, dim3(cus * wg_per_cu), dim3(256), 0, 0, buf, n_vec, reps, sink);
                else
                    hipLaunchKernelGGL(read_kernel_plain<8>, dim3(cus * wg_per_cu), dim3(256), 0, 0, buf, n_vec, reps, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const float gbs = (float)((double)bytes * reps / (ms * 1e-3) / 1e9);
                if (gbs > best[mode]) best[mode] = gbs;
            }
        printf("%8zu %12.0f %12.0f\n", mb, best[0], best[1]);
    }
    return 0;
}
