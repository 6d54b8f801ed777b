// LDS read throughput per CU (gfx950): ds_read_b32 / b64 / b128, conflict-free, 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W>   // W = floats per read
__global__ __launch_bounds__(512) void k(float *out, int iters, long long *ticks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float *base = lds + wv * 1024 + lane * W;      // 64 lanes x W floats contiguous: conflict-free
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *p = base + ((it + r) & 3) * 256;
            if (W == 1) acc += *(const volatile float *)p;
            else if (W == 2) { f2 v = *(const volatile f2 *)p; acc += v.x + v.y; }
            else { f4 v = *(const volatile f4 *)p; acc += v.x + v.w; }
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    float *out; long long *ticks;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);
    const int iters = 20000;
    for (int w : {1, 2, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, out, iters, ticks);
            else if (w == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 65536, 0, out, iters, ticks);
            else hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, out, iters, ticks);
            hipDeviceSynchronize();
        }
        long long h[256]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
        const double ns = mean * 10.0;
        const double instr = (double)iters * 16 * 8;          // wave-instructions per CU
        printf("ds_read_b%-3d  %.2f ns per wave-instruction per CU (%.1f clk at 2.4 GHz), %.0f B/ns per CU = %.1f TB/s chip\n", 32 * w,
               ns / instr, ns / instr * 2.4, instr * 64 * 4 * w / ns, instr * 64 * 4 * w / ns * 256 / 1000);
    }
    return 0;
}
