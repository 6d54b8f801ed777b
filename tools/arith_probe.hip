// Where does the 0.9 us arithmetic phase of a resident-walk step go?  Runs the product's partial_sums
// (irn_amd/csrc/walk_resident.hip) for radius 10 in isolation, one 512-thread workgroup per CU:
//   mode 0  as in the kernel: LDS window reads + FMAs, partial sums to LDS, barrier per step
//   mode 1  the same without the barrier
//   mode 2  FMAs only (windows taken from registers)      mode 3  LDS reads only (windows summed, no weights)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I irn_amd/csrc tools/arith_probe.hip -o tools/bin/arith_probe
#include "../irn_amd/csrc/walk_resident.hip"

namespace irn {
namespace {

template <int QI, int MODE>
__device__ __forceinline__ void probe_sums(const float (&wr)[Geom<10>::NS][4], const float *xrow, double (&acc)[4]) {
    using G = Geom<10>;
    constexpr int R = 10, H = G::H;
    static_for<2 * H + 1>([&](auto iy) __attribute__((always_inline)) {
        constexpr int dy = decltype(iy)::value - H;
        constexpr int lo = row_lo<R, QI>(dy), hi = row_hi<R, QI>(dy);
        if constexpr (lo <= hi) {
            constexpr int dxlo = kDisc<R>.dx[lo], dxhi = kDisc<R>.dx[hi];
            constexpr int c_lo = floor4(dxlo), c_hi = dxhi + 3;
            constexpr int N4 = (c_hi - c_lo) / 4 + 1;
            float win[4 * N4];
            const float *row = xrow + dy * G::LW + c_lo;
#pragma unroll
            for (int k = 0; k < N4; ++k) {
                if constexpr (MODE == 2) {
                    win[4 * k] = wr[k][0]; win[4 * k + 1] = wr[k][1]; win[4 * k + 2] = wr[k][2]; win[4 * k + 3] = wr[k][3];
                } else {
                    const f4a v = *reinterpret_cast<const f4a *>(row + 4 * k);
                    win[4 * k] = v.x; win[4 * k + 1] = v.y; win[4 * k + 2] = v.z; win[4 * k + 3] = v.w;
                }
            }
            float pf[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == 3) {
#pragma unroll
                for (int k = 0; k < N4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[j] += win[4 * k + j];
            } else {
                static_for<hi - lo + 1>([&](auto is) __attribute__((always_inline)) {
                    constexpr int s = lo + decltype(is)::value;
                    constexpr int dx = kDisc<R>.dx[s];
                    constexpr int k = s - QI * G::NS;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[j] = fmaf(wr[k][j], win[dx + j - c_lo], pf[j]);
                });
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += (double)pf[j];
        }
    });
}

// ---- explicit software pipelining: rows of the part as a compile-time list ----
template <int QI>
struct Rows {
    int n = 0;
    int dy[8] = {};
    constexpr Rows() {
        for (int y = -9; y <= 9; ++y)
            if (row_lo<10, QI>(y) <= row_hi<10, QI>(y)) dy[n++] = y;
    }
};
template <int QI>
inline constexpr Rows<QI> kRows{};

template <int QI, int RI>
struct RowInfo {
    static constexpr int dy = kRows<QI>.dy[RI];
    static constexpr int lo = row_lo<10, QI>(dy), hi = row_hi<10, QI>(dy);
    static constexpr int c_lo = floor4(kDisc<10>.dx[lo]), c_hi = kDisc<10>.dx[hi] + 3;
    static constexpr int N4 = (c_hi - c_lo) / 4 + 1;
};

template <int QI, int RI>
__device__ __forceinline__ void load_row(f4a (&w)[7], const float *xrow) {
    using RW = RowInfo<QI, RI>;
    const float *row = xrow + RW::dy * Geom<10>::LW + RW::c_lo;
#pragma unroll
    for (int k = 0; k < RW::N4; ++k) w[k] = *reinterpret_cast<const f4a *>(row + 4 * k);
}
template <int QI, int RI>
__device__ __forceinline__ void fma_row(const float (&wr)[Geom<10>::NS][4], const f4a (&w)[7], double (&acc)[4]) {
    using RW = RowInfo<QI, RI>;
    float pf[4] = {0.f, 0.f, 0.f, 0.f};
    static_for<RW::hi - RW::lo + 1>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = RW::lo + decltype(is)::value;
        constexpr int dx = kDisc<10>.dx[s];
        constexpr int k = s - QI * Geom<10>::NS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int base = 0;
            const int e = dx + j - RW::c_lo + base;
            pf[j] = fmaf(wr[k][j], w[e / 4][e % 4], pf[j]);
        }
    });
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += (double)pf[j];
}

// MODE 4: every window of the part is read first, then all FMAs.  MODE 5: the next row's window is
// read before the current row's FMAs (two windows live).
template <int QI, int MODE>
__device__ __forceinline__ void piped_sums(const float (&wr)[Geom<10>::NS][4], const float *xrow, double (&acc)[4]) {
    constexpr int NR = kRows<QI>.n;
    if constexpr (MODE == 4) {
        f4a w[NR][7];
        static_for<NR>([&](auto ir) __attribute__((always_inline)) { load_row<QI, decltype(ir)::value>(w[decltype(ir)::value], xrow); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) { fma_row<QI, decltype(ir)::value>(wr, w[decltype(ir)::value], acc); });
    } else {
        f4a w[2][7];
        load_row<QI, 0>(w[0], xrow);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            constexpr int r = decltype(ir)::value;
            if constexpr (r + 1 < NR) load_row<QI, r + 1>(w[(r + 1) & 1], xrow);
            __builtin_amdgcn_sched_barrier(0);
            fma_row<QI, r>(wr, w[r & 1], acc);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe_kernel(float *out, int iters,
                                                                                                 long long *ticks) {
    using G = Geom<10>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    double *part = reinterpret_cast<double *>(smem + G::XS_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * G::LH * G::LW; i += 512) xs[i] = 0.001f * (i % 97);
    float wr[G::NS][4];
#pragma unroll
    for (int k = 0; k < G::NS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[k][j] = 1e-3f * (float)((tid * 7 + k * 13 + j) % 101);
    const int ly = lane >> 3, lx = (lane & 7) * 4;
    __syncthreads();
    const long long t0 = wall_clock64();
    double tot = 0.0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const float *xrow = xs + (it & 1) * (G::LH * G::LW) + (ly + G::H) * G::LW + lx + G::HP;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (MODE >= 4) {
            switch (wv) {
                case 0: piped_sums<0, MODE>(wr, xrow, acc); break;
                case 1: piped_sums<1, MODE>(wr, xrow, acc); break;
                case 2: piped_sums<2, MODE>(wr, xrow, acc); break;
                case 3: piped_sums<3, MODE>(wr, xrow, acc); break;
                case 4: piped_sums<4, MODE>(wr, xrow, acc); break;
                case 5: piped_sums<5, MODE>(wr, xrow, acc); break;
                case 6: piped_sums<6, MODE>(wr, xrow, acc); break;
                default: piped_sums<7, MODE>(wr, xrow, acc); break;
            }
        } else {
            switch (wv) {
                case 0: probe_sums<0, MODE>(wr, xrow, acc); break;
                case 1: probe_sums<1, MODE>(wr, xrow, acc); break;
                case 2: probe_sums<2, MODE>(wr, xrow, acc); break;
                case 3: probe_sums<3, MODE>(wr, xrow, acc); break;
                case 4: probe_sums<4, MODE>(wr, xrow, acc); break;
                case 5: probe_sums<5, MODE>(wr, xrow, acc); break;
                case 6: probe_sums<6, MODE>(wr, xrow, acc); break;
                default: probe_sums<7, MODE>(wr, xrow, acc); break;
            }
        }
        double *pw = part + (it & 1) * (8 * 256) + wv * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) pw[j * 64] = acc[j];
        if (MODE == 0 || MODE == 6) __syncthreads();
        tot += acc[0];
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + tid] = (float)tot;
}

}  // namespace
}  // namespace irn

int main() {
    using namespace irn;
    using G = Geom<10>;
    float *out;
    long long *ticks;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&ticks, 256 * 8);
    const int iters = 4000;
    const char *names[6] = {"LDS reads + FMAs + barrier (as in the kernel)", "LDS reads + FMAs, no barrier", "FMAs only", "LDS reads only",
                            "all windows read first, then FMAs (no barrier)", "next row's window read ahead of the FMAs (no barrier)"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipFuncSetAttribute((const void *)probe_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<0>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
                case 1: hipFuncSetAttribute((const void *)probe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<1>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
                case 2: hipFuncSetAttribute((const void *)probe_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<2>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
                case 3: hipFuncSetAttribute((const void *)probe_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<3>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
                case 4: hipFuncSetAttribute((const void *)probe_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<4>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
                default: hipFuncSetAttribute((const void *)probe_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
                        hipLaunchKernelGGL(probe_kernel<5>, dim3(256), dim3(512), G::LDS_BYTES, 0, out, iters, ticks); break;
            }
            hipDeviceSynchronize();
        }
        long long h[256];
        hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += h[i];
        mean /= 256;
        printf("mode %d  %-48s %.3f us per step\n", mode, names[mode], mean * 0.01 / iters);
        fflush(stdout);
    }
    return 0;
}
