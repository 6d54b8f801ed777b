// Would 16 waves per CU (128 VGPRs each, 19 neighbours x 4 pixels per lane) run the arithmetic phase of a resident-walk
// step faster than 8 waves (256 VGPRs, 38 neighbours x 4 pixels)?  The phase waits on LDS latency once per row segment
// (DESIGN.md §4 lesson 18); twice the waves per SIMD could hide it — if the part still fits its register budget.
// Isolated model of the phase, radius 10, one workgroup per CU: weights in registers, LDS window reads (next row's window
// ahead of this row's FMAs when PREFETCH), one fp32 chain per pixel, partial sums to LDS, one barrier per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/wave_split_probe.hip -o tools/bin/wave_split_probe
//   tools/bin/wave_split_probe          -> us per step for {8, 16} waves x {prefetch, no prefetch}
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>
#include <vector>

typedef float f4a __attribute__((ext_vector_type(4)));

constexpr int R = 10, H = R - 1, LW = 96, LH = 8 + 2 * H, HP = H;

struct Disc {
    int n = 0;
    signed char dy[4 * R * R] = {}, dx[4 * R * R] = {};
    constexpr Disc() {
        for (int y = -(R - 1); y <= R - 1; ++y)
            for (int x = -(R - 1); x <= R - 1; ++x)
                if ((y != 0 || x != 0) && x * x + y * y < R * R) {
                    dy[n] = (signed char)y;
                    dx[n] = (signed char)x;
                    ++n;
                }
    }
};
inline constexpr Disc kDisc{};

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int NS, int QI>
constexpr int row_lo(int dy) {
    for (int s = QI * NS; s < (QI + 1) * NS; ++s)
        if (kDisc.dy[s] == dy) return s;
    return 1 << 20;
}
template <int NS, int QI>
constexpr int row_hi(int dy) {
    for (int s = (QI + 1) * NS - 1; s >= QI * NS; --s)
        if (kDisc.dy[s] == dy) return s;
    return -1;
}
constexpr int slot_floor(int v, int hp) { return v - (((v + hp) % 4 + 4) % 4); }

template <int NS, int QI>
struct Rows {
    int n = 0;
    int dy[2 * R] = {};
    constexpr Rows() {
        for (int y = -(R - 1); y <= R - 1; ++y)
            if (row_lo<NS, QI>(y) <= row_hi<NS, QI>(y)) dy[n++] = y;
    }
};
template <int NS, int QI>
inline constexpr Rows<NS, QI> kRows{};

template <int NS, int QI, int DY>
struct RowInfo {
    static constexpr int lo = row_lo<NS, QI>(DY), hi = row_hi<NS, QI>(DY);
    static constexpr int c_lo = slot_floor(kDisc.dx[lo], HP), c_hi = kDisc.dx[hi] + 3;
    static constexpr int N4 = (c_hi - c_lo) / 4 + 1;
};

template <int NS, int QI, int DY>
__device__ __forceinline__ void load_window(float (&w)[24], const float *xrow) {
    using RW = RowInfo<NS, QI, DY>;
    const float *row = xrow + DY * LW + RW::c_lo;
    static_for<RW::N4>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = decltype(ik)::value;
        const f4a v = *reinterpret_cast<const f4a *>(row + 4 * k);
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    });
}
template <int NS, int QI, int DY>
__device__ __forceinline__ void fma_window(const float (&wr)[NS][4], const float (&w)[24], float (&pf)[4]) {
    using RW = RowInfo<NS, QI, DY>;
    static_for<RW::hi - RW::lo + 1>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = RW::lo + decltype(is)::value;
        constexpr int dx = kDisc.dx[s];
        constexpr int k = s - QI * NS;
        static_for<4>([&](auto ij) __attribute__((always_inline)) {
            constexpr int j = decltype(ij)::value;
            pf[j] = fmaf(wr[k][j], w[dx + j - RW::c_lo], pf[j]);
        });
    });
}

template <int NS, int QI, bool PREFETCH>
__device__ __forceinline__ void partial_sums(const float (&wr)[NS][4], const float *xrow, float (&pf)[4]) {
    constexpr int NR = kRows<NS, QI>.n;
    pf[0] = pf[1] = pf[2] = pf[3] = 0.f;
    if constexpr (!PREFETCH) {
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            float w1[24];
            load_window<NS, QI, kRows<NS, QI>.dy[decltype(ir)::value]>(w1, xrow);
            fma_window<NS, QI, kRows<NS, QI>.dy[decltype(ir)::value]>(wr, w1, pf);
        });
    } else {
        float w[2][24];
        load_window<NS, QI, kRows<NS, QI>.dy[0]>(w[0], xrow);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            constexpr int r = decltype(ir)::value;
            if constexpr (r + 1 < NR) load_window<NS, QI, kRows<NS, QI>.dy[r + 1 < NR ? r + 1 : r]>(w[(r + 1) & 1], xrow);
            __builtin_amdgcn_sched_barrier(0);
            fma_window<NS, QI, kRows<NS, QI>.dy[r]>(wr, w[r & 1], pf);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

template <int WAVES, bool PREFETCH>
__global__ __launch_bounds__(WAVES * 64) void probe(const float *__restrict__ wsrc, float *__restrict__ out, int n_steps) {
    constexpr int NS = 304 / WAVES;
    __shared__ __attribute__((aligned(16))) float xs[2][LH * LW];
    __shared__ __attribute__((aligned(16))) float part[WAVES * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ly = lane >> 3, lx = (lane & 7) * 4;
    for (int i = tid; i < 2 * LH * LW; i += WAVES * 64) xs[0][i] = 1e-3f * (float)(i % 97);
    float wr[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[k][j] = wsrc[((blockIdx.x * WAVES + wv) * NS + k) * 256 + lane * 4 + j];
    __syncthreads();
    float total = 0.f;
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
        const float *xrow = xs[s & 1] + (ly + H) * LW + lx + HP;
        float acc[4];
        static_for<WAVES>([&](auto iq) __attribute__((always_inline)) {
            if (wv == decltype(iq)::value) partial_sums<NS, decltype(iq)::value, PREFETCH>(wr, xrow, acc);
        });
        *reinterpret_cast<f4a *>(part + wv * 256 + lane * 4) = f4a{acc[0], acc[1], acc[2], acc[3]};
        __syncthreads();
        if (wv == 0) {                        // stand-in for the combine: consume the partials of the first pixels
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < WAVES; ++q) t += part[q * 256 + lane];
            total += t;
            xs[(s + 1) & 1][(ly + H) * LW + lx + HP] = t * 1e-6f;
        }
        __syncthreads();
    }
    if (tid < 64) out[blockIdx.x * 64 + lane] = total;
}

template <int WAVES, bool PREFETCH>
static void run(const char *name, const float *w, float *out, int n_cu) {
    const int n_steps = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<WAVES, PREFETCH>), dim3(n_cu), dim3(WAVES * 64), 0, 0, w, out, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<WAVES, PREFETCH>), dim3(n_cu), dim3(WAVES * 64), 0, 0, w, out, n_steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    printf("%-28s %.3f us per step (%s)\n", name, 1e3 * ms / n_steps, hipGetErrorString(err));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    std::vector<float> hw((size_t)n_cu * 304 * 256, 0.001f);
    float *w, *out;
    hipMalloc(&w, hw.size() * 4);
    hipMalloc(&out, (size_t)n_cu * 64 * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<8, true>("8 waves, prefetch", w, out, n_cu);
    run<8, false>("8 waves, row after row", w, out, n_cu);
    run<16, true>("16 waves, prefetch", w, out, n_cu);
    run<16, false>("16 waves, row after row", w, out, n_cu);
    return 0;
}
