// What would it cost to build the affinities INSIDE resident_kernel<10>'s job prologue?
//
// Today: affinity_unrolled_kernel<10> writes the 152 half-plane weight planes (0.49 ms per 192 images = 10 us of CU time per
// 8 x 32 tile), and the resident kernel's prologue loads a tile's 304 DIRECTED weights per pixel from them (7.7 us per job).
// Fused: the job would stage the edge tile + halo in LDS and compute, per lane, the weights of its wave's 38 directed
// neighbours for its 4 pixels — forward ones from the path that starts at the pixel, backward ones from the path that starts
// at the neighbour (w_d(p - d)), i.e. EVERY symmetric weight twice per tile (the two pixels of a pair sit in different
// lanes, usually different waves) — straight into 152 registers.
//
// This probe times exactly that prologue in the resident kernel's geometry: one 512-thread workgroup per CU, 8 waves x 38
// neighbours, 4 consecutive pixels per lane, compile-time path offsets (ds_read immediates), v_max3 chains, the integer
// power in fp64 like the product, all 152 results kept live at once.  Reference arithmetic: misc/indexing.py:91-109, :133.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/affinity_prologue_probe.hip -o tools/bin/affinity_prologue_probe
//   tools/bin/affinity_prologue_probe [jobs per workgroup = 64]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

constexpr int R = 10, H = R - 1, TH = 8, TW = 32, LH = TH + 2 * H, LW = TW + 2 * H + 2;   // 26 x 52 floats
constexpr int kWaves = 8;

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
constexpr int NS = kDisc.n / kWaves;   // 38
static_assert(kDisc.n == 304 && NS * kWaves == 304, "radius-10 disc");

// thick-segment cells of direction (y, x), y > 0 or (y == 0 and x > 0): lattice points of the bounding box with
// (y*px - x*py)^2 < y^2 + x^2 (misc/indexing.py:37-46)
struct Path {
    int n = 0;
    signed char cy[40] = {}, cx[40] = {};
    constexpr Path(int y, int x) {
        const int lsq = y * y + x * x;
        const int x_lo = x < 0 ? x : 0, x_hi = x < 0 ? 0 : x;
        for (int py = 0; py <= y; ++py)
            for (int px = x_lo; px <= x_hi; ++px) {
                const int cross = y * px - x * py;
                if (cross * cross < lsq) {
                    cy[n] = (signed char)py;
                    cx[n] = (signed char)px;
                    ++n;
                }
            }
    }
};

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// weight of directed neighbour S of the pixel at tb (tb = its cell in the LDS edge tile)
template <int S>
__device__ __forceinline__ float directed_weight(const float *tb, int beta_int) {
    constexpr int dy = kDisc.dy[S], dx = kDisc.dx[S];
    constexpr bool fwd = dy > 0 || (dy == 0 && dx > 0);
    constexpr int py = fwd ? dy : -dy, px = fwd ? dx : -dx;      // the half-plane direction whose weight this is
    constexpr int by = fwd ? 0 : dy, bx = fwd ? 0 : dx;          // path origin relative to the pixel (backward: the neighbour)
    constexpr Path P(py, px);
    constexpr auto off = [](int k) constexpr { return (by + P.cy[k]) * LW + bx + P.cx[k]; };
    float m = tb[off(0)];
    static_for<(P.n - 1) / 2>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = 1 + 2 * decltype(ik)::value;
        m = max3(m, tb[off(k)], tb[off(k + 1)]);
    });
    if constexpr ((P.n - 1) % 2 == 1) {
        const float v = tb[off(P.n - 1)];
        m = max3(m, v, v);
    }
    double b = (double)(1.0f - m), r = 1.0;
    for (int e = beta_int; e; e >>= 1) {
        if (e & 1) r *= b;
        b *= b;
    }
    return (float)r;
}

__device__ __forceinline__ void keep_live(float a, float b, float c, float d) { asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d)); }

template <int QI>
__device__ __forceinline__ void wave_part(const float *tb, int beta_int, float (&acc)[4]) {
    float wr[NS][4];
    static_for<NS>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = QI * NS + decltype(is)::value;
        static_for<4>([&](auto ij) __attribute__((always_inline)) {
            wr[decltype(is)::value][decltype(ij)::value] = directed_weight<s>(tb + decltype(ij)::value, beta_int);
        });
    });
    // all 152 live here, as in the product (they ARE its register-resident operator): an empty asm uses them in groups
    static_for<NS>([&](auto is) __attribute__((always_inline)) {
        keep_live(wr[decltype(is)::value][0], wr[decltype(is)::value][1], wr[decltype(is)::value][2], wr[decltype(is)::value][3]);
    });
    static_for<NS>([&](auto is) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += wr[decltype(is)::value][j];       // (the degree: what the product does next)
    });
}

__global__ __launch_bounds__(512) void prologue_probe(const float *__restrict__ edge, int h, int w, int jobs, int beta_int,
                                                       float *__restrict__ out) {
    __shared__ float tile[LH * LW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ly = lane >> 3, lx0 = (lane & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int job = 0; job < jobs; ++job) {
        const int t = (blockIdx.x * jobs + job) % ((h / TH) * (w / TW));
        const int ty0 = (t / (w / TW)) * TH, tx0 = (t % (w / TW)) * TW;
        __syncthreads();
        for (int i = threadIdx.x; i < LH * LW; i += 512) {
            const int yy = i / LW, xx = i - yy * LW;
            const int gy = ty0 + yy - H, gx = tx0 + xx - H;
            tile[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? edge[gy * w + gx] : 1.0f;
        }
        __syncthreads();
        const float *tb = tile + (ly + H) * LW + lx0 + H;
        switch (wave) {
        case 0: wave_part<0>(tb, beta_int, acc); break;
        case 1: wave_part<1>(tb, beta_int, acc); break;
        case 2: wave_part<2>(tb, beta_int, acc); break;
        case 3: wave_part<3>(tb, beta_int, acc); break;
        case 4: wave_part<4>(tb, beta_int, acc); break;
        case 5: wave_part<5>(tb, beta_int, acc); break;
        case 6: wave_part<6>(tb, beta_int, acc); break;
        default: wave_part<7>(tb, beta_int, acc); break;
        }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char **argv) {
    const int jobs = argc > 1 ? atoi(argv[1]) : 64;
    const int h = 128, w = 128;
    std::vector<float> e((size_t)h * w);
    for (size_t i = 0; i < e.size(); ++i) e[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f * 0.6f;
    float *d_e = nullptr, *d_o = nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int n_wg = prop.multiProcessorCount;
    hipMalloc(&d_e, e.size() * 4);
    hipMalloc(&d_o, (size_t)n_wg * 512 * 4);
    hipMemcpy(d_e, e.data(), e.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(prologue_probe, dim3(n_wg), dim3(512), 0, nullptr, d_e, h, w, jobs, 10, d_o);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        printf("fused-affinity prologue, radius 10, %d workgroups x %d jobs: %.3f ms per launch = %.2f us per job "
               "(today: 7.7 us weight load in the job + 10 us of CU time per tile in affinity_unrolled_kernel<10>)\n",
               n_wg, jobs, ms, 1e3 * ms / jobs);
    }
    std::vector<float> o((size_t)n_wg * 512);
    hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
    double s = 0;
    for (float v : o) s += v;
    printf("checksum %.6e\n", s);
    return 0;
}
