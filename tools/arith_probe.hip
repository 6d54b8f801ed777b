// Round 6: can the arithmetic phase of a resident-walk step (radius 10) stop being LDS-bound by SHARING the state
// window across lanes?  Adjacent lanes' windows overlap by 20 of 24 floats; here each lane reads only its own aligned
// 16-byte slot of a neighbour row (+ one halo slot for the lanes at the ends of a tile row) and takes the other slots
// from its neighbour lanes through DPP.  One 512-thread workgroup per CU, the product's weights-in-registers budget.
//
//   mode 0  the shipped phase: partial_sums<10, QI, 2> of irn_amd/csrc/walk_resident.hip (8 x 32 tile, 6-7 ds_read_b128
//           per neighbour row), partial sums to LDS, barrier
//   mode 1  the same without the barrier
//   mode 2  4 x 64 tile (one DPP row of 16 lanes = one tile row), windows from LDS as in mode 0 (layout baseline)
//   mode 3  4 x 64 tile, DPP sharing: 2 ds_read_b128 per neighbour row (own slot A, halo slot B); a slot k lanes away is
//           row_ror of (lane < k ? B : A): one v_cndmask per window element, rotation fused into the FMA
//   mode 4  as 3 but every row's A/B is read before the first FMA (one s_waitcnt per step)
//   mode 5  4 x 64 tile, own slot only, out-of-row sources = 0 (WRONG at the tile's left/right edge: the ceiling of any
//           sharing scheme — what mode 3 would cost if the halo were free)
//   mode 6  FMAs only (windows from registers): the VALU floor
// Modes 2-4 must agree bit for bit (same fp32 chains in the same order); the host checks it.
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
//       -I irn_amd/csrc tools/arith_probe.hip -o tools/bin/arith_probe
#include "../irn_amd/csrc/walk_resident.hip"

namespace irn {
namespace {

using G10 = Geom<10>;
constexpr int NS10 = G10::NS;
// 4 x 64 tile: lane = (row r = lane >> 4, slot i = lane & 15), pixels 4i .. 4i+3 of tile row r
constexpr int TH2 = 4, TW2 = 64, HP2 = 12, LW2 = 128, LH2 = TH2 + 18;
static_assert(LW2 % 64 == 0, "16-lane groups of a ds_read_b128 then fall on distinct banks (guide, LDS table)");

template <int QI>
struct Rows {
    int n = 0;
    int dy[20] = {};
    constexpr Rows() {
        for (int y = -9; y <= 9; ++y)
            if (row_lo<10, QI>(y) <= row_hi<10, QI>(y)) dy[n++] = y;
    }
};
template <int QI>
inline constexpr Rows<QI> kRows{};

template <int QI, int RI>
struct RowD {
    static constexpr int dy = kRows<QI>.dy[RI];
    static constexpr int lo = row_lo<10, QI>(dy), hi = row_hi<10, QI>(dy);
    static constexpr int dxlo = kDisc<10>.dx[lo], dxhi = kDisc<10>.dx[hi];
    static constexpr int klo = (dxlo - ((dxlo % 4 + 4) % 4)) / 4, khi = (dxhi + 3 - (((dxhi + 3) % 4 + 4) % 4)) / 4;
};
// index of neighbour (dy, dx) inside part QI, or -1 (the centre is not a neighbour)
template <int QI>
constexpr int find_s(int lo, int hi, int dx) {
    for (int s = lo; s <= hi; ++s)
        if (kDisc<10>.dx[s] == dx) return s - QI * NS10;
    return -1;
}

// ---- mode 2: windows from LDS, next row's window ahead of this row's FMAs ----
template <int QI, int RI>
__device__ __forceinline__ void win_load(f4a (&w)[7], const float *arow) {
    using RD = RowD<QI, RI>;
    const float *row = arow + RD::dy * LW2 + 4 * RD::klo;
#pragma unroll
    for (int k = 0; k <= RD::khi - RD::klo; ++k) w[k] = *reinterpret_cast<const f4a *>(row + 4 * k);
}
template <int QI, int RI>
__device__ __forceinline__ void win_fma(const float (&wr)[NS10][4], const f4a (&w)[7], float (&pf)[4]) {
    using RD = RowD<QI, RI>;
    static_for<4 * (RD::khi - RD::klo + 1)>([&](auto ie) __attribute__((always_inline)) {
        constexpr int e = 4 * RD::klo + decltype(ie)::value;
        static_for<4>([&](auto ij) __attribute__((always_inline)) {
            constexpr int j = decltype(ij)::value;
            constexpr int s = find_s<QI>(RD::lo, RD::hi, e - j);
            if constexpr (s >= 0) pf[j] = fmaf(wr[s][j], w[(e - 4 * RD::klo) / 4][(e - 4 * RD::klo) % 4], pf[j]);
        });
    });
}
template <int QI>
__device__ __forceinline__ void sums_lds(const float (&wr)[NS10][4], const float *arow, float (&pf)[4]) {
    constexpr int NR = kRows<QI>.n;
    f4a w[2][7];
    win_load<QI, 0>(w[0], arow);
    static_for<NR>([&](auto ir) __attribute__((always_inline)) {
        constexpr int r = decltype(ir)::value;
        if constexpr (r + 1 < NR) win_load<QI, (r + 1 < NR ? r + 1 : r)>(w[(r + 1) & 1], arow);
        __builtin_amdgcn_sched_barrier(0);
        win_fma<QI, r>(wr, w[r & 1], pf);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- modes 3-5: DPP sharing ----
// A = the lane's own slot of neighbour row dy, B = slot i+16 (lanes 0-2) / i-16 (lanes 13-15) of the same row.
// Slot i+k (k > 0) is lane (i+k) mod 16's A unless i+k >= 16, then lane i+k-16's B: every source lane s feeds exactly
// one reader, and it must hand over B iff s < k — a select at the SOURCE, then a rotation (no invalid lanes).
// The DPP operand is folded into the FMA by hand (v_fmac_f32_dpp): LLVM's DPP combine runs while the FMAs are still
// three-address VOP3 and leaves a v_mov_b32_dpp per element (+37 VALU instructions per wave and step).  All selects of a
// row come first, then `s_nop 1` (VALU write -> DPP read of the same VGPR needs 2 wait states; nobody checks inline asm).
template <int CTRL>
__device__ __forceinline__ void fmac_dpp(float &acc, float src, float w) {
    if constexpr ((CTRL & 0xff0) == 0x120) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(CTRL & 15));
    else if constexpr ((CTRL & 0xff0) == 0x100) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(CTRL & 15));
    else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(CTRL & 15));
}
__device__ __forceinline__ float pick(float a, float b, unsigned long long m) {
    float r;
    asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
template <int QI, int RI, bool EXACT>
__device__ __forceinline__ void dpp_fma(const float (&wr)[NS10][4], const f4a &A, const f4a &B, float (&pf)[4],
                                        const unsigned long long (&mask)[7]) {
    using RD = RowD<QI, RI>;
    float sel[7][4];
    if constexpr (EXACT) {
        static_for<RD::khi - RD::klo + 1>([&](auto ik) __attribute__((always_inline)) {
            constexpr int k = RD::klo + decltype(ik)::value;
            static_for<4>([&](auto im) __attribute__((always_inline)) {
                constexpr int m = decltype(im)::value, e = 4 * k + m;
                if constexpr (k != 0 && e >= RD::dxlo && e <= RD::dxhi + 3)
                    sel[k + 3][m] = pick(A[m], B[m], mask[k + 3]);
            });
        });
        asm volatile("s_nop 1");
    }
    static_for<RD::khi - RD::klo + 1>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = RD::klo + decltype(ik)::value;
        static_for<4>([&](auto im) __attribute__((always_inline)) {
            constexpr int m = decltype(im)::value, e = 4 * k + m;
            if constexpr (e >= RD::dxlo && e <= RD::dxhi + 3) {
                static_for<4>([&](auto ij) __attribute__((always_inline)) {
                    constexpr int j = decltype(ij)::value;
                    constexpr int s = find_s<QI>(RD::lo, RD::hi, e - j);
                    if constexpr (s >= 0) {
                        if constexpr (k == 0) pf[j] = fmaf(wr[s][j], A[m], pf[j]);
                        else if constexpr (EXACT) fmac_dpp<0x120 + (k > 0 ? 16 - k : -k)>(pf[j], sel[k + 3][m], wr[s][j]);   // lane i <- lane (i + k) mod 16
                        else fmac_dpp<(k > 0 ? 0x100 + k : 0x110 - k)>(pf[j], A[m], wr[s][j]);      // sources outside the row: lane disabled
                    }
                });
            }
        });
    });
}
template <int QI, int RI>
__device__ __forceinline__ void ab_load(f4a &A, f4a &B, const float *arow, const float *brow) {
    A = *reinterpret_cast<const f4a *>(arow + RowD<QI, RI>::dy * LW2);
    B = *reinterpret_cast<const f4a *>(brow + RowD<QI, RI>::dy * LW2);
}
template <int QI, int MODE>
__device__ __forceinline__ void sums_dpp(const float (&wr)[NS10][4], const float *arow, const float *brow, float (&pf)[4],
                                         const unsigned long long (&mask)[7]) {
    constexpr int NR = kRows<QI>.n;
    if constexpr (MODE == 4) {
        f4a A[NR], B[NR];
        static_for<NR>([&](auto ir) __attribute__((always_inline)) { ab_load<QI, decltype(ir)::value>(A[decltype(ir)::value], B[decltype(ir)::value], arow, brow); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) { dpp_fma<QI, decltype(ir)::value, true>(wr, A[decltype(ir)::value], B[decltype(ir)::value], pf, mask); });
    } else {
        f4a A[2], B[2];
        ab_load<QI, 0>(A[0], B[0], arow, brow);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            constexpr int r = decltype(ir)::value;
            if constexpr (r + 1 < NR) ab_load<QI, (r + 1 < NR ? r + 1 : r)>(A[(r + 1) & 1], B[(r + 1) & 1], arow, brow);
            __builtin_amdgcn_sched_barrier(0);
            dpp_fma<QI, r, MODE == 3>(wr, A[r & 1], B[r & 1], pf, mask);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// ---- mode 6: FMAs only ----
template <int QI>
__device__ __forceinline__ void sums_regs(const float (&wr)[NS10][4], float (&pf)[4], float seed) {
    constexpr int NR = kRows<QI>.n;
    static_for<NR>([&](auto ir) __attribute__((always_inline)) {
        f4a w[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) w[k] = f4a{seed + k, seed * 2.f + k, seed * 3.f + k, seed * 4.f + k};
        win_fma<QI, decltype(ir)::value>(wr, w, pf);
    });
}

#define PROBE_SWITCH(CALL)            \
    switch (wv) {                     \
        case 0: { constexpr int QI = 0; CALL; } break; \
        case 1: { constexpr int QI = 1; CALL; } break; \
        case 2: { constexpr int QI = 2; CALL; } break; \
        case 3: { constexpr int QI = 3; CALL; } break; \
        case 4: { constexpr int QI = 4; CALL; } break; \
        case 5: { constexpr int QI = 5; CALL; } break; \
        case 6: { constexpr int QI = 6; CALL; } break; \
        default: { constexpr int QI = 7; CALL; } break; \
    }

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe_kernel(float *out, int iters, long long *ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    constexpr int XS_FLOATS = (MODE <= 1) ? G10::LH * G10::LW : LH2 * LW2;
    float *partf = xs + 2 * XS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * XS_FLOATS; i += 512) xs[i] = 1e-3f * (float)((i * 2654435761u >> 20) & 1023u);
    float wr[NS10][4];
#pragma unroll
    for (int k = 0; k < NS10; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[k][j] = 1e-3f * (float)(((tid & 255) * 7 + k * 13 + j) % 101);   // same weights in every wave's lane l
    __syncthreads();
    const long long t0 = wall_clock64();
    float tot = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        float pf[4] = {0.f, 0.f, 0.f, 0.f};
        const float *xb = xs + (it & 1) * XS_FLOATS;
        if constexpr (MODE <= 1) {
            const int ly = lane >> 3, lx = (lane & 7) * 4;
            const float *xrow = xb + (ly + G10::H) * G10::LW + lx + G10::HP;
            PROBE_SWITCH((partial_sums<10, QI, 2>(wr, xrow, pf)))
        } else if constexpr (MODE == 6) {
            float seed = xb[lane];
            PROBE_SWITCH((sums_regs<QI>(wr, pf, seed)))
        } else {
            const int r = lane >> 4, i = lane & 15;
            const int sb = i < 3 ? i + 16 : (i >= 13 ? i - 16 : i);
            const float *arow = xb + (r + 9) * LW2 + HP2 + 4 * i;
            const float *brow = xb + (r + 9) * LW2 + HP2 + 4 * sb;
            if constexpr (MODE == 2) { PROBE_SWITCH((sums_lds<QI>(wr, arow, pf))) }
            else {
                // lane masks of the selects: slot i+k comes from lane (i+k) mod 16, which hands over its halo slot iff it wrapped
                unsigned long long mask[7];
#pragma unroll
                for (int k = -3; k <= 3; ++k) mask[k + 3] = __builtin_amdgcn_ballot_w64(k > 0 ? i < k : i >= 16 + k);
                PROBE_SWITCH((sums_dpp<QI, MODE>(wr, arow, brow, pf, mask)))
            }
        }
        int wl = lane;
        asm volatile("" : "+v"(wl));
        *reinterpret_cast<f4a *>(partf + (it & 1) * (8 * 256) + wv * 256 + wl * 4) = f4a{pf[0], pf[1], pf[2], pf[3]};
        if (MODE == 0) __syncthreads();
        tot += pf[0] + pf[1] + pf[2] + pf[3];
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + tid] = tot;
}

template <int MODE>
void run(float *out, long long *ticks, int iters, float *host_out) {
    constexpr int XS_FLOATS = (MODE <= 1) ? G10::LH * G10::LW : LH2 * LW2;
    const int lds = 2 * XS_FLOATS * 4 + 2 * 8 * 256 * 4;
    (void)hipFuncSetAttribute((const void *)probe_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe_kernel<MODE>, dim3(256), dim3(512), lds, 0, out, iters, ticks);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(host_out, out, 512 * 4, hipMemcpyDeviceToHost);
}

}  // namespace
}  // namespace irn

int main() {
    using namespace irn;
    float *out;
    long long *ticks;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&ticks, 256 * 8);
    const int iters = 4000;
    const char *names[7] = {"shipped: 8x32 tile, LDS windows, barrier", "shipped, no barrier", "4x64 tile, LDS windows",
                            "4x64 tile, DPP sharing (A + halo B, select + ror)", "4x64 tile, DPP sharing, all rows read first",
                            "4x64 tile, own slot only (ceiling, wrong at edges)", "FMAs only"};
    static float res[7][512];
    for (int mode = 0; mode < 7; ++mode) {
        switch (mode) {
            case 0: run<0>(out, ticks, iters, res[0]); break;
            case 1: run<1>(out, ticks, iters, res[1]); break;
            case 2: run<2>(out, ticks, iters, res[2]); break;
            case 3: run<3>(out, ticks, iters, res[3]); break;
            case 4: run<4>(out, ticks, iters, res[4]); break;
            case 5: run<5>(out, ticks, iters, res[5]); break;
            default: run<6>(out, ticks, iters, res[6]); break;
        }
        long long h[256];
        (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += (double)h[i];
        mean /= 256;
        printf("mode %d  %-52s %.3f us per step\n", mode, names[mode], mean * 0.01 / iters);
        fflush(stdout);
    }
    int bad3 = 0, bad4 = 0, diff5 = 0;
    for (int i = 0; i < 512; ++i) {
        bad3 += res[3][i] != res[2][i];
        bad4 += res[4][i] != res[2][i];
        diff5 += res[5][i] != res[2][i];
    }
    printf("check: mode 3 vs 2: %d of 512 threads differ; mode 4 vs 2: %d; mode 5 vs 2: %d (edge lanes expected)\n", bad3, bad4, diff5);
    return bad3 || bad4;
}
