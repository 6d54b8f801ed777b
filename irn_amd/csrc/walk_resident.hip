// Weights-stationary persistent random walk (gfx950) — "variant 2" of irn_walk_run.
//
// The transition operator of an image is the same in every application of the walk
// (reference misc/indexing.py:136-137 squares ONE matrix).  The streaming sweeps of walk.hip re-read
// its |S| weight planes from HBM every sweep (4*|S|*N bytes: the HBM roofline of SURVEY.md §8d).
// This kernel reads them ONCE per image: every workgroup keeps the 2|S| directed weights of its
// pixels in VGPRs for the whole walk, so a sweep moves only the state (8*C*N bytes per image, all of
// it L2/fabric traffic between neighbouring workgroups) and the chip's 128 MB of register file is the
// cache the weights live in.
//
// Decomposition (radius 10: 304 directed neighbours per pixel)
//   * one workgroup = 512 threads = 8 waves, one per CU (256 VGPRs per lane), resident for the launch;
//   * a "slab" is 8 rows x 32 columns of pixels = 64 lanes x 4 consecutive pixels; Q waves share a
//     slab and split the neighbour disc between them in raster order (radius 10: Q = 8 waves x 38
//     neighbours x 4 px = 152 weight registers per lane; radius 5: Q = 2 x 34, four slabs per
//     workgroup).  Each wave's part of the disc is a different, fully expanded instruction stream
//     (wave-uniform switch), so every weight has a fixed register;
//   * a step = (operator application t, channel c), c fastest.  Waves 4-7 poll and stage y_t[c] of the tile + halo into
//     LDS (double-buffered); every wave runs ONE fp32 FMA chain per pixel over its part of the disc (38 terms at radius
//     10), reading the staged state through 16-byte-aligned register windows; the Q chains of a pixel meet in LDS; waves
//     0-3 add them pairwise in fp32, add the centre term and multiply by 1/deg in fp64, apply the recurrence of the
//     schedule (below) and store y_{t+1}[c] (numerics: tests/test_precision_model.py).  Channels are
//     independent chains: with C >= 2 the poll of the next step flies during the arithmetic of this one.
//   * the step loop (walk_resident_steps.inc) exists once per wave ROLE since round 4: the polling waves' copy never
//     stores, the combining waves' copy holds no poll registers; a wave branches into its copy once per job and the
//     barriers of the two copies match one for one (LESSONS.md 31).  Radius 5, single-channel jobs: the polling waves
//     take half of the combine.
//
// Exchange between workgroups (tiles of one image; no kernel boundary between sweeps)
//   state buffers hold one 8-byte granule {tag = sweep + 1, fp32 value} per pixel and channel, written
//   by ONE agent-scope (sc1, write-through) store and polled with agent-scope loads until the tag
//   matches: the data is its own flag (MI355X guide, Guideline 16 form R2), so a sweep costs one
//   store->load hop and no fence.  Ping-pong buffers are WAR-safe: a workgroup overwrites x_t of its
//   tile (while producing x_{t+2}) only after it has seen x_{t+1} of every tile within the halo, i.e.
//   after every reader of its x_t has finished reading.  Polls are bounded (wall clock): a
//   workgroup that cannot make progress reports through `err` and the launch fails loudly.
//
// Schedule (round 3, walk.hip header): a step is one application of the operator inside the three-term recurrence
//   y_{t+1} = 2 T y_t - y_{t-1},  s += c_{t+1} y_{t+1}   (first step y_1 = T y_0; plain powers when `cheb` is 0)
// — what is exchanged between tiles is y_t, exactly as before; y_{t-1} and s of a pixel are private to the thread that
// combines it and live in LDS ({prev, s} pairs, Geom::CAPC channels) or, for the channels beyond that, in the `xc`
// region of the workspace (plain loads and stores by the owner).  The last step hands out s.
//
// Scheduling: grid = one workgroup per CU.  The host packs images into rounds of <= #CU tiles in
// descending cost order (tile = 8x32 px at radius 10, 16x64 at radius 5); workgroup b runs
// job[round][b] for every round.
// Tiles of an image sit on consecutive slots of one XCD (slot -> block id b = idx*8 + xcd; observed
// placement, speed only).  No grid-wide barrier exists, so rounds pipeline.
#include <algorithm>
#include <mutex>

#include "walk_ctx.hpp"

namespace irn {
namespace {

typedef unsigned long long u64;
typedef const double IRN_GLOBAL *gcd_t;
typedef float IRN_GLOBAL *gf_t;
typedef float f4a __attribute__((ext_vector_type(4)));

#ifndef IRN_R10_COMBINER_UNSPLIT
#define IRN_R10_COMBINER_UNSPLIT 1 // radius 10, two channels: the combining waves (no poll to issue) run the arithmetic as ONE pipeline
#endif
#ifndef IRN_R5_SHARED_COMBINE
#define IRN_R5_SHARED_COMBINE 1    // radius 5, single-channel jobs: the polling waves take half of the combine (walk_resident_steps.inc)
#endif
#ifndef IRN_PROF_COMBINE
#define IRN_PROF_COMBINE 0     // diagnostic builds only (tools/combine_profile.py): 1 / 2 move the PROF stamps into the combine phase
#endif
#ifndef IRN_PS_LDS_BYTES
#define IRN_PS_LDS_BYTES (96 * 1024)       // LDS spent on the recurrence's private terms: 48 channels at radius 10, 12 at radius 5
#endif
constexpr int kSlabH = 8, kSlabW = 32;     // 64 lanes x 4 px
constexpr int kWaves = 8;                  // 512 threads

template <int R>
struct RCfg;
template <>
struct RCfg<10> {
    static constexpr int Q = 8, SL_Y = 1, SL_X = 1;
    static constexpr bool PREFETCH_ROWS = true;     // read the next neighbour row's window ahead of this row's FMAs
#ifndef IRN_R10_ROLE_SPLIT
#define IRN_R10_ROLE_SPLIT 1
#endif
    static constexpr bool ROLE_SPLIT = IRN_R10_ROLE_SPLIT != 0;     // one copy of the step loop per wave role (see the step loop)
};
template <>
struct RCfg<5> {
    static constexpr int Q = 2, SL_Y = 2, SL_X = 2;
    static constexpr bool PREFETCH_ROWS = true;     // fits since the fp32 chains freed the fp64 accumulators (250 VGPRs)
#ifndef IRN_R5_ROLE_SPLIT
#define IRN_R5_ROLE_SPLIT 1
#endif
    // split, this instantiation spills six loop-invariant registers that the polling waves reload once per step (their
    // staging addresses) — and is still 3.6 % faster (profiles/r04_s11_role_split_radius5_ab.txt)
    static constexpr bool ROLE_SPLIT = IRN_R5_ROLE_SPLIT != 0;
};

// The neighbour disc (dy,dx) != (0,0), dx^2 + dy^2 < R^2, in raster order: the union of the
// reference's half-plane direction set S (misc/indexing.py:18-56) and its mirror image.
template <int R>
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
template <int R>
inline constexpr Disc<R> kDisc{};

template <int R>
struct Geom {
    using C = RCfg<R>;
    static constexpr int H = R - 1;
    // LDS column of tile column 0.  Windows are read in 16-byte slots aligned in LDS, i.e. starting at offsets c with
    // (HP + c) % 4 == 0 from the lane's first pixel.  HP = H puts a slot boundary exactly at the leftmost neighbour of a full
    // disc row (dx = -H): such a row then needs 6 slots instead of 7 (radius 10: 9 of the 19 rows), which is 8 fewer live
    // VGPRs for the two windows in flight and 6 % fewer LDS reads.
    static constexpr int HP = H;
    static constexpr int TH = kSlabH * C::SL_Y, TW = kSlabW * C::SL_X;
    // LDS row stride = 32 banks (mod 64): the 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...)
    // span four tile rows; with this stride their 16-byte chunks fall on distinct banks
    // (56 floats, the tight fit, was 2-3-way conflicted: 0.95 us of a 3.1 us sweep).
    static constexpr int LH = TH + 2 * H, LW = 96;
    static_assert(LW >= TW + 2 * HP && LW % 64 == 32, "LDS row stride");
    static_assert((TW + 2 * H) % 2 == 0, "staged rows are polled in pixel pairs");
    static constexpr int NK = (LH * ((TW + 2 * H) / 2) + 255) / 256;   // staged pixel PAIRS per polling lane (waves 4-7)
    static constexpr int LWU = TW + 2 * H;               // columns actually staged
    static constexpr int RG = LH * LWU;                  // staged pixels per channel
    static constexpr int D = kDisc<R>.n;
    static constexpr int Q = C::Q;
    static constexpr int NS = D / Q;                     // neighbours per wave
    static constexpr int SLABS = C::SL_Y * C::SL_X;
    static_assert(D % Q == 0, "disc must split evenly over the waves of a slab");
    static_assert(SLABS * Q == kWaves, "8 waves per workgroup");
    // LDS carve (bytes)
    static constexpr int XS_BYTES = 2 * LH * LW * 4;               // [2][LH][LW] fp32, double-buffered per step
    static constexpr int PART_BYTES = 2 * kWaves * 4 * 64 * 8;      // [2][wave][lane][j]: fp64 for the degree (prologue), fp32 chains in the steps
    static constexpr int INVD_BYTES = SLABS * 4 * 64 * 8;           // [slab][row][column] fp64
    // {y_{t-1}, s_t} of the tile's own pixels for the first CAPC channels of a job (the rest go through the workspace)
    static constexpr int TPX = TH * TW;
    static constexpr int CAPC = IRN_PS_LDS_BYTES / (TPX * 8);
    static constexpr int PS_OFF = XS_BYTES + PART_BYTES + INVD_BYTES + 16;
    static constexpr int LDS_BYTES = PS_OFF + CAPC * TPX * 8;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget of one compute unit");
};

// first / last neighbour of row dy inside wave part QI (raster order => contiguous), or lo > hi
template <int R, int QI>
constexpr int row_lo(int dy) {
    constexpr int NS = Geom<R>::NS;
    for (int s = QI * NS; s < (QI + 1) * NS; ++s)
        if (kDisc<R>.dy[s] == dy) return s;
    return 1 << 20;
}
template <int R, int QI>
constexpr int row_hi(int dy) {
    constexpr int NS = Geom<R>::NS;
    for (int s = (QI + 1) * NS - 1; s >= QI * NS; --s)
        if (kDisc<R>.dy[s] == dy) return s;
    return -1;
}
// largest c <= v with (c + hp) % 4 == 0: the 16-byte-aligned slot boundary at or left of neighbour offset v
constexpr int slot_floor(int v, int hp) { return v - (((v + hp) % 4 + 4) % 4); }

// ---- per-job: weights of this lane's 4 pixels for the wave's part of the disc -> registers ----
template <int R, int QI>
__device__ __forceinline__ void load_weights(float (&wr)[Geom<R>::NS][4], const WalkImg &I, int gy, int gx) {
    using G = Geom<R>;
    const bool row_ok = gy < I.h && gx < I.w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(I.wts - I.front_pad), 0, (int)(I.n_dirs * I.plane_stride * 4), 0x00020000);
    // out-of-image lanes read beyond num_records: the buffer load returns 0.  The front pad goes into
    // the scalar offset so that it stays >= 0 for backward reads of plane 0 (the hardware adds it
    // as an unsigned 32-bit number).
    const int p4 = row_ok ? (gy * I.w + gx) * 4 : 0x7ffffff0;
    const int ps4 = (int)(I.plane_stride * 4);
    const int fp4 = I.front_pad * 4;
    static_for<G::NS>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = QI * G::NS + decltype(is)::value;
        constexpr int dy = kDisc<R>.dy[s], dx = kDisc<R>.dx[s];
        constexpr bool fwd = dy > 0 || (dy == 0 && dx > 0);
        // forward neighbour p+d uses w_d(p); backward neighbour p-d' (d' = -d in S) uses w_d'(p-d')
        constexpr int plane = fwd ? plane_of<R>(dy, dx) : plane_of<R>(-dy, -dx);
        const int soff = fp4 + plane * ps4 + (fwd ? 0 : (dy * I.w + dx) * 4);
        const f4a v = __builtin_bit_cast(f4a, __builtin_amdgcn_raw_buffer_load_b128(rsrc, p4, soff, 0));
        wr[decltype(is)::value][0] = v.x;
        wr[decltype(is)::value][1] = gx + 1 < I.w ? v.y : 0.f;
        wr[decltype(is)::value][2] = gx + 2 < I.w ? v.z : 0.f;
        wr[decltype(is)::value][3] = gx + 3 < I.w ? v.w : 0.f;
    });
}

// deg(p) - 1 = sum of the 2|S| directed weights of p (column sum of misc/indexing.py:135): this wave's part of it,
// from the registers load_weights just filled, in fp64.  A backward weight whose source pixel lies outside
// the image reads a stored zero by construction of the planes (walk.hip, layout); it is masked out here all
// the same, like degree_kernel does, so that the degree never depends on that property.
template <int R, int QI>
__device__ __forceinline__ void degree_partial(const float (&wr)[Geom<R>::NS][4], const WalkImg &I, int gy, int gx,
                                               double (&ds)[4]) {
    using G = Geom<R>;
    ds[0] = ds[1] = ds[2] = ds[3] = 0.0;
    // validity of a backward source as 0/1 FACTORS in VGPRs, one per column offset (dx + j) and per row: kept as
    // per-neighbour lane masks the conditions needed ~150 SGPR pairs at once and spilled
    float fx[2 * R + 2];
#pragma unroll
    for (int e = 0; e < 2 * R + 2; ++e) fx[e] = (unsigned)(gx + e - (R - 1)) < (unsigned)I.w ? 1.f : 0.f;
    static_for<G::NS>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = QI * G::NS + decltype(is)::value;
        constexpr int dy = kDisc<R>.dy[s], dx = kDisc<R>.dx[s];
        constexpr bool fwd = dy > 0 || (dy == 0 && dx > 0);
        if constexpr (fwd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ds[j] += (double)wr[decltype(is)::value][j];
        } else {
            const float fy = gy + dy >= 0 ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) ds[j] += (double)(wr[decltype(is)::value][j] * (fx[dx + j + R - 1] * fy));
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

// ---- per sweep and channel: this wave's partial sums for its 4 pixels ----
// Neighbour rows of wave part QI that belong to HALF (0: the rows up to and including the one that
// crosses the half-way point of the part, 1: the rest; radius 5 keeps everything in half 0 — the split
// pushed that instantiation into scratch; HALF 2 = all rows in one piece).  The C = 2 schedule issues a
// poll between the halves; the other schedules run the whole part as one pipeline.
template <int R, int QI, int HALF>
struct RowList {
    int n = 0;
    int dy[2 * R] = {};
    constexpr RowList() {
        constexpr int NS = Geom<R>::NS;
        for (int y = -(R - 1); y <= R - 1; ++y) {
            const int lo = row_lo<R, QI>(y), hi = row_hi<R, QI>(y);
            if (lo > hi) continue;
            const bool first_half = R == 5 || lo - QI * NS < NS / 2;
            if (HALF == 2 || first_half == (HALF == 0)) dy[n++] = y;
        }
    }
};
template <int R, int QI, int HALF>
inline constexpr RowList<R, QI, HALF> kRowList{};

template <int R, int QI, int DY>
struct RowInfo {
    static constexpr int lo = row_lo<R, QI>(DY), hi = row_hi<R, QI>(DY);
    static constexpr int c_lo = slot_floor(kDisc<R>.dx[lo], Geom<R>::HP), c_hi = kDisc<R>.dx[hi] + 3;
    static constexpr int N4 = (c_hi - c_lo) / 4 + 1;          // aligned 16-byte reads of the state window
};

constexpr int kMaxWin = 6;   // a full radius-10 row spans 22 floats = 6 aligned float4 slots (see Geom::HP)

// The state window of a neighbour row: floats c_lo .. c_lo + 4*N4 - 1 relative to the lane's first pixel (16-byte
// aligned), of which dx_lo .. dx_hi + 3 are used; N4 aligned 16-byte LDS reads.  Reading only the used floats of the two
// end slots (4- and 8-byte reads; a full row over-reads 6 of 28 floats) was measured and is SLOWER (radius 10: +1.2 %
// launch time, COCO shape +1.4 %): with 4 pixels per lane a 4-byte read of a wave touches every fourth bank only, the four
// tile rows of a 32-lane group collide 4-way (8 LDS cycles against 4 for the conflict-free 16-byte read), and the LDS row
// stride that makes the 16-byte reads conflict-free cannot do the same for narrower ones.
template <int R, int QI, int DY>
__device__ __forceinline__ void load_window(float (&w)[kMaxWin * 4], const float *xrow) {
    using RW = RowInfo<R, QI, DY>;
    static_assert(RW::N4 <= kMaxWin, "window");
    const float *row = xrow + DY * Geom<R>::LW + RW::c_lo;
    static_for<RW::N4>([&](auto ik) __attribute__((always_inline)) {
        constexpr int k = decltype(ik)::value;
        const f4a v = *reinterpret_cast<const f4a *>(row + 4 * k);
        w[4 * k] = v.x;
        w[4 * k + 1] = v.y;
        w[4 * k + 2] = v.z;
        w[4 * k + 3] = v.w;
    });
}

// ONE fp32 chain per pixel over all neighbours of the wave's part (<= 38 terms at radius 10, 34 at radius 5); the Q
// chains of a pixel are combined in fp64 together with the centre term and the normalisation.  A numpy model of this
// scheme (tests/test_precision_model.py) is as close to the exact operator as folding every neighbour row into fp64
// separately (1.5e-6 after 256 sweeps: the floor set by storing the state in fp32), and it saves 2 x 4 conversions /
// fp64 additions per row segment and wave.  (More chains for ILP and packed FMAs were both measured: slower / no gain.)
template <int R, int QI, int DY>
__device__ __forceinline__ void fma_window(const float (&wr)[Geom<R>::NS][4], const float (&w)[kMaxWin * 4], float (&pf)[4]) {
    using RW = RowInfo<R, QI, DY>;
    static_for<RW::hi - RW::lo + 1>([&](auto is) __attribute__((always_inline)) {
        constexpr int s = RW::lo + decltype(is)::value;
        constexpr int dx = kDisc<R>.dx[s];
        constexpr int k = s - QI * Geom<R>::NS;
        static_for<4>([&](auto ij) __attribute__((always_inline)) {
            constexpr int j = decltype(ij)::value;
            constexpr int e = dx + j - RW::c_lo;
            pf[j] = fmaf(wr[k][j], w[e], pf[j]);
        });
    });
}

// The window of row r+1 is read BEFORE the FMAs of row r (two windows live).  Isolated in
// tools/arith_probe.hip the phase is LDS-bound: the window reads alone take 0.44 us per step, reads
// then FMAs row after row 0.78 us (no overlap at all: every wave of the CU is in the same phase), this
// order 0.62 us; reading every window of the part first 0.70 us.
template <int R, int QI, int HALF>
__device__ __forceinline__ void partial_sums(const float (&wr)[Geom<R>::NS][4], const float *xrow, float (&pf)[4]) {
    constexpr int NR = kRowList<R, QI, HALF>.n;
    if constexpr (HALF != 1) pf[0] = pf[1] = pf[2] = pf[3] = 0.f;
    if constexpr (NR > 0 && !RCfg<R>::PREFETCH_ROWS) {
        // row after row
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            float w1[kMaxWin * 4];
            load_window<R, QI, kRowList<R, QI, HALF>.dy[decltype(ir)::value]>(w1, xrow);
            fma_window<R, QI, kRowList<R, QI, HALF>.dy[decltype(ir)::value]>(wr, w1, pf);
        });
    } else if constexpr (NR > 0) {
        float w[2][kMaxWin * 4];
        load_window<R, QI, kRowList<R, QI, HALF>.dy[0]>(w[0], xrow);
        static_for<NR>([&](auto ir) __attribute__((always_inline)) {
            constexpr int r = decltype(ir)::value;
            if constexpr (r + 1 < NR) load_window<R, QI, kRowList<R, QI, HALF>.dy[r + 1 < NR ? r + 1 : r]>(w[(r + 1) & 1], xrow);
            __builtin_amdgcn_sched_barrier(0);
            fma_window<R, QI, kRowList<R, QI, HALF>.dy[r]>(wr, w[r & 1], pf);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// Granule = {fp32 value, tag} in one naturally aligned 8-byte word, moved by ONE agent-scope access:
// buffer_load/store_dwordx2 with sc1 (aux 16) — the store writes through to memory, the load
// bypasses this CU's L1 (MI355X guide, Guideline 16 R1/R2).  Buffer addressing keeps the base in
// SGPRs and needs one 32-bit VGPR per item (64-bit flat addresses cost 2 and pushed the kernel
// into scratch).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16;
__device__ __forceinline__ u2v ld_granule(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, kSc1));
}
template <int AUX = kSc1>
__device__ __forceinline__ void st_granule(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, unsigned tag, float v) {
    u2v g;
    g.x = __float_as_uint(v);
    g.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(g, rsrc, voff, soff, AUX);
}

// two adjacent pixels' granules in ONE 16-byte write-through store (8-byte sc1 stores are one fabric write
// each and cost ~2.7x per byte: MI355X guide, stores table); each 8-byte half is still written whole
template <int AUX = kSc1>
__device__ __forceinline__ void st_granule2(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, unsigned tag, float v0, float v1) {
    u4v g;
    g.x = __float_as_uint(v0);
    g.y = tag;
    g.z = __float_as_uint(v1);
    g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, voff, soff, AUX);
}

// x_0 = cam * (1 - edge) (misc/indexing.py:162; instance split step/make_ins_seg_labels.py:77-80) as
// granules with tag 1 into xa; xb's tags are cleared so that no stale tag of an earlier run matches.
typedef float f2a __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void x0_granule_kernel(const WalkImg *__restrict__ imgs, int cheb, float c0) {
    const WalkImg I = imgs[blockIdx.y];
    const long n = (long)I.h * I.w;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float one_minus = 1.0f - I.edge[p];
    const int k = I.inst ? I.k_inst : 1;
    const int id = I.inst ? I.inst[p] : 0;
    u64 *xa = (u64 *)I.xa, *xb = (u64 *)I.xb;
    for (int c = 0; c < I.C; ++c) {
        const int cls = c / k, kk = c - cls * k;
        float v = I.cam[(long)cls * n + p];
        if (I.inst) v = v * (id == kk ? 1.0f : 0.0f);
        xa[(long)c * n + p] = ((u64)1 << 32) | (u64)__float_as_uint(v * one_minus);
        xb[(long)c * n + p] = 0;
        if (cheb) ((f2a *)I.xc)[(long)c * n + p] = f2a{0.f, c0 * (v * one_minus)};     // {y_{-1}, s_0 = c_0 y_0}
    }
}

// Which XCD does block b run on?  (start-up self-check of the slot -> block mapping, resident_check_placement)
__global__ __launch_bounds__(64) void xcc_probe_kernel(unsigned char *__restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned char)((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u);   // HW_REG_XCC_ID
}

// PROF: per-step time stamps for tools/resident_profile.py (its own instantiation: the stamp pointer would
// cost the production kernel two live VGPRs per lane, and it sits exactly at the 256-register limit)
template <int R, bool PROF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void resident_kernel(
    const WalkImg *__restrict__ imgs, const int4 *__restrict__ jobs, int n_rounds, int t_first, int t_count,
    int t_total, unsigned *err, long long timeout_ticks, long long *prof, int poll_delay, unsigned long long *votes,
    const float *__restrict__ coef, int cheb) {
    using G = Geom<R>;
    constexpr int H = G::H, HP = G::HP, LH = G::LH, LW = G::LW, LWU = G::LWU, RG = G::RG, Q = G::Q, NK = G::NK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    double *part = reinterpret_cast<double *>(smem + G::XS_BYTES);
    double *invd = reinterpret_cast<double *>(smem + G::XS_BYTES + G::PART_BYTES);
    int *abort_flag = reinterpret_cast<int *>(smem + G::XS_BYTES + G::PART_BYTES + G::INVD_BYTES);
    f2a *psl = reinterpret_cast<f2a *>(smem + G::PS_OFF);          // [CAPC][TPX] {y_{t-1}, s_t}, combine's pixel order

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab = wv / Q, qi = wv % Q;
    const int ly = (slab / G::C::SL_X) * kSlabH + (lane >> 3);
    const int lx = (slab % G::C::SL_X) * kSlabW + (lane & 7) * 4;
    if (tid == 0) *abort_flag = 0;
    // Option "plain_store" (radius 5 only; `votes` non-null): state stores without sc1 when every tile of the image
    // has been SEEN to run on this XCD.  Such a store keeps the line in the XCD's L2, which then serves a same-XCD
    // neighbour's sc1 poll without the trip through the fabric (C = 1 sweep 2.61 -> 2.23 us).  A reader on another XCD
    // never sees it in time (radius 10, 64 tiles = 2 XCDs per image: every launch ran into the bounded wait), and HIP
    // promises nothing about block -> XCD placement, so each job votes: every tile adds 1 to the byte of its XCC id
    // in the image's 64-bit word and waits until all tiles have voted; one byte holding them all = same XCD.
    int *plain_flag = abort_flag + 1;
    const int delay_plain = (poll_delay >> 16) & 0xffff;
    poll_delay &= 0xffff;


    float wr[G::NS][4];
    // poll_delay: units of s_sleep(1) = 64 clocks.  Fixed on purpose: steering the delay
    // from hits and misses was tried twice and lost both times — a miss usually means a neighbour
    // was late, not that this workgroup polled early, so every tile backs off together (symmetric
    // steering: 3.6 us per sweep, late-only steering: 3.5 us and drifting, fixed: 2.7-3.0 us).

#pragma unroll 1
    for (int round = 0; round < n_rounds; ++round) {
        const int4 je = jobs[round * gridDim.x + blockIdx.x];
        if (je.x < 0) continue;
        long long *jslot = nullptr;     // diagnostic: per-job prologue stamps of two workgroups, rounds 0-7 -> rows 248-255
        if (PROF && prof && round < 8 && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))
            jslot = prof + ((blockIdx.x == 0 ? 0 : 256) + 248 + round) * 4;
        if (PROF && jslot) jslot[0] = wall_clock64();
        const WalkImg I = imgs[je.x];
        const int ty0 = je.y, tx0 = je.z;
        const int h = I.h, w = I.w;
        const unsigned n = (unsigned)(h * w);
        const int gy = ty0 + ly, gx = tx0 + lx;
        if constexpr (R == 5) {
            if (votes && tid == 0) {
                const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;   // HW_REG_XCC_ID
                unsigned long long *v = votes + je.x;
                __hip_atomic_fetch_add(v, 1ull << (8 * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long t0 = wall_clock64();
                int same = 0;
                for (;;) {
                    const unsigned long long word = __hip_atomic_load(v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned total = 0;
                    bool single = false;
                    for (int f = 0; f < 8; ++f) {
                        const unsigned cnt = (unsigned)(word >> (8 * f)) & 0xffu;
                        total += cnt;
                        single |= cnt == (unsigned)je.w;
                    }
                    if (total >= (unsigned)je.w) {
                        same = single ? 1 : 0;
                        break;
                    }
                    if (wall_clock64() - t0 > timeout_ticks) break;      // sc1 stores are always safe
                    __builtin_amdgcn_s_sleep(16);
                }
                *plain_flag = same;
            }
        }

        double dsum[4];
#define IRN_LOAD_PART(QI)                          \
    load_weights<R, (QI) % Q>(wr, I, gy, gx);      \
    degree_partial<R, (QI) % Q>(wr, I, gy, gx, dsum)
        switch (qi) {
            case 0: IRN_LOAD_PART(0); break;
            case 1: IRN_LOAD_PART(1); break;
            case 2: IRN_LOAD_PART(2); break;
            case 3: IRN_LOAD_PART(3); break;
            case 4: IRN_LOAD_PART(4); break;
            case 5: IRN_LOAD_PART(5); break;
            case 6: IRN_LOAD_PART(6); break;
            default: IRN_LOAD_PART(7); break;
        }
#undef IRN_LOAD_PART
        // 1/deg of the tile: the waves' parts meet in LDS (the combine's partial-sum buffer), fp64 throughout
        __syncthreads();   // previous job's readers of part / invd / xs are done
        if (PROF && jslot) jslot[1] = wall_clock64();
        {
            double *pw = part + wv * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) pw[j] = dsum[j];
        }
        __syncthreads();
        const bool plain_st = R == 5 && votes && *plain_flag != 0;
        const int job_delay = plain_st ? delay_plain : poll_delay;
#pragma unroll
        for (int i = tid; i < G::SLABS * 256; i += 512) {     // [slab][row][column] like the combine's thread order
            const int s2 = i >> 8, prow = (i >> 5) & 7, x = i & 31;
            const int yy = ty0 + (s2 / G::C::SL_X) * kSlabH + prow;
            const int xx = tx0 + (s2 % G::C::SL_X) * kSlabW + x;
            const double *pr = part + (s2 * Q) * 256 + (prow * 8 + (x >> 2)) * 4 + (x & 3);
            double deg = 1.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) deg += pr[q * 256];
            invd[i] = (yy < h && xx < w) ? 1.0 / deg : 0.0;
        }
        // Wave roles: waves 4-7 poll and stage the state, waves 0-3 combine and store it.  A wave
        // that issued divergent stores after its prefetched poll loads can only wait for those loads
        // with vmcnt(0), i.e. for the write-through acknowledge of the stores as well (that put the
        // whole store latency, ~0.5 us, into every step); a polling wave never stores.
        const bool poller = wv >= 4;
        // Staged pixel PAIRS of a polling lane (the same for every channel and sweep of the job): two adjacent
        // granules come with ONE 16-byte load (half as many poll requests as one load per granule; each
        // 8-byte half is read whole).  Per pair: granule index of its first pixel inside one channel (high
        // bits) | LDS index (low 12 bits); vmask has one bit per HALF.  Positions outside the image are
        // zeroed once here and never examined (their half of the load reads a neighbouring granule or, out
        // of range, 0 from the bounds-checked buffer).
        unsigned btab[NK];
        unsigned vmask = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = (tid - 256) + k * 256;
            const int ry = i / (LWU / 2);
            int rx = (i - ry * (LWU / 2)) * 2;
            const int yy = ty0 - H + ry;
            int xx = tx0 - H + rx;
            bool second = true;
            if (xx == -1) {          // pair straddling the left image edge: load (0, 1) instead and use its first half only
                xx = 0;
                rx += 1;
                second = false;
            }
            const bool row_ok = poller && i < RG / 2 && yy >= 0 && yy < h;
            const bool ok0 = row_ok && xx >= 0 && xx < w, ok1 = row_ok && second && xx >= 0 && xx + 1 < w;
            // pairs without a valid half read granule 0 (in range, never examined)
            btab[k] = ((ok0 ? (unsigned)(yy * w + xx) : 0u) << 15) | (unsigned)(ry * LW + rx + (HP - H));
            if (ok0) vmask |= 1u << (2 * k);
            if (ok1) vmask |= 2u << (2 * k);
        }
        if (PROF && jslot) jslot[2] = wall_clock64();
        for (int i = tid; i < 2 * LH * LW; i += 512) xs[i] = 0.f;
        // the recurrence's private terms {y_{t-1}, s_t} of the tile's own pixels, first CAPC channels: from the workspace
        // (written by x0_granule_kernel, or by the previous launch of a walk cut into several) into LDS
        const __amdgpu_buffer_rsrc_t prs =
            __builtin_amdgcn_make_buffer_rsrc((void *)I.xc, 0, (int)(8u * n * (unsigned)I.C), 0x00020000);
        const int c_lds = cheb ? (I.C < G::CAPC ? I.C : G::CAPC) : 0;
        for (int idx = tid; idx < c_lds * G::TPX; idx += 512) {
            const int cc = idx / G::TPX, i = idx - cc * G::TPX;
            const int s2 = i >> 8, prow = (i >> 5) & 7, x = i & 31;
            const int yy = ty0 + (s2 / G::C::SL_X) * kSlabH + prow;
            const int xx = tx0 + (s2 % G::C::SL_X) * kSlabW + x;
            f2a v{0.f, 0.f};
            if (yy < h && xx < w)
                v = __builtin_bit_cast(f2a, __builtin_amdgcn_raw_buffer_load_b64(prs, (yy * w + xx) * 8, cc * (int)(8u * n), 0));
            psl[idx] = v;
        }
        __syncthreads();
        if (PROF && jslot) jslot[3] = wall_clock64();

        // ---- the walk of this tile: a pipeline of steps (sweep t, channel c), c fastest ----
        // Step k stages x_t[c] into xs[k & 1], forms the partial sums into part[k & 1], combines and
        // stores x_{t+1}[c].  Channels are independent chains, so with C >= 2 the poll of step k+1
        // is issued before the arithmetic of step k and its round trip through the fabric is hidden;
        // with C = 1 the next step's input does not exist before this step's stores have landed in
        // the neighbouring tiles, and the poll waits `poll_delay` behind our own stores instead.
        const int C = I.C;
        const int ch_bytes = (int)(8u * n);
        const int state_bytes = (int)(8u * n * (unsigned)C);
        const int n_steps = t_count * C;
        // delays are counted in s_sleep(1) = 64 clocks; reading a clock instead (s_memrealtime in
        // every wave, several times per step) cost more than 1 us per step
        auto nap = [&](int units) {
            for (int d = 0; d < units; ++d) __builtin_amdgcn_s_sleep(1);
        };
        auto state_rsrc = [&](int tt) {
            // + 16: the 16-byte poll of the last granule reaches one granule past the state (the workspace keeps
            // >= 64 bytes of slack behind every state buffer)
            return __builtin_amdgcn_make_buffer_rsrc((void *)((tt & 1) ? I.xb : I.xa), 0, state_bytes + 16, 0x00020000);
        };
        // One poll slot of NK granule pairs per polling lane.  Loads are unconditional (straight-line code).
        u4v va[NK];
        auto issue = [&](__amdgpu_buffer_rsrc_t rs, int soff) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
                va[kk] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((btab[kk] >> 15) << 3), soff, kSc1));
        };
        int t = t_first, c = 0;
        bool fresh = true;             // first step of the job: its input has been there since before the launch
        bool polled = false;           // the poll of the current step is already in flight
        // One copy of the step loop per wave ROLE (round 4): the polling waves' copy has no stores and the
        // combining waves' copy no poll registers, so no wait — and no register-allocation accident — of one role can land on
        // the other; at radius 10 each copy's arithmetic switch holds only its own four parts, so the code does not grow.  Measured
        // (profiles/r04_s10_role_split_ab.txt): default workload +4.2 %, COCO shape +6.1 %; steps 2.12 / 1.49 / 1.38 ->
        // 1.94 / 1.40 / 1.29 us at 1 / 2 / 3 channels; radius 5 (r04_s11): 66 398 -> 68 815 images/s, steps 1.88 / 1.54 -> 1.79 / 1.41 us.
        if constexpr (RCfg<R>::ROLE_SPLIT) {
            auto step_loop = [&](auto role_tag) __attribute__((always_inline)) -> bool {
                constexpr int ROLE = decltype(role_tag)::value;
                const bool poller_ = ROLE == 1;
#define IRN_STEP_ABORT return true
#include "walk_resident_steps.inc"
#undef IRN_STEP_ABORT
                return false;
            };
            if (poller) {
                if (step_loop(std::integral_constant<int, 1>{})) return;
            } else {
                if (step_loop(std::integral_constant<int, 2>{})) return;
            }
        } else {
            constexpr int ROLE = 0;
            const bool poller_ = poller;
#define IRN_STEP_ABORT return
#include "walk_resident_steps.inc"
#undef IRN_STEP_ABORT
        }
        // a walk cut into several launches (test hook): the LDS-held terms go back to the workspace for the next one
        if (c_lds > 0 && t_first + t_count < t_total) {
            __syncthreads();
            for (int idx = tid; idx < c_lds * G::TPX; idx += 512) {
                const int cc = idx / G::TPX, i = idx - cc * G::TPX;
                const int s2 = i >> 8, prow = (i >> 5) & 7, x = i & 31;
                const int yy = ty0 + (s2 / G::C::SL_X) * kSlabH + prow;
                const int xx = tx0 + (s2 % G::C::SL_X) * kSlabW + x;
                if (yy < h && xx < w)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, psl[idx]), prs, (yy * w + xx) * 8, cc * (int)(8u * n), 0);
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static void tile_shape(int radius, int *th, int *tw) {
    if (radius == 10) { *th = Geom<10>::TH; *tw = Geom<10>::TW; }
    else { *th = Geom<5>::TH; *tw = Geom<5>::TW; }
}

bool resident_supported(const irn_walk_ctx *ctx) { return ctx->radius == 5 || ctx->radius == 10; }

template <int R, bool PROF>
static int resident_capacity(int n_cu, int *capacity);

void resident_destroy(irn_walk_ctx *ctx) {
    if (ctx->res_jobs_dev) (void)hipFree(ctx->res_jobs_dev);
    if (ctx->res_err_dev) (void)hipFree(ctx->res_err_dev);
    if (ctx->res_err_host) (void)hipHostFree(ctx->res_err_host);
    if (ctx->res_prof_dev) (void)hipFree(ctx->res_prof_dev);
    if (ctx->res_votes_dev) (void)hipFree(ctx->res_votes_dev);
    ctx->res_votes_dev = nullptr;
    ctx->res_votes_cap = 0;
    ctx->res_prof_dev = nullptr;
    ctx->res_jobs_dev = nullptr;
    ctx->res_err_dev = nullptr;
    ctx->res_err_host = nullptr;
}

// The slot -> block mapping below puts the tiles of an image on consecutive slots of ONE XCD under the assumption that
// blocks b, b + 8, b + 16 ... of a launch run on one XCD (round-robin dispatch).  HIP promises nothing of the kind (it is
// what this driver does on an idle MI355X), so the assumption is CHECKED once per device and process: a 64-thread probe launch of the same grid size
// reads HW_REG_XCC_ID per block.  When it does not hold the mapping falls back to slot = block (still correct: the
// exchange is placement-independent, only the share of same-XCD hand-offs changes) and a line on stderr says so.
// g_placement[dev]: 0 = not checked, 1 = round robin holds, 2 = it does not.
static int g_placement[64] = {};
// both per-device caches (g_placement, g_poll_delay) are written once per process by whichever context gets there first:
// contexts configured from different host threads take turns (the second one finds the first one's answer)
static std::mutex g_tune_mu;

int resident_check_placement(irn_walk_ctx *ctx) {
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return IRN_OK;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_placement[dev] == 0) {
        const int n_wg = ctx->res_nwg;
        unsigned char *d = nullptr;
        std::vector<unsigned char> hst((size_t)n_wg, 0xff);
        IRN_HIP_TRY(hipMalloc((void **)&d, (size_t)n_wg));
        IRN_HIP_TRY(hipMemset(d, 0xff, (size_t)n_wg));
        int votes_ok = 0;
        for (int rep = 0; rep < 2; ++rep) {           // twice: the first launch of a context may be placed differently
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(n_wg), dim3(64), 0, nullptr, d);
            IRN_LAUNCH_CHECK("xcc_probe_kernel");
            IRN_HIP_TRY(hipMemcpy(hst.data(), d, (size_t)n_wg, hipMemcpyDeviceToHost));
            // what the slot mapping needs: blocks with equal b % 8 share an XCD and the eight residues sit on eight different
            // XCDs.  The dispatcher round-robins on from wherever the previous launch stopped, so block 0 may land on any XCD
            // (round 5: "first blocks on XCDs 7 0 1 2 3 4 5 6" on boxes whose earlier launches were not multiples of 8 blocks
            // — round 4's check for XCD == b % 8 exactly then switched the packing off for nothing)
            bool ok = n_wg % 8 == 0;
            unsigned seen = 0;
            for (int r = 0; ok && r < 8; ++r) {
                ok = hst[(size_t)r] < 8 && !(seen & (1u << hst[(size_t)r]));
                seen |= 1u << (hst[(size_t)r] & 7);
            }
            for (int b = 8; ok && b < n_wg; ++b) ok = hst[(size_t)b] == hst[(size_t)(b % 8)];
            votes_ok += ok ? 1 : 0;
        }
        (void)hipFree(d);
        g_placement[dev] = votes_ok == 2 ? 1 : 2;
        if (g_placement[dev] == 2)
            fprintf(stderr, "irn_hip: device %d: blocks with equal b %% 8 do not share an XCD here (first blocks on XCDs %d %d %d %d %d %d %d %d); "
                            "tiles keep their launch order (speed only)\n", dev, hst[0], hst[1], hst[2], hst[3], hst[4], hst[5], hst[6], hst[7]);
    }
    ctx->res_placement = g_placement[dev];
    return IRN_OK;
}

// Pack a batch into rounds of at most n_wg tiles (pure host arithmetic: no device, no context — also exported for the CPU
// tests as irn_walk_plan_rounds).  False when an image does not fit one round or is narrower than the radius.
static bool pack_rounds(int radius, const std::vector<int> &h, const std::vector<int> &w, const std::vector<int> &c, int n_wg,
                        int placement, std::vector<std::vector<int4>> &rounds) {
    int th, tw;
    tile_shape(radius, &th, &tw);
    const int n = (int)h.size();
    std::vector<int> tiles(n);
    for (int i = 0; i < n; ++i) {
        if (w[i] < radius) return false;
        tiles[i] = cdiv(h[i], th) * cdiv(w[i], tw);
        if (tiles[i] > n_wg) return false;
    }
    // First-fit over the images in descending cost order (channels = steps per sweep, then tiles);
    // slots of an image are consecutive.  Workgroups run their rounds back to back without a grid
    // barrier, so what matters is that the tile groups sharing a round carry similar work (equal
    // channel counts) and that the light images form the tail.  Results do not depend on the order.
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (c[a] != c[b]) return c[a] > c[b];
        return tiles[a] > tiles[b];
    });
    std::vector<std::vector<int>> round_imgs;
    std::vector<int> used;
    for (int oi = 0; oi < n; ++oi) {
        const int i = order[oi];
        size_t r = 0;
        for (; r < round_imgs.size(); ++r)
            if (used[r] + tiles[i] <= n_wg) break;
        if (r == round_imgs.size()) {
            round_imgs.emplace_back();
            used.push_back(0);
        }
        round_imgs[r].push_back(i);
        used[r] += tiles[i];
    }
    // Placement inside a round.  The tile groups of the same slot range run their rounds back to back, so a range that
    // was handed the heavier image of every mixed round (descending order: always the first one) finishes last — by up
    // to the cost difference between the heaviest and the lightest image of the batch.  Where the images of a round
    // have equal tile counts (the slot ranges are interchangeable) the heaviest image goes to the range that is free
    // first.  Cost model: steps x channels x the measured step period of that channel count (relative weights only;
    // profiles/r04_s15_resident_step_profile.txt: 1.94 / 1.37 / 1.26 us at radius 10, 1.68 / 1.46 / 1.47 at radius 5).
    auto cost_of = [&](int i) {
        const int ch = c[i];
        const double per = radius == 10 ? (ch == 1 ? 1.5 : ch == 2 ? 1.07 : 1.0) : (ch == 1 ? 1.15 : 1.0);
        return 7.0 + ch * per * 84.0;
    };
    std::vector<double> busy(n_wg, 0.0);                   // modelled finishing time of each slot so far
    rounds.clear();
    for (auto &imgs : round_imgs) {
        rounds.emplace_back(n_wg, make_int4(-1, 0, 0, 0));
        const int m = (int)imgs.size();
        bool uniform = true;
        for (int j = 1; j < m; ++j) uniform = uniform && tiles[imgs[j]] == tiles[imgs[0]];
        std::vector<int> first(m);                         // first slot of range j
        for (int j = 0, sl = 0; j < m; ++j) {
            first[j] = sl;
            sl += tiles[imgs[j]];
        }
        std::vector<int> range_of(m);                      // image j of the round (descending cost) -> range
        for (int j = 0; j < m; ++j) range_of[j] = j;
        if (uniform && m > 1) {
            std::vector<double> ready(m, 0.0);
            for (int j = 0; j < m; ++j)
                for (int q = 0; q < tiles[imgs[0]]; ++q) ready[j] = std::max(ready[j], busy[first[j] + q]);
            std::stable_sort(range_of.begin(), range_of.end(), [&](int a, int b) { return ready[a] < ready[b]; });
        }
        for (int j = 0; j < m; ++j) {
            const int i = imgs[j], rg = range_of[j];
            double start = 0.0;
            for (int q = 0; q < tiles[i]; ++q) start = std::max(start, busy[first[rg] + q]);
            for (int q = 0; q < tiles[i]; ++q) busy[first[rg] + q] = start + cost_of(i);
            int slot = first[rg];
            for (int ty = 0; ty < h[i]; ty += th)
                for (int tx = 0; tx < w[i]; tx += tw, ++slot) {
                    // consecutive slots share an XCD: block b is dispatched to XCD b % 8 (checked: resident_check_placement)
                    const int per = (n_wg + 7) / 8;
                    int b = (slot % per) * 8 + slot / per;
                    if (n_wg % 8 != 0 || b >= n_wg || placement == 2) b = slot;
                    rounds[rounds.size() - 1][b] = make_int4(i, ty, tx, tiles[i]);
                }
        }
    }
    return true;
}

bool resident_plan_rounds(int radius, int n, const int *h, const int *w, const int *c, int n_wg, int placement,
                          std::vector<int> &jobs, int *n_rounds) {
    std::vector<std::vector<int4>> rounds;
    if (!pack_rounds(radius, std::vector<int>(h, h + n), std::vector<int>(w, w + n), std::vector<int>(c, c + n), n_wg, placement,
                     rounds))
        return false;
    jobs.clear();
    for (auto &r : rounds)
        for (const int4 &j : r) {
            jobs.push_back(j.x);
            jobs.push_back(j.y);
            jobs.push_back(j.z);
            jobs.push_back(j.w);
        }
    *n_rounds = (int)rounds.size();
    return true;
}

// Pack the batch into rounds of at most n_wg tiles.  Sets ctx->res_ok = false (not an error) when an
// image does not fit one round or is narrower than the radius (then irn_walk_run falls back to the
// streaming sweeps).
int resident_configure(irn_walk_ctx *ctx) {
    ctx->res_ok = false;
    if (!resident_supported(ctx)) return IRN_OK;
    if (ctx->res_nwg == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        IRN_HIP_TRY(hipGetDevice(&dev));
        IRN_HIP_TRY(hipGetDeviceProperties(&prop, dev));
        ctx->res_nwg = prop.multiProcessorCount;
    }
    const int n_wg = ctx->res_nwg;
    if (ctx->res_placement == 0) {
        const int rc_p = resident_check_placement(ctx);
        if (rc_p) return rc_p;
    }
    std::vector<std::vector<int4>> rounds;
    if (!pack_rounds(ctx->radius, ctx->h, ctx->w, ctx->c, n_wg, ctx->res_placement, rounds)) return IRN_OK;
    const int n = (int)ctx->h.size();
    const int total = (int)rounds.size() * n_wg;
    if (total > ctx->res_cap_jobs) {
        if (ctx->res_jobs_dev) (void)hipFree(ctx->res_jobs_dev);
        ctx->res_jobs_dev = nullptr;
        IRN_HIP_TRY(hipMalloc((void **)&ctx->res_jobs_dev, sizeof(int4) * total));
        ctx->res_cap_jobs = total;
    }
    std::vector<int4> flat;
    flat.reserve(total);
    for (auto &r : rounds) flat.insert(flat.end(), r.begin(), r.end());
    IRN_HIP_TRY(hipMemcpy(ctx->res_jobs_dev, flat.data(), sizeof(int4) * total, hipMemcpyHostToDevice));
    if (ctx->res_votes_cap < n) {
        if (ctx->res_votes_dev) (void)hipFree(ctx->res_votes_dev);
        ctx->res_votes_dev = nullptr;
        IRN_HIP_TRY(hipMalloc((void **)&ctx->res_votes_dev, sizeof(unsigned long long) * n));
        ctx->res_votes_cap = n;
    }
    if (!ctx->res_err_dev) {
        IRN_HIP_TRY(hipMalloc((void **)&ctx->res_err_dev, 4 * sizeof(unsigned)));
        IRN_HIP_TRY(hipHostMalloc((void **)&ctx->res_err_host, 4 * sizeof(unsigned), hipHostMallocDefault));
    }
    ctx->res_rounds = (int)rounds.size();
    ctx->res_max_round_channels = 1;
    for (int i = 0; i < n; ++i) ctx->res_max_round_channels = std::max(ctx->res_max_round_channels, ctx->c[i]);
    // the grid must be co-resident: ask the runtime how many workgroups of this kernel fit (a partitioned device or a
    // different LDS / register budget changes the answer); otherwise the streaming sweeps take the batch
    int capacity = 0;
    const int rc_cap = ctx->radius == 10 ? resident_capacity<10, false>(n_wg, &capacity) : resident_capacity<5, false>(n_wg, &capacity);
    if (rc_cap) return rc_cap;
    ctx->res_ok = capacity >= n_wg;
    return IRN_OK;
}

// One workgroup per compute unit only works if every workgroup of the grid is resident at the same time (tiles wait
// for each other inside the launch).  How many fit is asked of the runtime, not assumed: 1 workgroup of 8 waves x 256
// VGPRs and ~100 KB of LDS per compute unit on an idle MI355X.
template <int R, bool PROF>
static int resident_capacity(int n_cu, int *capacity) {
    using G = Geom<R>;
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    static bool attr_set[64] = {};                  // the dynamic-LDS limit is a per-device function attribute
    if (dev < 0 || dev >= 64) return fail(IRN_ERR_STATE, "device ordinal %d out of range", dev);
    if (!attr_set[dev]) {
        IRN_HIP_TRY(hipFuncSetAttribute((const void *)resident_kernel<R, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        G::LDS_BYTES));
        attr_set[dev] = true;
    }
    int per_cu = 0;
    IRN_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)resident_kernel<R, PROF>, 512,
                                                             G::LDS_BYTES));
    *capacity = per_cu * n_cu;
    return IRN_OK;
}

template <int R, bool PROF>
static int launch_resident(irn_walk_ctx *ctx, int t_first, int t_count, int t_total, hipStream_t stream) {
    using G = Geom<R>;
    int capacity = 0;
    int rc = resident_capacity<R, PROF>(ctx->res_nwg, &capacity);
    if (rc) return rc;
    if (capacity < ctx->res_nwg)
        return fail(IRN_ERR_STATE, "resident walk: only %d of %d workgroups can be resident", capacity, ctx->res_nwg);
    // Bounded waits: a tile that cannot make progress reports instead of hanging.  The bound covers the slowest legal
    // wait — a workgroup that has moved on to its next round waits for neighbours still busy with a heavy image of the
    // previous one — so it grows with the work of the launch (3 us per channel-step is twice the measured rate).
    long long ticks = 200000000LL;                  // 2 s of the 100 MHz wall clock
    ticks += 4LL * 300LL * (long long)ctx->res_max_round_channels * (long long)t_count * (long long)std::max(ctx->res_rounds, 1);
    if (ctx->res_inject_timeout) {                  // test hook: every poll that misses once gives up
        ticks = -1;
        ctx->res_inject_timeout = 0;
    }
    unsigned long long *votes = nullptr;
    if (R == 5 && ctx->res_plain_store) {           // one vote word per image, cleared before every launch
        votes = ctx->res_votes_dev;
        IRN_HIP_TRY(hipMemsetAsync(votes, 0, sizeof(unsigned long long) * ctx->n, stream));
    }
    const WalkImg *imgs = ctx->imgs_dev;
    const int4 *jobs = ctx->res_jobs_dev;
    int n_rounds = ctx->res_rounds;
    unsigned *err = ctx->res_err_dev;
    long long *prof = ctx->res_prof_dev;
    int delays = ctx->res_poll_delay | (ctx->res_poll_delay_plain << 16);
    const float *coef = ctx->coef_dev;
    int cheb = ctx->sched_cheb ? 1 : 0;
    if (ctx->res_cooperative && !ctx->res_coop_refused) {
        void *args[] = {&imgs, &jobs, &n_rounds, &t_first, &t_count, &t_total, &err, &ticks, &prof, &delays, &votes, &coef, &cheb};
        const hipError_t e = hipLaunchCooperativeKernel((const void *)resident_kernel<R, PROF>, dim3(ctx->res_nwg), dim3(512),
                                                        args, G::LDS_BYTES, stream);
        if (e == hipSuccess) return IRN_OK;
        (void)hipGetLastError();                    // refused (e.g. a partition without cooperative queues): plain launch,
        ctx->res_coop_refused = true;               // the occupancy check above still holds
    }
    hipLaunchKernelGGL((resident_kernel<R, PROF>), dim3(ctx->res_nwg), dim3(512), G::LDS_BYTES, stream, imgs, jobs, n_rounds,
                       t_first, t_count, t_total, err, ticks, prof, delays, votes, coef, cheb);
    IRN_LAUNCH_CHECK("resident_kernel");
    return IRN_OK;
}

// x_0 and all operator applications of the configured batch, enqueued on `stream` (the schedule is already set).
static int resident_enqueue(irn_walk_ctx *ctx, hipStream_t stream) {
    const int n = ctx->n;
    const int n_sweeps = ctx->sched_steps;
    IRN_HIP_TRY(hipMemsetAsync(ctx->res_err_dev, 0, 4 * sizeof(unsigned), stream));
    hipLaunchKernelGGL(x0_granule_kernel, dim3(cdiv(ctx->max_n, 256), n), dim3(256), 0, stream, ctx->imgs_dev,
                       ctx->sched_cheb ? 1 : 0, ctx->sched_cheb ? (float)ctx->sched_coef[0] : 0.f);
    IRN_LAUNCH_CHECK("x0_granule_kernel");
    const int step = ctx->res_sweeps_per_launch > 0 ? ctx->res_sweeps_per_launch : n_sweeps;
    for (int t = 0; t < n_sweeps; t += step) {
        const int cnt = std::min(step, n_sweeps - t);
        int rc;
        if (ctx->res_prof_dev)
            rc = ctx->radius == 10 ? launch_resident<10, true>(ctx, t, cnt, n_sweeps, stream)
                                   : launch_resident<5, true>(ctx, t, cnt, n_sweeps, stream);
        else
            rc = ctx->radius == 10 ? launch_resident<10, false>(ctx, t, cnt, n_sweeps, stream)
                                   : launch_resident<5, false>(ctx, t, cnt, n_sweeps, stream);
        if (rc) return rc;
    }
    IRN_HIP_TRY(hipMemcpyAsync(ctx->res_err_host, ctx->res_err_dev, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    return IRN_OK;
}

// The poll delay of single-channel jobs (units of 64 clocks between a tile's own stores and its first poll of the
// neighbours') was tuned by hand on one box: 8 / 10 / 12 within 2 % of each other there, 20 before the stores were
// widened.  It is a property of the fabric's store -> load latency at this clock, so instead of trusting the constant the
// first representative batch of a process (radius 10, >= 4 rounds, at least a third of its images single-channel) is
// run with 8, 10, 12 and 14 — twice each, the walk writes the same bits whatever the delay — and the fastest wins if it
// beats the default (10) by more than 0.7 %; cached per device.  (Round 4: with the shared step loop 8 was 0.6-1.5 % ahead of 10 on
// three boxes; with one loop per wave role 10 and 12 are level and 8 is 3.5 % behind — the optimum moves with the code.)  Option "poll_delay" (or IRN_POLL_DELAY) pins the value and
// switches the probe off.  g_poll_delay[dev]: 0 = not probed yet.
static int g_poll_delay[64] = {};

static int resident_probe_poll_delay(irn_walk_ctx *ctx, hipStream_t stream) {
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || ctx->radius != 10 || !ctx->res_poll_auto) return IRN_OK;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_poll_delay[dev]) {
        ctx->res_poll_delay = g_poll_delay[dev];
        return IRN_OK;
    }
    if (ctx->res_prof_dev || ctx->res_sweeps_per_launch > 0 || ctx->res_inject_timeout || ctx->res_rounds < 4 ||
        ctx->sched_steps < 32)
        return IRN_OK;
    int single = 0;
    for (int i = 0; i < ctx->n; ++i) single += ctx->c[i] == 1 ? 1 : 0;
    if (3 * single < ctx->n) return IRN_OK;
    hipEvent_t e0, e1;
    IRN_HIP_TRY(hipEventCreate(&e0));
    IRN_HIP_TRY(hipEventCreate(&e1));
    constexpr int NC = 4;
    const int cand[NC] = {10, 8, 12, 14};           // the hand-tuned default first
    float best_ms[NC] = {1e30f, 1e30f, 1e30f, 1e30f};
    int rc = IRN_OK;
    for (int rep = 0; rep < 2 && !rc; ++rep)
        for (int k = 0; k < NC && !rc; ++k) {
            ctx->res_poll_delay = cand[k];
            (void)hipEventRecord(e0, stream);
            rc = resident_enqueue(ctx, stream);
            (void)hipEventRecord(e1, stream);
            if (rc) break;
            if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(IRN_ERR_HIP, "poll-delay probe: synchronisation failed"); break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ctx->res_err_host[0] != 0) ms = 1e30f;       // a launch that gave up says nothing about the delay
            best_ms[k] = std::min(best_ms[k], ms);
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    int pick = 0;
    for (int k = 1; k < NC; ++k)
        if (best_ms[k] < 0.993f * best_ms[0] && best_ms[k] < best_ms[pick]) pick = k;      // 0.7 %: twice the run-to-run spread of the minimum of two
    ctx->res_poll_delay = cand[pick];
    ctx->res_poll_probe_ms[0] = best_ms[1];      // 8, 10, 12, 14 in that order
    ctx->res_poll_probe_ms[1] = best_ms[0];
    ctx->res_poll_probe_ms[2] = best_ms[2];
    ctx->res_poll_probe_ms[3] = best_ms[3];
    if (!rc && best_ms[0] < 1e29f) g_poll_delay[dev] = cand[pick];
    return rc;
}

// x_0 and all sweeps of the configured batch.  The descriptors (imgs_dev) are already uploaded.
int resident_run(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream) {
    {
        const int rc_s = walk_schedule(ctx, n_sweeps, stream);     // operator applications of x . T^n_sweeps (walk.hip)
        if (rc_s) return rc_s;
    }
    {
        const int rc_p = resident_probe_poll_delay(ctx, stream);   // first representative batch of the process only
        if (rc_p) return rc_p;
    }
    return resident_enqueue(ctx, stream);
}

}  // namespace irn
