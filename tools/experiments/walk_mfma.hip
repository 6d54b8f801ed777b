// EXPERIMENT (round 2, sessions 8-9) — NOT part of libirn_hip.so.  Measured slower than the VALU kernel (DESIGN.md §4 lesson 20,
// profiles/r02_s8_mfma_experiment.txt) and taken out of the build; kept for the record.  It was wired in by commit 580886e
// (host hooks mfma_supported / mfma_capacity / mfma_launch in walk_resident.hip, option "mfma_min_c"): check that commit out to
// build and run it.
//
// Weights-stationary persistent random walk for MANY-CHANNEL images on the fp32 matrix pipe (gfx950, radius 10).
//
// Same operator, data layout in HBM, tile decomposition (8 x 32 pixels per workgroup) and tile-to-tile exchange (tagged
// 8-byte granules, walk_resident.hip) as the VALU kernel — what changes is the arithmetic.  The per-pixel weights rule
// out a GEMM, but the contribution of ONE source pixel q to FOUR destination pixels p_0..p_3 for FOUR channels is an
// outer product,
//
//     out[p_i, c_j] += W(p_i, q) * x[q, c_j],
//
// which is one block of v_mfma_f32_4x4x1_16B_f32: 16 such blocks per instruction, f32 in / f32 accumulate (bitwise an
// fmaf chain), 512 flops per 8 cycles = the same 64 flop/clk/SIMD as v_fma_f32 — but every weight register feeds four
// channels per issue instead of one, and the matrix pipe reaches its peak where the VALU path of walk_resident.hip is
// held at a third of it by LDS latency and issue overheads (DESIGN.md §4 lesson 18).  It pays from ~3 channels per weight
// use up; irn_walk_run sends images with at least `mfma_min_c` channels here (default 12: COCO-shape class counts, heavy
// instance splits), everything else to the VALU kernel.
//
// Mapping (one workgroup = 4 waves = one 8 x 32 tile, ONE wave per SIMD with the whole 512-entry register file:
// wave w = tile rows 2w and 2w + 1)
//   * a "group" = 4 consecutive pixels of a row; the wave's 16 groups (2 rows x 8) are the 16 blocks of the MFMA;
//   * source list of a group: every (sy, sx) relative to the group's first pixel that is a neighbour (or the centre) of
//     at least one of its 4 pixels — the disc rows widened by 3: NE = 362 entries.  Register A[r] of a lane (block b,
//     row m) holds W(pixel m of group b, source r), 0 where that pair is not a neighbour pair (or leaves the image), 1
//     for the centre: 362 registers per lane (MFMA operands may live in AGPRs), loaded ONCE per image — the same
//     77 824 + padding weights per tile as the VALU kernel holds, in half as many waves;
//   * B operand of MFMA r: lane (b, j) reads x[source r of group b][channel 4g + j] from LDS — planar staging buffers
//     whose plane stride is 1 (mod 32) floats, so the 32 lanes of a half-wave hit 32 different banks;
//   * four accumulator sets (r mod 4) keep the matrix pipe issuing back to back from a single wave and shorten the fp32
//     chains to ~90 terms;
//   * no cross-wave combine: a wave owns its 64 pixels completely.  D -> x 1/deg in fp64 -> tagged granules, 16 bytes
//     per store.
// Steps are GROUPS of 4 channels of one sweep.  The polls of group g + 2 are issued behind the stores of group g and
// consumed after the matrix phase of group g + 1 (every lane polls: 3 granule pairs per channel and lane), so an image
// needs >= 3 groups (12 channels) for the hand-off latency to stay hidden.
#include <algorithm>

#include "walk_ctx.hpp"

namespace irn {
namespace {

typedef float f4a __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float IRN_GLOBAL *gf_t;
constexpr int kSc1 = 16;

constexpr int R = 10, H = R - 1;
constexpr int TH = 8, TW = 32;
constexpr int LH = TH + 2 * H;                 // 26 staged rows
constexpr int LWU = TW + 2 * H;                // 50 staged columns
constexpr int LWM = 52;                        // row stride of a staged plane (floats)
constexpr int PLANE = LH * LWM + 25;           // 1377 = 1 (mod 32): lanes (block b, channel j) -> bank 4b + j + const
static_assert(PLANE % 32 == 1, "plane stride must spread the 4 channels of a block over adjacent banks");
constexpr int NPLANES = 8;                     // two groups of 4 channels: the one being multiplied, the one being staged
constexpr int NK = 3;                          // granule pairs per lane and channel (650 pairs over 256 lanes)
constexpr int RGP = LH * (LWU / 2);            // staged pixel pairs per channel

// ---- the source list of a pixel group: disc rows widened by 3 columns (relative to the group's first pixel) ----
struct SrcList {
    int n = 0;
    signed char sy[400] = {}, sx[400] = {};
    constexpr SrcList() {
        for (int y = -H; y <= H; ++y) {
            int wmax = -1;
            for (int x = 0; x <= H; ++x)
                if (x * x + y * y < R * R) wmax = x;
            for (int x = -wmax; x <= wmax + 3; ++x) {
                sy[n] = (signed char)y;
                sx[n] = (signed char)x;
                ++n;
            }
        }
    }
};
inline constexpr SrcList kSrc{};
constexpr int NE = kSrc.n;                     // A registers per lane
static_assert(NE == 362, "source list of radius 10");
constexpr int kThreads = 256;

// plane index of direction (dy, dx) of the half-plane set S in raster order, or -1 (host + device)
struct PlaneTab {
    short v[(2 * R - 1) * (2 * R - 1)] = {};
    constexpr PlaneTab() {
        for (int dy = -H; dy <= H; ++dy)
            for (int dx = -H; dx <= H; ++dx) {
                short p = -1;
                const int ady = dy < 0 || (dy == 0 && dx < 0) ? -dy : dy, adx = dy < 0 || (dy == 0 && dx < 0) ? -dx : dx;
                if ((dy != 0 || dx != 0) && dx * dx + dy * dy < R * R) p = (short)plane_of<R>(ady, adx);
                v[(dy + H) * (2 * R - 1) + dx + H] = p;
            }
    }
};
__device__ const PlaneTab kPlaneTab{};

__device__ __forceinline__ void st_granule(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, unsigned tag, float v) {
    u2v g;
    g.x = __float_as_uint(v);
    g.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(g, rsrc, voff, soff, kSc1);
}
__device__ __forceinline__ void st_granule2(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, unsigned tag, float v0, float v1) {
    u4v g;
    g.x = __float_as_uint(v0);
    g.y = tag;
    g.z = __float_as_uint(v1);
    g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, voff, soff, kSc1);
}

constexpr int LDS_BYTES = NPLANES * PLANE * 4 + 64;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void resident_mfma_kernel(
    const WalkImg *__restrict__ imgs, const int4 *__restrict__ jobs, int n_rounds, int t_first, int t_count, int t_total,
    unsigned *err, long long timeout_ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    int *abort_flag = reinterpret_cast<int *>(smem + NPLANES * PLANE * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave owns tile rows 2 wv and 2 wv + 1
    const int blk = lane >> 2, sub = lane & 3;                    // MFMA block, row (A / D) or column (B) inside it
    const int grp = blk & 7, trow = 2 * wv + (blk >> 3);          // the block's pixel group: tile row, 4-pixel group of it
    if (tid == 0) *abort_flag = 0;

#pragma unroll 1
    for (int round = 0; round < n_rounds; ++round) {
        const int4 je = jobs[round * gridDim.x + blockIdx.x];
        if (je.x < 0) continue;
        const WalkImg I = imgs[je.x];
        const int ty0 = je.y, tx0 = je.z;
        const int h = I.h, w = I.w;
        const unsigned n = (unsigned)(h * w);
        const int C = I.C;
        __syncthreads();                                          // previous job's readers of xs are done

        // ---- weights of this lane: A[r] = W(pixel `sub` of its block's group, source r of the group's list) ----
        float A[NE];
        double deg = 0.0;
        {
            const int gy = ty0 + trow, gx = tx0 + 4 * grp + sub;
            const bool dst_ok = gy < h && gx < w;
            const float *wts = I.wts;
            const long ps = I.plane_stride;
            // (opaque copy: everything below that depends on the lane only — disc tests, plane numbers — would otherwise
            // be hoisted out of the job loop: hundreds of live registers)
            int sb = sub;
            asm volatile("" : "+v"(sb));
            static_for<NE>([&](auto ir) __attribute__((always_inline)) {
                constexpr int r = decltype(ir)::value;
                constexpr int dy = kSrc.sy[r];
                const int dx = kSrc.sx[r] - sb;                   // offset from this lane's pixel to the source
                const int qy = gy + dy, qx = gx + dx;
                float wgt = 0.f;
                const bool in_disc = dx >= -H && dx <= H && dx * dx + dy * dy < R * R;
                if (dst_ok && in_disc && qy >= 0 && qy < h && qx >= 0 && qx < w) {
                    if (dy == 0 && dx == 0) {
                        wgt = 1.f;                                // centre term of the sweep (unit diagonal)
                    } else {
                        const int pl = kPlaneTab.v[(dy + H) * (2 * R - 1) + dx + H];
                        const bool fwd = dy > 0 || (dy == 0 && dx > 0);
                        // forward pair {p, p+d} is stored at p in plane d; backward pair {p-d', p} at p-d' = the source
                        const long pix = fwd ? (long)gy * w + gx : (long)qy * w + qx;
                        wgt = wts[(long)pl * ps + pix];
                    }
                }
                A[r] = wgt;
                deg += (double)wgt;
                // keep the 362 gathers from being scheduled all at once (their address registers would not fit)
                if constexpr (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            });
        }
        // 1/deg of the 4 pixels whose results end up in this lane's D registers (register i <-> pixel i of the group):
        // pixel i's sum sits in lane 4 * blk + i
        double invd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double d = __shfl(deg, (lane & ~3) | i);
            const int px = tx0 + 4 * grp + i;
            invd[i] = (ty0 + trow < h && px < w && d > 0.0) ? 1.0 / d : 0.0;
        }

        // ---- staging tables: this lane's granule pairs of a staged plane (the same for every channel and sweep) ----
        unsigned btab[NK];
        unsigned vmask = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = tid + k * kThreads;
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
            const bool row_ok = i < RGP && yy >= 0 && yy < h;
            const bool ok0 = row_ok && xx >= 0 && xx < w, ok1 = row_ok && second && xx >= 0 && xx + 1 < w;
            btab[k] = ((ok0 ? (unsigned)(yy * w + xx) : 0u) << 15) | (unsigned)(ry * LWM + rx);
            if (ok0) vmask |= 1u << (2 * k);
            if (ok1) vmask |= 2u << (2 * k);
        }
        for (int i = tid; i < NPLANES * PLANE; i += kThreads) xs[i] = 0.f;
        __syncthreads();

        const int ch_bytes = (int)(8u * n);
        const int state_bytes = (int)(8u * n * (unsigned)C);
        auto state_rsrc = [&](int tt) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)((tt & 1) ? I.xb : I.xa), 0, state_bytes + 16, 0x00020000);
        };
        const int G = (C + 3) / 4;                                // channel groups per sweep
        const int n_groups = t_count * G;

        // poll slots: TWO channels x NK pairs (a group's four channels are polled in two halves: registers)
        u4v va[2][NK];
        auto issue_pair = [&](int tt, int gg, int j0) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs = state_rsrc(tt);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = min(4 * gg + j0 + j, C - 1);       // a partial last group polls its last channel again (ignored)
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
                    va[j][kk] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((btab[kk] >> 15) << 3), ch * ch_bytes, kSc1));
            }
        };
        // consume the slots into planes pbase + j0, pbase + j0 + 1 (re-polling until every tag matches, bounded)
        auto consume_pair = [&](int tt, int gg, int j0, int pbase) __attribute__((always_inline)) {
            const unsigned want = (unsigned)(tt + 1);
            unsigned pend[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) pend[j] = (4 * gg + j0 + j < C) ? vmask : 0u;
            long long t_start = 0;
            for (;;) {
                unsigned any = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float *plane = xs + (pbase + j0 + j) * PLANE;
#pragma unroll
                    for (int kk = 0; kk < NK; ++kk) {
                        if (((pend[j] >> (2 * kk)) & 1u) && va[j][kk].y == want) {
                            plane[btab[kk] & 0x7fff] = __uint_as_float(va[j][kk].x);
                            pend[j] &= ~(1u << (2 * kk));
                        }
                        if (((pend[j] >> (2 * kk)) & 2u) && va[j][kk].w == want) {
                            plane[(btab[kk] & 0x7fff) + 1] = __uint_as_float(va[j][kk].z);
                            pend[j] &= ~(2u << (2 * kk));
                        }
                    }
                    any |= pend[j];
                }
                if (!__builtin_amdgcn_ballot_w64(any != 0)) break;
                issue_pair(tt, gg, j0);
                const long long now = wall_clock64();
                if (t_start == 0) t_start = now;
                else if (now - t_start > timeout_ticks || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0 && atomicCAS(err, 0u, 1u) == 0u) {
                        err[1] = (unsigned)je.x;
                        err[2] = (unsigned)tt;
                        err[3] = blockIdx.x;
                    }
                    *abort_flag = 1;
                    break;
                }
            }
        };

        // LDS address (floats) of this lane's B operand for source (sy, sx): plane `sub`, row trow + H + sy, column
        // H + 4 * grp + sx
        const int bbase = sub * PLANE + (trow + H) * LWM + H + 4 * grp;

        // group 0: its input has been there since before the launch (or since the previous launch of a chunked run)
        int t = t_first, gg = 0;
        issue_pair(t, gg, 0);
        consume_pair(t, gg, 0, 0);
        issue_pair(t, gg, 2);
        consume_pair(t, gg, 2, 0);
        __syncthreads();
        if (*abort_flag) return;
        {
            int tn = t, gn = gg + 1;
            if (gn == G) { gn = 0; ++tn; }
            if (1 < n_groups) issue_pair(tn, gn, 0);
        }

#pragma unroll 1
        for (int g = 0; g < n_groups; ++g) {
            const int pbase = (g & 1) * 4;
            if (timeout_ticks < 0 && g > 0) {                   // test hook (option inject_timeout)
                if (tid == 0 && atomicCAS(err, 0u, 1u) == 0u) {
                    err[1] = (unsigned)je.x;
                    err[2] = (unsigned)t;
                    err[3] = blockIdx.x;
                }
                return;
            }
            // next group (its first two channels have been polled since the end of the previous iteration)
            int tn = t, gn = gg + 1;
            if (gn == G) { gn = 0; ++tn; }
            const bool more = g + 1 < n_groups;
            // ---- matrix phase: 362 outer-product steps, four accumulator sets; the next group is staged under it ----
            f4a acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            {
                const float *bp = xs + pbase * PLANE + bbase;
                // B operands through an explicit ring of kAhead registers, loaded kAhead steps before their MFMA: a single
                // wave per SIMD has nobody else to hide the LDS latency behind (left to itself the compiler waits for
                // every operand right before its use: 10 us per group instead of ~2)
                constexpr int kAhead = 8;
                float bq[kAhead];
                static_for<kAhead>([&](auto ir) __attribute__((always_inline)) {
                    constexpr int r = decltype(ir)::value;
                    bq[r] = bp[kSrc.sy[r] * LWM + kSrc.sx[r]];
                });
                static_for<NE>([&](auto ir) __attribute__((always_inline)) {
                    constexpr int r = decltype(ir)::value;
                    acc[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[r], bq[r % kAhead], acc[r & 3], 0, 0, 0);
                    if constexpr (r + kAhead < NE) {
                        constexpr int rn = r + kAhead;
                        bq[r % kAhead] = bp[kSrc.sy[rn] * LWM + kSrc.sx[rn]];
                    }
                    __builtin_amdgcn_sched_barrier(0);          // keep this order: MFMA r, then the load for r + kAhead
                    if constexpr (r == NE / 2) {
                        // half-way: channels 0,1 of the next group have landed; stage them and send the polls of 2,3,
                        // which fly under the second half
                        if (more) {
                            consume_pair(tn, gn, 0, 4 - pbase);
                            issue_pair(tn, gn, 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
            if (more) consume_pair(tn, gn, 2, 4 - pbase);

            // ---- epilogue: normalisation, store ----
            {
                const f4a d = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                float res[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) res[i] = (float)((double)d[i] * invd[i]);
                const int ch = 4 * gg + sub;                      // D column = channel
                const bool last = (t + 1 == t_total);
                const int yy = ty0 + trow;
#pragma unroll
                for (int i0 = 0; i0 < 4; i0 += 2) {              // pixels (0,1) and (2,3) of the group: one 16-byte store each
                    const int xx = tx0 + 4 * grp + i0;
                    const float r0 = res[i0], r1 = res[i0 + 1];
                    if (ch < C && yy < h && xx < w) {
                        const unsigned o = (unsigned)(yy * w + xx);
                        if (last) {
                            ((gf_t)I.out)[(unsigned)ch * n + o] = r0;
                            if (xx + 1 < w) ((gf_t)I.out)[(unsigned)ch * n + o + 1] = r1;
                        } else {
                            const __amdgpu_buffer_rsrc_t dst = state_rsrc(t + 1);
                            if (xx + 1 < w) st_granule2(dst, (int)o * 8, ch * ch_bytes, (unsigned)(t + 2), r0, r1);
                            else st_granule(dst, (int)o * 8, ch * ch_bytes, (unsigned)(t + 2), r0);
                        }
                    }
                }
            }
            __syncthreads();                                      // next group's planes complete; this group's planes free
            if (*abort_flag) return;
            // ---- polls of the group after next: behind our stores, consumed after the next matrix phase ----
            t = tn;
            gg = gn;
            if (g + 2 < n_groups) {
                int t2 = t, g2 = gg + 1;
                if (g2 == G) { g2 = 0; ++t2; }
                issue_pair(t2, g2, 0);
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool mfma_supported(const irn_walk_ctx *ctx) { return ctx->radius == 10; }

int mfma_capacity(int n_cu, int *capacity) {
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev < 0 || dev >= 64) return fail(IRN_ERR_STATE, "device ordinal %d out of range", dev);
    if (!attr_set[dev]) {
        IRN_HIP_TRY(hipFuncSetAttribute((const void *)resident_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    int per_cu = 0;
    IRN_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)resident_mfma_kernel, kThreads, LDS_BYTES));
    *capacity = per_cu * n_cu;
    return IRN_OK;
}

int mfma_launch(irn_walk_ctx *ctx, int t_first, int t_count, int t_total, long long ticks, hipStream_t stream) {
    int capacity = 0;
    int rc = mfma_capacity(ctx->res_nwg, &capacity);
    if (rc) return rc;
    if (capacity < ctx->res_nwg)
        return fail(IRN_ERR_STATE, "resident walk (matrix form): only %d of %d workgroups can be resident", capacity, ctx->res_nwg);
    const WalkImg *imgs = ctx->imgs_dev;
    const int4 *jobs = ctx->mfma_jobs_dev;
    int n_rounds = ctx->mfma_rounds;
    unsigned *err = ctx->res_err_dev;
    if (ctx->res_cooperative && !ctx->res_coop_refused) {
        void *args[] = {&imgs, &jobs, &n_rounds, &t_first, &t_count, &t_total, &err, &ticks};
        const hipError_t e = hipLaunchCooperativeKernel((const void *)resident_mfma_kernel, dim3(ctx->res_nwg), dim3(kThreads), args,
                                                        LDS_BYTES, stream);
        if (e == hipSuccess) return IRN_OK;
        (void)hipGetLastError();
        ctx->res_coop_refused = true;
    }
    hipLaunchKernelGGL(resident_mfma_kernel, dim3(ctx->res_nwg), dim3(kThreads), LDS_BYTES, stream, imgs, jobs, n_rounds, t_first,
                       t_count, t_total, err, ticks);
    IRN_LAUNCH_CHECK("resident_mfma_kernel");
    return IRN_OK;
}

}  // namespace irn
