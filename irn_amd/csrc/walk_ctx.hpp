// Batch descriptors and the host-side context of the random walk, shared by the streaming sweeps
// (walk.hip) and the weights-stationary persistent kernel (walk_resident.hip).
#pragma once
#include <utility>
#include <vector>

#include "kernels.hpp"

namespace irn {

struct WalkImg {
    const float *edge;   // [h,w]
    const float *cam;    // [C/k_inst, h, w]
    const int *inst;     // [h,w] cluster map or null
    float *out;          // [C,h,w]
    float *wts;          // plane 0 / pixel 0 (front pad lies before it)
    double *inv_deg;     // [h*w]
    float *xa, *xb;      // [C, h*w] ping-pong state: fp32 (streaming sweeps) or 8-byte {tag,value} granules (resident)
    float *xc;           // [C, h*w] x 8 bytes, private to the pixel's owner: the polynomial schedule's carried terms —
                         // resident walk: {y_{t-1}, s_t} pairs; streaming sweeps: s_t as plain fp32 [C, h*w]
    int h, w, C, k_inst;
    long plane_stride;
    int front_pad, n_dirs;
};

// Pointers read out of a descriptor in memory are generic ("flat") to the compiler; the kernels
// want global_load / buffer_load, so say what they are.
#define IRN_GLOBAL __attribute__((address_space(1)))

template <int R>
__host__ __device__ constexpr bool in_set(int dy, int dx) {     // (dy,dx) in S, dy >= 0
    return dy == 0 ? (dx >= 1 && dx < R) : (dy < R && dx > -R && dx < R && dx * dx + dy * dy < R * R);
}

template <int R>
__host__ __device__ constexpr int plane_of(int dy, int dx) {    // raster index of (dy,dx) in S
    int n = 0;
    for (int y = 0; y < R; ++y)
        for (int x = -R + 1; x < R; ++x) {
            if (y == dy && x == dx) return n;
            if (in_set<R>(y, x)) ++n;
        }
    return -1;
}

// compile-time loop: f(integral_constant<int,0>) ... f(integral_constant<int,N-1>).  The neighbour
// loops MUST be expanded at compile time (plane numbers, window offsets and the disc test all fold
// to constants); `#pragma unroll` gives up on the nest of radius 10.
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

}  // namespace irn

struct irn_walk_ctx {
    using DeviceTable = irn::DeviceTable;
    using WalkImg = irn::WalkImg;
    using AffJob = irn::AffJob;
    int radius = 0;
    const DeviceTable *tab = nullptr;   // raster order
    int variant = 1;                    // 0 generic, 1 blocked streaming sweeps, 2 weights-stationary persistent walk
                                        // (irn_walk_create picks 2 for radius 5/10, 0 otherwise)
    // blocked sweep: all tiles of an image on one XCD, one tile shape (walk.hip kTile*), the channel-chunk classes of a sweep on
    // separate streams
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // batch
    int n = 0;
    std::vector<int> h, w, c;
    std::vector<size_t> off_wts, off_deg, off_xa, off_xb, off_xc;   // byte offsets into the workspace
    std::vector<long> plane_stride;
    std::vector<int> front_pad;
    size_t ws_bytes = 0;
    int max_h = 0, max_w = 0, max_n = 0;
    bool all_blocked_ok = false;
    // device-side descriptor storage (library-private)
    WalkImg *imgs_dev = nullptr;
    AffJob *jobs_dev = nullptr;
    int4 *map_dev = nullptr;
    int cap_imgs = 0, cap_map = 0;
    int cls_begin[5] = {0, 0, 0, 0, 0};    // block-map slice of channel-chunk width k: [cls_begin[k], +cls_count[k])
    int cls_count[5] = {0, 0, 0, 0, 0};
    int map_len = 0, max_nch = 1;
    // pinned staging for the per-run descriptors (2 slots, guarded by events)
    void *stage[2] = {nullptr, nullptr};
    size_t stage_cap = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_next = 0;
    // timing: one event pair per timed run since the last irn_walk_last_sweep_ms call
    int timing = 0;
    std::vector<hipEvent_t> ev_pool;       // all events ever created (reused)
    size_t ev_used = 0;                    // events handed out since the last read-out
    int pending_launches = 0;
    // weights-stationary persistent kernel (variant 2, walk_resident.hip)
    int4 *res_jobs_dev = nullptr;          // [res_rounds][res_nwg] (image, tile row0, tile col0, -)
    int res_cap_jobs = 0, res_rounds = 0, res_nwg = 0;
    int res_max_round_channels = 1;        // most channels of any image of the batch (bounds the waits of the launch)
    unsigned *res_err_dev = nullptr;       // [4] time-out diagnostics written by the kernel
    unsigned *res_err_host = nullptr;      // pinned mirror
    long long *res_prof_dev = nullptr;     // [2][256][4] time stamps (option "profile")
    int res_poll_delay = 10;   // s_sleep(1) units (64 clocks): first poll this long after our own stores
    int res_poll_auto = 1;     // 1: the first representative batch of the process probes 8 / 10 / 12 (walk_resident.hip); 0: pinned
    float res_poll_probe_ms[4] = {0.f, 0.f, 0.f, 0.f};   // launch times the probe measured for delays 8, 10, 12, 14 (0 = not probed by this context)
    int res_placement = 0;     // block -> XCD round robin: 0 not checked, 1 holds, 2 does not (resident_check_placement)
    bool res_plain_store = false;          // radius 5: plain state stores for images whose tiles share an XCD (voted in-kernel)
    int res_poll_delay_plain = 2;          // poll delay of such jobs
    unsigned long long *res_votes_dev = nullptr;   // [n] per-image XCD vote words
    int res_votes_cap = 0;
    bool deg_stale = false;    // last run was resident: the inv_deg array of the workspace was not written
    bool res_ok = false;                   // the configured batch fits the resident kernel
    int res_sweeps_per_launch = 0;         // 0 = all sweeps in one launch; k = relaunch every k sweeps (test hook)
    int res_cooperative = 1;               // launch with hipLaunchCooperativeKernel (co-residency of the grid is requested,
                                           // not assumed); falls back to a plain launch when the runtime refuses
    bool res_coop_refused = false;         // the runtime refused a cooperative launch once: plain launches from then on
    int res_inject_timeout = 0;            // test hook: the next resident launch reports a timeout without waiting
    // last irn_walk_run (for irn_walk_sync: wait, and re-run on the streaming sweeps if the persistent kernel gave up)
    hipStream_t last_stream = nullptr;
    hipEvent_t run_done_ev = nullptr;      // recorded behind every irn_walk_run
    bool last_valid = false, last_resident = false;
    int last_n_sweeps = 0;
    int fallback_runs = 0;                 // batches re-run on the streaming sweeps after a resident time-out
    // polynomial schedule of the walk (walk.hip: irn::walk_schedule)
    int accel = 1;                         // 1: T^n as a truncated Chebyshev series when that needs fewer operator applications
    int accel_tol_exp = 7;                 // truncation bound 10^-accel_tol_exp on the series' dropped coefficients: 84 applications for
                                           // n = 256.  6 (78 applications, +7 %) stays <= 3e-6 from the fp64 oracle but flipped one grid
                                           // argmax at a 1e-6 tie in the GPU suite (round 4, session 1) where 7 flips none: not the default
    int sched_n = -1, sched_accel = -1, sched_tol = -1;   // what coef_dev currently holds
    int sched_steps = 0;                   // operator applications of the schedule
    bool sched_cheb = false;               // three-term recurrence (else plain powers)
    float *coef_dev = nullptr;             // [sched_steps + 1] series coefficients c_0 .. c_steps
    std::vector<double> sched_coef;        // the same on the host
    std::vector<float> sched_coef_f;       // staging of the upload (must outlive the asynchronous copy)
    int coef_cap = 0;
};

namespace irn {
// walk_resident.hip: weights-stationary persistent walk (variant 2)
bool resident_supported(const irn_walk_ctx *ctx);
int resident_configure(irn_walk_ctx *ctx);
// the packing of a batch into rounds of the persistent launch, as plain host arithmetic (irn_walk_plan_rounds)
bool resident_plan_rounds(int radius, int n, const int *h, const int *w, const int *c, int n_wg, int placement,
                          std::vector<int> &jobs, int *n_rounds);
int resident_run(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream);
// walk.hip: the schedule that evaluates x . T^n_sweeps for this context's options (uploads the coefficient table on
// first use; stream-ordered) — ctx->sched_steps operator applications
int walk_schedule(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream);
// series coefficients of lambda^n in the Chebyshev basis, truncated where the dropped ones sum to <= tol; returns the
// number of recurrence steps K (coef gets K + 1 entries), or n itself with the unit "series" when that is not shorter
int chebyshev_power_series(int n, double tol, std::vector<double> *coef, bool *cheb);
// walk.hip: degree + x0 + n_sweeps streaming sweeps of the configured batch (weights already built)
int streaming_run(irn_walk_ctx *ctx, int n_sweeps, hipStream_t stream, bool timed);
void resident_destroy(irn_walk_ctx *ctx);
}  // namespace irn
