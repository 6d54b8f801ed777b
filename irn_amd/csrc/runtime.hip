// Library-private device state: path tables cached per (radius, order, device) and a small
// stream-ordered upload scratch for descriptor arrays.
#include <map>
#include <mutex>
#include <tuple>

#include "kernels.hpp"

namespace irn {

namespace {
std::mutex g_mu;
std::map<std::tuple<int, int, int>, DeviceTable *> g_tables;

template <typename T>
int upload(const std::vector<T> &v, T **dev) {
    IRN_HIP_TRY(hipMalloc((void **)dev, sizeof(T) * v.size()));
    IRN_HIP_TRY(hipMemcpy(*dev, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return IRN_OK;
}

// Descriptor scratch of the batched entry points: a small ring of slots, each with its own completion event, so that
// back-to-back calls do not wait for each other's kernels (one slot made every call block the host until the previous
// call's kernels — e.g. the 300-iteration centroid kernel — had finished)
struct ScratchSlot {
    void *dev = nullptr;
    void *host = nullptr;    // pinned
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int device = -1;
};
constexpr int kScratchSlots = 4;
struct Scratch {
    ScratchSlot slot[kScratchSlots];
    int next = 0, last = 0;
};
thread_local Scratch t_scratch;
}  // namespace

int get_device_table(int radius, int order, const DeviceTable **out) {
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(radius, order, dev);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) {
        *out = it->second;
        return IRN_OK;
    }
    DeviceTable *t = new DeviceTable();
    t->radius = radius;
    t->order = order;
    t->host = build_path_table(radius, order);
    t->n_dirs = t->host.n_dirs();
    t->n_cells = t->host.n_cells();
    int rc = upload(t->host.dy, &t->dir_dy);
    if (!rc) rc = upload(t->host.dx, &t->dir_dx);
    if (!rc) rc = upload(t->host.start, &t->dir_start);
    if (!rc) rc = upload(t->host.cy, &t->cell_dy);
    if (!rc) rc = upload(t->host.cx, &t->cell_dx);
    // Affinity kernel: LDS offsets of the path cells inside its edge tile (row length depends on the
    // radius only), every path padded to a multiple of 8 by repeating its last cell (max is
    // idempotent) so that the kernel fetches eight offsets per scalar load.
    {
        const int lw = kAffTileW + 2 * (radius - 1);
        std::vector<int> off8, start8(1, 0);
        for (int d = 0; d < t->n_dirs; ++d) {
            const int k0 = t->host.start[d], k1 = t->host.start[d + 1];
            for (int k = k0; k < k1; ++k) off8.push_back(t->host.cy[k] * lw + t->host.cx[k]);
            while (off8.size() % 8) off8.push_back(off8.back());
            start8.push_back((int)off8.size());
        }
        if (!rc) rc = upload(off8, &t->cell_off8);
        if (!rc) rc = upload(start8, &t->dir_start8);
    }
    // plane_tab[dy][ix], dx = ix - (radius-1): index of direction (dy,dx) if it is in the set; otherwise
    // ~index of the nearest in-set direction of the same row (the sweep kernel loads that plane — hot in
    // cache — and multiplies the weight by zero, instead of branching around the slot).
    const int wcols = 2 * radius - 1;
    std::vector<int> ptab((size_t)radius * wcols, 0);
    {
        std::vector<int> exact((size_t)radius * wcols, -1);
        for (int i = 0; i < t->n_dirs; ++i) exact[(size_t)t->host.dy[i] * wcols + t->host.dx[i] + radius - 1] = i;
        for (int dy = 0; dy < radius; ++dy)
            for (int ix = 0; ix < wcols; ++ix) {
                int best = -1;
                for (int step = 0; step < wcols && best < 0; ++step) {
                    if (ix - step >= 0 && exact[(size_t)dy * wcols + ix - step] >= 0) best = exact[(size_t)dy * wcols + ix - step];
                    else if (ix + step < wcols && exact[(size_t)dy * wcols + ix + step] >= 0) best = exact[(size_t)dy * wcols + ix + step];
                }
                ptab[(size_t)dy * wcols + ix] = exact[(size_t)dy * wcols + ix] >= 0 ? best : ~best;
            }
    }
    if (!rc) rc = upload(ptab, &t->plane_tab);
    if (rc) {
        delete t;
        return rc;
    }
    g_tables[key] = t;
    *out = t;
    return IRN_OK;
}

int scratch_upload(const void *host, size_t bytes, void **dev_out, hipStream_t stream) {
    Scratch &ring = t_scratch;
    ring.last = ring.next;
    ring.next = (ring.next + 1) % kScratchSlots;
    ScratchSlot &s = ring.slot[ring.last];
    int dev = 0;
    IRN_HIP_TRY(hipGetDevice(&dev));
    if (s.ev) IRN_HIP_TRY(hipEventSynchronize(s.ev));   // the slot's previous user (kScratchSlots calls ago) is done with it
    if (bytes > s.cap || dev != s.device) {
        if (s.dev) (void)hipFree(s.dev);
        if (s.host) (void)hipHostFree(s.host);
        s.dev = s.host = nullptr;
        const size_t cap = round_up(bytes, 4096);
        IRN_HIP_TRY(hipMalloc(&s.dev, cap));
        IRN_HIP_TRY(hipHostMalloc(&s.host, cap, hipHostMallocDefault));
        s.cap = cap;
        s.device = dev;
        if (!s.ev) IRN_HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    }
    memcpy(s.host, host, bytes);
    IRN_HIP_TRY(hipMemcpyAsync(s.dev, s.host, bytes, hipMemcpyHostToDevice, stream));
    *dev_out = s.dev;
    return IRN_OK;
}

// Callers record completion of the kernels that read the scratch so the next upload can wait.
int scratch_release(hipStream_t stream) {
    ScratchSlot &s = t_scratch.slot[t_scratch.last];
    if (s.ev) IRN_HIP_TRY(hipEventRecord(s.ev, stream));
    return IRN_OK;
}

}  // namespace irn
