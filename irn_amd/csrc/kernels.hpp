// Device-side descriptors and launch prototypes shared by the .hip translation units.
#pragma once
#include "common.hpp"

namespace irn {

// Path table resident in device memory (library-private, immutable after upload).
struct DeviceTable {
    int radius = 0, order = 0, n_dirs = 0, n_cells = 0;
    int *dir_dy = nullptr, *dir_dx = nullptr;   // [n_dirs]
    int *dir_start = nullptr;                   // [n_dirs+1]
    int *cell_dy = nullptr, *cell_dx = nullptr; // [n_cells]
    int *cell_off8 = nullptr;                   // LDS word offset of every path cell in the affinity kernel's edge
    int *dir_start8 = nullptr;                  // tile, each path padded to a multiple of 8 cells; [n_dirs+1]
    int *plane_tab = nullptr;                   // [radius][2*radius-1]: index of direction (dy, dx=ix-(radius-1))
                                                // in this table's order, or ~(nearest in-set index of the row)
    PathTable host;
};

// cached per (radius, order) for the life of the process; uploaded on the calling thread's device
int get_device_table(int radius, int order, const DeviceTable **out);

// Copy `bytes` of host data into a library-private device scratch buffer (grown on demand) with
// stream order.  The scratch is a ring of four slots: the returned pointer stays valid until the fourth scratch_upload
// after this one on this thread (each batched entry point uploads once and enqueues its kernels before it returns).
int scratch_upload(const void *host, size_t bytes, void **dev_out, hipStream_t stream);
// Record, on `stream`, that the kernels reading the scratch have been enqueued; the next
// scratch_upload that reuses this slot waits for them before it overwrites the buffers.
int scratch_release(hipStream_t stream);

// tile of the affinity kernel: source rows x cols per workgroup (one wave covers two rows)
constexpr int kAffTileH = 8, kAffTileW = 32;

// One grid handed to the affinity kernel.
struct AffJob {
    const float *edge;   // [gh, gw]
    float *out;          // plane 0, element of source pixel (0,0)
    int gh, gw;          // grid the edge lives on
    int oy, ox;          // origin of the source rectangle inside the grid
    int sh, sw;          // source rectangle (output plane is [sh, sw] row-major)
    long plane_stride;   // floats between consecutive direction planes
};

int launch_affinity(const AffJob *jobs_dev, int n_jobs, int max_sh, int max_sw, const DeviceTable &tab,
                    bool with_pow, float beta, hipStream_t stream);

}  // namespace irn
