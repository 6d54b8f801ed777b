// Shared host-side helpers for libirn_hip.so (gfx950 only; no other back end exists).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/irn_hip.h"

namespace irn {

inline std::string &last_error_slot() {
    static thread_local std::string s;
    return s;
}

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

#define IRN_HIP_TRY(expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::irn::fail(IRN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                           \
    } while (0)

#define IRN_LAUNCH_CHECK(name)                                                                \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return ::irn::fail(IRN_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------
// Radial path table (reference misc/indexing.py:18-56), built on the host once per radius.
// ------------------------------------------------------------------------------------------
struct PathTable {
    int radius = 0;
    std::vector<int> dy, dx;      // destination of each direction
    std::vector<int> start;       // CSR offsets into cy/cx, size n_dirs+1
    std::vector<int> cy, cx;      // path cells, far-to-near
    int n_dirs() const { return (int)dy.size(); }
    int n_cells() const { return (int)cy.size(); }
};

// order 0 = reference channel order, 1 = raster (dy,dx) order.
PathTable build_path_table(int radius, int order);

// index (in raster order) of direction (dy,dx) of the half-plane set, or -1
inline bool in_half_plane_set(int radius, int dy, int dx) {
    if (dy == 0) return dx >= 1 && dx < radius;
    if (dy < 0 || dy >= radius) return false;
    return dx > -radius && dx < radius && dx * dx + dy * dy < radius * radius;
}

}  // namespace irn
