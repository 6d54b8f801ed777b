// Host construction of the radial path table — the part of the reference's PathIndex that does not
// depend on the image size (misc/indexing.py:18-56).  The size-dependent flat index tensors
// (:58-88, 32 MB-1.1 GB of int64 per image, re-uploaded per image at :96-99) are never built:
// the kernels turn (dy,dx) into addresses themselves.
#include <algorithm>
#include <cstdlib>
#include <map>

#include "common.hpp"

namespace irn {

namespace {
struct Cell {
    int y, x;
};

// Thick rasterised segment (0,0)->(dy,dx): every lattice point of the bounding box whose squared
// distance to the line, (dy*x - dx*y)^2 / (dy^2+dx^2), is < 1 (misc/indexing.py:37-46); ordered
// far-to-near by |y|+|x| with row-major enumeration as the stable tie order (:48).
std::vector<Cell> thick_segment(int dy, int dx) {
    const int lsq = dy * dy + dx * dx;
    std::vector<Cell> cells;
    for (int y = std::min(0, dy); y <= std::max(0, dy); ++y)
        for (int x = std::min(0, dx); x <= std::max(0, dx); ++x) {
            const int cross = dy * x - dx * y;
            if (cross * cross < lsq) cells.push_back({y, x});
        }
    std::stable_sort(cells.begin(), cells.end(), [](const Cell &a, const Cell &b) {
        return std::abs(a.y) + std::abs(a.x) > std::abs(b.y) + std::abs(b.x);
    });
    return cells;
}
}  // namespace

PathTable build_path_table(int radius, int order) {
    PathTable t;
    t.radius = radius;
    // discovery order (misc/indexing.py:24-30)
    std::vector<Cell> dirs;
    for (int x = 1; x < radius; ++x) dirs.push_back({0, x});
    for (int y = 1; y < radius; ++y)
        for (int x = -radius + 1; x < radius; ++x)
            if (x * x + y * y < radius * radius) dirs.push_back({y, x});

    std::vector<std::vector<Cell>> paths;
    for (auto &d : dirs) paths.push_back(thick_segment(d.y, d.x));

    std::vector<int> perm(dirs.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
    if (order == 0) {
        // grouped by path length, ascending; discovery order inside a group (:50-53)
        std::stable_sort(perm.begin(), perm.end(),
                         [&](int a, int b) { return paths[a].size() < paths[b].size(); });
    } else {
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
            return dirs[a].y != dirs[b].y ? dirs[a].y < dirs[b].y : dirs[a].x < dirs[b].x;
        });
    }
    t.start.push_back(0);
    for (int i : perm) {
        t.dy.push_back(dirs[i].y);
        t.dx.push_back(dirs[i].x);
        for (auto &c : paths[i]) {
            t.cy.push_back(c.y);
            t.cx.push_back(c.x);
        }
        t.start.push_back((int)t.cy.size());
    }
    return t;
}

}  // namespace irn
