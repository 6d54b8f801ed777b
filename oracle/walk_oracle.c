/*
 * CPU ORACLE (C) for the IRN random walk — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of reference misc/indexing.py:141-165 (propagate_to_edge) in its sparse
 * stencil form, fp64 throughout — the same algorithm as oracle/irn_oracle.py
 * (propagate_to_edge_stencil), fast enough to serve at BASELINE sizes and as the timed CPU
 * baseline ("port").  Pinned against the reference through tests/test_oracle_c.py, which checks
 * it against tests/golden/walk.npz (outputs of the reference itself).
 *
 *   path cells        misc/indexing.py:32-48   thick rasterised segment, squared distance < 1
 *   direction set     misc/indexing.py:22-30   half plane, x^2+y^2 < r^2
 *   padding = 1.0     misc/indexing.py:150     out-of-image cells are boundaries
 *   aff = 1 - max     misc/indexing.py:103-105
 *   ^beta, col-norm   misc/indexing.py:133-135
 *   x0 = cam*(1-edge) misc/indexing.py:162
 *   2^exp_times steps misc/indexing.py:136-137,164 (matrix squaring == repeated application)
 *
 * build:  gcc -O3 -fopenmp -shared -fPIC -o oracle/_build/libwalk_oracle.so oracle/walk_oracle.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int in_set(int r, int dy, int dx) {
    if (dy == 0) return dx >= 1 && dx < r;
    return dy > 0 && dy < r && dx > -r && dx < r && dx * dx + dy * dy < r * r;
}

int irn_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out[c,h,w] = (cam * (1-edge)) . T^n_sweeps ; returns 0 on success */
int irn_oracle_walk(const float *edge, const float *cam, int C, int h, int w, int radius, double beta,
                    int n_sweeps, float *out) {
    const int r = radius, n = h * w;
    int nd = 0;
    for (int dy = 0; dy < r; ++dy)
        for (int dx = -r + 1; dx < r; ++dx) nd += in_set(r, dy, dx);
    int *ddy = (int *)malloc(sizeof(int) * nd), *ddx = (int *)malloc(sizeof(int) * nd);
    double *wt = (double *)malloc(sizeof(double) * (size_t)nd * n);
    double *deg = (double *)malloc(sizeof(double) * n);
    double *xa = (double *)malloc(sizeof(double) * (size_t)C * n), *xb = (double *)malloc(sizeof(double) * (size_t)C * n);
    if (!ddy || !ddx || !wt || !deg || !xa || !xb) return 1;
    int k = 0;
    for (int dy = 0; dy < r; ++dy)
        for (int dx = -r + 1; dx < r; ++dx)
            if (in_set(r, dy, dx)) { ddy[k] = dy; ddx[k] = dx; ++k; }

    /* weights: w_d(p) = fp32((1 - max over path(d) of edge)^beta), out-of-image = 1 */
#pragma omp parallel for schedule(static)
    for (int d = 0; d < nd; ++d) {
        const int dy = ddy[d], dx = ddx[d], lsq = dy * dy + dx * dx;
        const int x0 = dx < 0 ? dx : 0, x1 = dx > 0 ? dx : 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float m = -INFINITY;
                for (int cy = 0; cy <= dy; ++cy)
                    for (int cx = x0; cx <= x1; ++cx) {
                        const int cross = dy * cx - dx * cy;
                        if (cross * cross >= lsq) continue;
                        const int yy = y + cy, xx = x + cx;
                        const float e = (yy < h && xx >= 0 && xx < w) ? edge[yy * w + xx] : 1.0f;
                        if (e > m) m = e;
                    }
                const float a = 1.0f - m;
                wt[(size_t)d * n + y * w + x] = (double)(float)pow((double)a, beta);
            }
    }
    /* degree = 1 + sum over both directions */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double s = 1.0;
            for (int d = 0; d < nd; ++d) {
                s += wt[(size_t)d * n + y * w + x];
                const int yy = y - ddy[d], xx = x - ddx[d];
                if (yy >= 0 && xx >= 0 && xx < w) s += wt[(size_t)d * n + yy * w + xx];
            }
            deg[y * w + x] = s;
        }
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < n; ++p) xa[(size_t)c * n + p] = (double)(cam[(size_t)c * n + p] * (1.0f - edge[p]));

    double *src = xa, *dst = xb;
    for (int t = 0; t < n_sweeps; ++t) {
#pragma omp parallel for schedule(static) collapse(2)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y) {
                const double *xc = src + (size_t)c * n;
                for (int x = 0; x < w; ++x) {
                    const int p = y * w + x;
                    double acc = xc[p];
                    for (int d = 0; d < nd; ++d) {
                        const int dy = ddy[d], dx = ddx[d];
                        const double *wd = wt + (size_t)d * n;
                        if (y + dy < h && x + dx >= 0 && x + dx < w) acc += wd[p] * xc[p + dy * w + dx];
                        if (y - dy >= 0 && x - dx >= 0 && x - dx < w) acc += wd[p - dy * w - dx] * xc[p - dy * w - dx];
                    }
                    dst[(size_t)c * n + p] = acc / deg[p];
                }
            }
        double *tmp = src; src = dst; dst = tmp;
    }
    for (size_t i = 0; i < (size_t)C * n; ++i) out[i] = (float)src[i];
    free(ddy); free(ddx); free(wt); free(deg); free(xa); free(xb);
    return 0;
}


/* ------------------------------------------------------------------------------------------------------------
 * The same algorithm laid out the way a CPU wants it — the timed CPU baseline of bench.py (kind "port"):
 *   - one image per thread (images are independent units: the reference's own sharding, misc/torchutils.py:66-68),
 *     no fork/join inside an image (the per-sweep parallel regions above cost more than a 128x128 sweep itself);
 *   - direction-outer, row-inner loops: for every stored direction the forward and the backward contribution of a
 *     whole row are unit-stride multiply-adds the compiler vectorises; the |S| weight planes stream once per sweep.
 * fp64 throughout like irn_oracle_walk; the summation order differs (per direction instead of per pixel), i.e.
 * results agree to fp64 rounding.  Pinned by tests/test_oracle_c.py against irn_oracle_walk and the reference's
 * golden outputs.
 * ---------------------------------------------------------------------------------------------------------- */
static double powi_or_pow(double a, double beta) {
    const int bi = (int)beta;
    if ((double)bi == beta && bi >= 1 && bi <= 64) {
        double r = 1.0, b = a;
        for (int e = bi; e; e >>= 1) {
            if (e & 1) r *= b;
            b *= b;
        }
        return r;
    }
    return pow(a, beta);
}

static int walk_rows(const float *edge, const float *cam, int C, int h, int w, int radius, double beta, int n_sweeps,
                     float *out) {
    const int r = radius, n = h * w;
    int nd = 0;
    for (int dy = 0; dy < r; ++dy)
        for (int dx = -r + 1; dx < r; ++dx) nd += in_set(r, dy, dx);
    int *ddy = (int *)malloc(sizeof(int) * nd), *ddx = (int *)malloc(sizeof(int) * nd);
    double *wt = (double *)malloc(sizeof(double) * (size_t)nd * n);
    double *inv = (double *)malloc(sizeof(double) * n);
    double *xa = (double *)malloc(sizeof(double) * (size_t)C * n), *xb = (double *)malloc(sizeof(double) * (size_t)C * n);
    if (!ddy || !ddx || !wt || !inv || !xa || !xb) return 1;
    int k = 0;
    for (int dy = 0; dy < r; ++dy)
        for (int dx = -r + 1; dx < r; ++dx)
            if (in_set(r, dy, dx)) { ddy[k] = dy; ddx[k] = dx; ++k; }
    for (int d = 0; d < nd; ++d) {
        const int dy = ddy[d], dx = ddx[d], lsq = dy * dy + dx * dx;
        const int x0 = dx < 0 ? dx : 0, x1 = dx > 0 ? dx : 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float m = -INFINITY;
                for (int cy = 0; cy <= dy; ++cy)
                    for (int cx = x0; cx <= x1; ++cx) {
                        const int cross = dy * cx - dx * cy;
                        if (cross * cross >= lsq) continue;
                        const int yy = y + cy, xx = x + cx;
                        const float e = (yy < h && xx >= 0 && xx < w) ? edge[yy * w + xx] : 1.0f;
                        if (e > m) m = e;
                    }
                /* a pair whose far end lies outside the image does not exist: weight 0 so the row loops need no test */
                const int inside = (y + dy < h && x + dx >= 0 && x + dx < w);
                wt[(size_t)d * n + y * w + x] = inside ? (double)(float)powi_or_pow((double)(1.0f - m), beta) : 0.0;
            }
    }
    for (int p = 0; p < n; ++p) inv[p] = 1.0;
    for (int d = 0; d < nd; ++d) {
        const double *wd = wt + (size_t)d * n;
        const int off = ddy[d] * w + ddx[d];
        for (int p = 0; p < n; ++p) {
            if (wd[p] != 0.0) {          /* unit diagonal + both ends of every existing pair */
                inv[p] += wd[p];
                inv[p + off] += wd[p];
            }
        }
    }
    for (int p = 0; p < n; ++p) inv[p] = 1.0 / inv[p];
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < n; ++p) xa[(size_t)c * n + p] = (double)(cam[(size_t)c * n + p] * (1.0f - edge[p]));
    double *src = xa, *dst = xb;
    for (int t = 0; t < n_sweeps; ++t) {
        memcpy(dst, src, sizeof(double) * (size_t)C * n);
        for (int d = 0; d < nd; ++d) {
            const int dy = ddy[d], dx = ddx[d], off = dy * w + dx;
            const double *wd = wt + (size_t)d * n;
            const int xs = dx < 0 ? -dx : 0, xe = dx > 0 ? w - dx : w;
            for (int c = 0; c < C; ++c) {
                const double *xc = src + (size_t)c * n;
                double *ac = dst + (size_t)c * n;
                for (int y = 0; y + dy < h; ++y) {
                    const double *wr = wd + y * w, *xr = xc + y * w;
                    double *ar = ac + y * w;
                    for (int x = xs; x < xe; ++x) ar[x] += wr[x] * xr[x + off];            /* p gathers from p + d */
                    for (int x = xs; x < xe; ++x) ar[x + off] += wr[x] * xr[x];            /* p + d gathers from p */
                }
            }
        }
        for (int c = 0; c < C; ++c)
            for (int p = 0; p < n; ++p) dst[(size_t)c * n + p] *= inv[p];
        double *tmp = src; src = dst; dst = tmp;
    }
    for (size_t i = 0; i < (size_t)C * n; ++i) out[i] = (float)src[i];
    free(ddy); free(ddx); free(wt); free(inv); free(xa); free(xb);
    return 0;
}

/* n independent images, one per thread at a time (dynamic schedule); returns the number of failed images */
int irn_oracle_walk_batch(int n_images, const float *const *edge, const float *const *cam, const int *C, const int *h,
                          const int *w, int radius, double beta, int n_sweeps, float *const *out) {
    int failed = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed)
    for (int i = 0; i < n_images; ++i)
        failed += walk_rows(edge[i], cam[i], C[i], h[i], w[i], radius, beta, n_sweeps, out[i]);
    return failed;
}
