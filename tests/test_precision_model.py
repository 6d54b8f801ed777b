"""Why the sweep kernel accumulates the way it does (irn_amd/csrc/walk.hip header): a numpy model of
three accumulation schemes over 256 sweeps against the fp64 operator."""
import numpy as np

from oracle import irn_oracle as O


def _sweep(x, dirs, wts, deg, scheme):
    _, h, w = x.shape
    x = x.astype(np.float32)
    W = wts.astype(np.float32)
    if scheme == "f32":
        acc = x.copy()
    parts = {}
    for (dy, dx), wd in zip(dirs, W):
        ys, ye = max(0, -dy), h - max(0, dy)
        xs, xe = max(0, -dx), w - max(0, dx)
        if scheme == "f32":
            acc[:, ys:ye, xs:xe] += wd[ys:ye, xs:xe] * x[:, ys + dy:ye + dy, xs + dx:xe + dx]
            acc[:, ys + dy:ye + dy, xs + dx:xe + dx] += wd[ys:ye, xs:xe] * x[:, ys:ye, xs:xe]
        else:   # fp32 partial per neighbour row, fp64 combine — the kernel's scheme
            a = parts.setdefault(dy, np.zeros_like(x))
            a[:, ys:ye, xs:xe] += wd[ys:ye, xs:xe] * x[:, ys + dy:ye + dy, xs + dx:xe + dx]
            b = parts.setdefault(-dy, np.zeros_like(x))
            b[:, ys + dy:ye + dy, xs + dx:xe + dx] += wd[ys:ye, xs:xe] * x[:, ys:ye, xs:xe]
    if scheme == "f32":
        return acc / deg.astype(np.float32)
    acc = x.astype(np.float64)
    for k in sorted(parts):
        acc += parts[k].astype(np.float64)
    return (acc / deg).astype(np.float32)


def test_row_partial_scheme_tracks_fp64(golden):
    wk = golden("walk")
    n = "r10_b10_e8"
    h, w, c, r, b, e = wk[n + "_params"]
    x0 = (wk[n + "_cam"] * (1 - wk[n + "_edge"])).astype(np.float32)
    dirs, wts = O.stencil_weights(wk[n + "_edge"], r, b)
    deg = O.stencil_degree(dirs, wts)
    truth = O.propagate_to_edge_stencil(wk[n + "_cam"], wk[n + "_edge"], r, b, e)[:, 0]
    err = {}
    for scheme in ("f32", "rows"):
        cur = x0
        for _ in range(2 ** e):
            cur = _sweep(cur, dirs, wts, deg, scheme)
        err[scheme] = np.abs(cur - truth).max()
    assert err["rows"] <= 5e-6          # as good as full fp64 accumulation
    assert err["f32"] >= 5e-5           # plain fp32 accumulation is 100x worse and misses the 1e-4 bar vs the reference
    assert np.abs(truth[:, None] - wk[n + "_rw"]).max() <= 1e-4
