"""CPU ORACLE for the IRN pseudo-label hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The shipped path (``irn_amd``) never does: it fails loudly when the HIP library
is missing.

What it is: a numpy restatement of the reference algorithm (jiwoon-ahn/irn), each function citing
the reference file:line it follows.  Two forms of the random walk are given:

* ``propagate_to_edge_dense``   — line-by-line: pad, path-max affinity, symmetric dense matrix,
                                   Hadamard power, column normalise, repeated squaring, x @ T.
                                   fp32 like the reference.  O(N^3): small grids only.
* ``propagate_to_edge_stencil`` — the same operator applied as 2^exp_times sparse sweeps in
                                   fp64 (what the HIP kernels implement).

Pinning: the reference has no tests or golden vectors, so this oracle is pinned on
outputs of the reference's own code run on CPU in the build container —
``tests/golden/*.npz`` written by ``tests/golden/make_golden.py`` — see
``tests/test_oracle_golden.py``.  Third-party arithmetic on the path (torch ``pow``,
``interpolate``, ``argmax``; skimage ``label``) is restated from its documented semantics and
pinned by the same fixtures (torch 2.10 CPU kernels; scipy.ndimage.label standing in for
skimage, which is not installed).
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------
# PathIndex  (reference misc/indexing.py:6-88)
# ------------------------------------------------------------------------------------------


def search_directions(radius):
    """Half-plane neighbour set in discovery order (misc/indexing.py:22-30):
    (0,x) for 1<=x<r, then (y,x) for 1<=y<r, -r<x<r with x^2+y^2<r^2."""
    dirs = [(0, x) for x in range(1, radius)]
    for y in range(1, radius):
        for x in range(-radius + 1, radius):
            if x * x + y * y < radius * radius:
                dirs.append((y, x))
    return dirs


def path_cells(dy, dx):
    """Cells of the thick rasterised segment (0,0)->(dy,dx): all integer points of the bounding
    box whose squared distance to the line is < 1, sorted far-to-near by |y|+|x| with a stable
    sort over row-major enumeration (misc/indexing.py:32-48)."""
    lsq = dy * dy + dx * dx
    cells = []
    for y in range(min(0, dy), max(0, dy) + 1):
        for x in range(min(0, dx), max(0, dx) + 1):
            if (dy * x - dx * y) ** 2 / lsq < 1:
                cells.append((y, x))
    cells.sort(key=lambda c: -abs(c[0]) - abs(c[1]))
    return cells


def search_paths_dst(radius):
    """Paths grouped by length (ascending), groups keep discovery order; destinations = first cell
    of every path, concatenated over groups (misc/indexing.py:20-56)."""
    by_len = {}
    for d in search_directions(radius):
        p = path_cells(*d)
        by_len.setdefault(len(p), []).append(p)
    groups = [np.asarray(by_len[k], np.int64) for k in sorted(by_len)]
    dst = np.concatenate([g[:, 0] for g in groups], axis=0)
    return groups, dst


class PathIndexOracle:
    """Flat index tables on a (Hp, Wp) grid (misc/indexing.py:58-88)."""

    def __init__(self, radius, size):
        self.radius = radius
        self.radius_floor = int(math.ceil(radius) - 1)
        self.search_paths, self.search_dst = search_paths_dst(radius)
        hp, wp = size
        rf = self.radius_floor
        ch, cw = hp - rf, wp - 2 * rf
        grid = np.arange(hp * wp, dtype=np.int64).reshape(hp, wp)

        def window(dy, dx):
            return grid[dy:dy + ch, rf + dx:rf + dx + cw].reshape(-1)

        self.path_indices = [np.stack([np.stack([window(dy, dx) for dy, dx in p]) for p in g])
                             for g in self.search_paths]
        self.src_indices = window(0, 0)
        self.dst_indices = np.concatenate([pi[:, 0] for pi in self.path_indices], axis=0)


def edge_to_affinity(edge_flat, path_indices):
    """aff[d, s] = 1 - max over path(d) of edge (misc/indexing.py:91-109).  edge_flat: [Hp*Wp]."""
    out = []
    for ind in path_indices:                       # [n_paths, L, Ns]
        out.append(1 - edge_flat[ind].max(axis=1))
    return np.concatenate(out, axis=0)


def edge_to_affinity_backward(edge, grad_aff, radius):
    """Vector-Jacobian product of AffinityDisplacementLoss.to_affinity (reference net/resnet50_irn.py:
    162-175) as autograd computes it through index_select + max_pool2d: the gradient of aff[b,d,s]
    goes, negated (aff = 1 - max), to the FIRST cell of path(d) that attains the maximum.
    edge [B,Hp,Wp], grad_aff [B,|S|,(Hp-rf)*(Wp-2rf)] in the reference's channel order -> [B,Hp,Wp]."""
    edge = np.asarray(edge, np.float32)
    b, hp, wp = edge.shape
    rf = radius - 1
    ch, cw = hp - rf, wp - 2 * rf
    paths, _dst = search_paths_dst(radius)
    grad = np.zeros(edge.shape, np.float64)
    d = 0
    ys, xs = np.mgrid[0:ch, 0:cw]
    for group in paths:
        for path in group:
            vals = np.stack([edge[:, dy:dy + ch, rf + dx:rf + dx + cw] for dy, dx in path], 1)   # [B,L,ch,cw]
            arg = np.argmax(vals, 1)                                                              # first maximum
            g = np.asarray(grad_aff, np.float32)[:, d].reshape(b, ch, cw)
            pdy = np.asarray([p[0] for p in path])[arg]
            pdx = np.asarray([p[1] for p in path])[arg]
            for bi in range(b):
                np.add.at(grad[bi], (ys + pdy[bi], xs + rf + pdx[bi]), -g[bi].astype(np.float64))
            d += 1
    return grad.astype(np.float32)


def pair_displacement(disp, radius):
    """AffinityDisplacementLoss.to_pair_displacement (reference net/resnet50_irn.py:177-193):
    disp [B,C,Hp,Wp] -> [B,C,|S|,(Hp-rf)*(Wp-2rf)], source cell minus the cell (dy,dx) away, directions
    in the reference's channel order."""
    disp = np.asarray(disp, np.float32)
    b, c, hp, wp = disp.shape
    rf = radius - 1
    ch, cw = hp - rf, wp - 2 * rf
    _paths, dst = search_paths_dst(radius)
    src = disp[:, :, :ch, rf:rf + cw]
    out = np.stack([src - disp[:, :, dy:dy + ch, rf + dx:rf + dx + cw] for dy, dx in dst], 2)
    return out.reshape(b, c, len(dst), ch * cw)


def pair_displacement_backward(grad_out, radius, size):
    """Vector-Jacobian product of the above (what autograd does through the slices, the stack and the
    subtraction): +g to the source cell, -g to the destination cell; fp64 accumulation.  size = (Hp, Wp)."""
    g = np.asarray(grad_out, np.float64)
    b, c, nd, _ = g.shape
    hp, wp = size
    rf = radius - 1
    ch, cw = hp - rf, wp - 2 * rf
    g = g.reshape(b, c, nd, ch, cw)
    _paths, dst = search_paths_dst(radius)
    grad = np.zeros((b, c, hp, wp), np.float64)
    for d, (dy, dx) in enumerate(dst):
        grad[:, :, :ch, rf:rf + cw] += g[:, :, d]
        grad[:, :, dy:dy + ch, rf + dx:rf + dx + cw] -= g[:, :, d]
    return grad.astype(np.float32)


def affinity_dense(aff, src, dst, n):
    """Symmetric dense matrix with unit diagonal (misc/indexing.py:112-129)."""
    a = np.zeros((n, n), np.float32)
    rows = np.broadcast_to(src[None, :], dst.shape)
    a[rows.reshape(-1), dst.reshape(-1)] = aff.reshape(-1)
    a[dst.reshape(-1), rows.reshape(-1)] = aff.reshape(-1)
    a[np.arange(n), np.arange(n)] = 1.0
    return a


def powf(a, beta):
    """torch.pow(float32 tensor, python number) = correctly-rounded-ish powf per element; the
    fp64 pow rounded to fp32 agrees with it to <= 1 ulp (SURVEY.md §7 'Hard parts')."""
    return np.power(a.astype(np.float64), float(beta)).astype(np.float32)


def to_transition_matrix(a, beta, times):
    """misc/indexing.py:132-139 in fp32."""
    s = powf(a, beta)
    t = s / s.sum(axis=0, keepdims=True, dtype=np.float32)
    for _ in range(times):
        t = t @ t
    return t


def propagate_to_edge_dense(x, edge, radius=5, beta=10, exp_times=8):
    """misc/indexing.py:141-165 restated line by line (fp32, dense).  x: [..., h, w], edge [1,h,w]
    or [h,w]; returns [C', 1, h, w]."""
    x = np.asarray(x, np.float32)
    edge = np.asarray(edge, np.float32).reshape(x.shape[-2:])
    h, w = edge.shape
    hp, wp = h + radius, w + 2 * radius
    pi = PathIndexOracle(radius, (hp, wp))
    ep = np.ones((hp, wp), np.float32)
    ep[:h, radius:radius + w] = edge
    aff = edge_to_affinity(ep.reshape(-1), pi.path_indices)
    dense = affinity_dense(aff, pi.src_indices, pi.dst_indices, hp * wp)
    dense = dense.reshape(hp, wp, hp, wp)[:-radius, radius:-radius, :-radius, radius:-radius]
    dense = np.ascontiguousarray(dense).reshape(h * w, h * w)
    t = to_transition_matrix(dense, beta, exp_times)
    xe = x.reshape(-1, h, w) * (1 - edge)
    rw = xe.reshape(-1, h * w) @ t
    return rw.reshape(-1, 1, h, w)


# ------------------------------------------------------------------------------------------
# Stencil form (SURVEY.md §3.4 'Stencil restatement')
# ------------------------------------------------------------------------------------------

def stencil_weights(edge, radius, beta):
    """w[d, y, x] = fp32((1 - max_{c in path(d)} edge[(y,x)+c]) ** beta) for the |S| half-plane
    directions in *raster* (dy, dx) order; out-of-image cells count as edge = 1
    (misc/indexing.py:150 pads with 1.0).  Returns (dirs [|S|,2] int, w [|S|,h,w] float32)."""
    edge = np.asarray(edge, np.float32)
    h, w = edge.shape
    r = radius
    ep = np.ones((h + 2 * r, w + 2 * r), np.float32)
    ep[r:r + h, r:r + w] = edge
    dirs = sorted(search_directions(radius))
    out = np.empty((len(dirs), h, w), np.float32)
    for i, (dy, dx) in enumerate(dirs):
        m = None
        for cy, cx in path_cells(dy, dx):
            v = ep[r + cy:r + cy + h, r + cx:r + cx + w]
            m = v if m is None else np.maximum(m, v)
        out[i] = powf(1 - m, beta)
    return np.asarray(dirs, np.int32), out


def stencil_degree(dirs, wts):
    """d_p = 1 + sum_d w_d(p) + sum_d w_d(p - d), in fp64."""
    _, h, w = wts.shape
    deg = np.ones((h, w), np.float64)
    for (dy, dx), wd in zip(dirs, wts.astype(np.float64)):
        deg += wd
        ys, ye = max(0, dy), h + min(0, dy)
        xs, xe = max(0, dx), w + min(0, dx)
        deg[ys:ye, xs:xe] += wd[ys - dy:ye - dy, xs - dx:xe - dx]
    return deg


def stencil_sweep(x, dirs, wts, deg):
    """One application of the column-normalised transition operator in gather form:
    x'[c,p] = (x[c,p] + sum_{+-d} w * x[c, p+-d]) / d_p  (fp64)."""
    _, h, w = x.shape
    acc = x.copy()
    for (dy, dx), wd in zip(dirs, wts):
        ys, ye = max(0, -dy), h - max(0, dy)
        xs, xe = max(0, -dx), w - max(0, dx)
        # +d : neighbour p+d, weight stored at p
        acc[:, ys:ye, xs:xe] += wd[ys:ye, xs:xe] * x[:, ys + dy:ye + dy, xs + dx:xe + dx]
        # -d : neighbour q = p-d, weight stored at q
        acc[:, ys + dy:ye + dy, xs + dx:xe + dx] += wd[ys:ye, xs:xe] * x[:, ys:ye, xs:xe]
    return acc / deg


def propagate_to_edge_stencil(x, edge, radius=5, beta=10, exp_times=8, dtype=np.float64):
    """2^exp_times sweeps from x0 = x * (1 - edge); returns [C', 1, h, w] in ``dtype``."""
    x = np.asarray(x, np.float32)
    edge = np.asarray(edge, np.float32).reshape(x.shape[-2:])
    h, w = edge.shape
    dirs, wts = stencil_weights(edge, radius, beta)
    deg = stencil_degree(dirs, wts).astype(dtype)
    wts = wts.astype(dtype)
    cur = (x.reshape(-1, h, w) * (1 - edge)).astype(dtype)
    for _ in range(2 ** exp_times):
        cur = stencil_sweep(cur, dirs, wts, deg)
    return cur.reshape(-1, 1, h, w)


# ------------------------------------------------------------------------------------------
# Label epilogue  (reference step/make_sem_seg_labels.py:43-49, step/make_ins_seg_labels.py:137-147)
# ------------------------------------------------------------------------------------------

def _bilinear_axis(n_in, n_out, scale):
    """torch upsample_bilinear2d, align_corners=False, scale_factor given: src = (dst+0.5)/scale
    - 0.5 clamped at 0; i0 = floor(src); i1 = min(i0+1, n_in-1); weights in fp32."""
    dst = np.arange(n_out, dtype=np.float32)
    src = np.maximum((dst + np.float32(0.5)) * np.float32(1.0 / scale) - np.float32(0.5), np.float32(0))
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1) - l1).astype(np.float32)
    return i0, i1, l0, l1


def _fma32(a, b, c):
    """fp32 fused multiply-add: the product of two fp32 is exact in fp64."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def _bilinear_taps(x, y0, y1, ly0, ly1, x0, x1, lx0, lx1):
    """ATen CPU upsample_bilinear2d arithmetic (torch 2.10, pinned bit-exactly by
    tests/golden/semseg.npz and cam_merge.npz):
        top = fma(v00, lx0, v01*lx1); bot = fma(v10, lx0, v11*lx1); out = fma(ly0, top, ly1*bot)"""
    r0, r1 = x[:, y0], x[:, y1]
    top = _fma32(r0[:, :, x0], lx0, r0[:, :, x1] * lx1)
    bot = _fma32(r1[:, :, x0], lx0, r1[:, :, x1] * lx1)
    return _fma32(ly0[None, :, None], top, ly1[None, :, None] * bot)


def upsample_bilinear(x, scale=4, out_hw=None):
    """F.interpolate(x[C,1,h,w], scale_factor=scale, mode='bilinear', align_corners=False) cropped to
    out_hw, fp32."""
    x = np.asarray(x, np.float32)
    c, h, w = x.shape[0], x.shape[-2], x.shape[-1]
    x = x.reshape(c, h, w)
    H, W = h * scale, w * scale
    y0, y1, ly0, ly1 = _bilinear_axis(h, H, scale)
    x0, x1, lx0, lx1 = _bilinear_axis(w, W, scale)
    if out_hw is not None:
        oh, ow = out_hw
        y0, y1, ly0, ly1 = y0[:oh], y1[:oh], ly0[:oh], ly1[:oh]
        x0, x1, lx0, lx1 = x0[:ow], x1[:ow], lx0[:ow], lx1[:ow]
    return _bilinear_taps(x, y0, y1, ly0, ly1, x0, x1, lx0, lx1)


def sem_seg_epilogue(rw, out_hw, keys, bg_thres=0.25):
    """rw [C',1,h,w] -> (rw_up/max [C',H,W] fp32, label uint8 [H,W]).
    step/make_sem_seg_labels.py:43-49: upsample x4, crop, divide by global max, prepend constant
    background plane, argmax (first maximum wins), look up keys (0 for background, class+1)."""
    up = upsample_bilinear(rw, 4, out_hw)
    up = up / up.max()
    stack = np.concatenate([np.full((1,) + up.shape[1:], bg_thres, np.float32), up], axis=0)
    idx = np.argmax(stack, axis=0)
    lut = np.concatenate([[0], np.asarray(keys, np.int64) + 1])
    return up, lut[idx].astype(np.uint8), idx.astype(np.int32)


# ------------------------------------------------------------------------------------------
# Instance front-end  (reference step/make_ins_seg_labels.py:18-105)
# ------------------------------------------------------------------------------------------

def find_centroids_with_refinement(dp, iterations=300):
    """step/make_ins_seg_labels.py:18-56.  State is float32; `centroid - floor(centroid).astype
    (int32)` promotes to float64, so the bilinear increment is evaluated in float64 (left to
    right as written) and the in-place += rounds to float32."""
    dp = np.asarray(dp, np.float32)
    h, w = dp.shape[1:]
    cy = np.repeat(np.arange(h, dtype=np.float32)[:, None], w, axis=1)
    cx = np.repeat(np.arange(w, dtype=np.float32)[None, :], h, axis=0)
    for _ in range(iterations):
        uy, ly = np.ceil(cy).astype(np.int32), np.floor(cy).astype(np.int32)
        ux, lx = np.ceil(cx).astype(np.int32), np.floor(cx).astype(np.int32)
        fy = cy.astype(np.float64) - ly
        fx = cx.astype(np.float64) - lx
        incs = []
        for ch in (0, 1):
            f = dp[ch]
            incs.append(f[uy, ux] * fy * fx + f[ly, ux] * (1 - fy) * fx +
                        f[uy, lx] * fy * (1 - fx) + f[ly, lx] * (1 - fy) * (1 - fx))
        cy = (cy.astype(np.float64) + incs[0]).astype(np.float32)
        cx = (cx.astype(np.float64) + incs[1]).astype(np.float32)
        cy = np.clip(cy, 0, h - 1)
        cx = np.clip(cx, 0, w - 1)
    return np.stack([np.round(cy).astype(np.int32), np.round(cx).astype(np.int32)], axis=0)


def label4(mask):
    """4-connected component labelling, ids 1.. in raster order of each component's first pixel
    (semantics of skimage.measure.label(connectivity=1, background=0) used at
    step/make_ins_seg_labels.py:66,92).  Two-pass union-find."""
    mask = np.asarray(mask).astype(bool)
    h, w = mask.shape
    parent = [0]
    lab = np.zeros((h, w), np.int64)
    for y in range(h):
        for x in range(w):
            if not mask[y, x]:
                continue
            up = lab[y - 1, x] if y > 0 else 0
            left = lab[y, x - 1] if x > 0 else 0
            if up == 0 and left == 0:
                parent.append(len(parent))
                lab[y, x] = len(parent) - 1
            else:
                roots = []
                for v in (up, left):
                    if v:
                        while parent[v] != v:
                            v = parent[v]
                        roots.append(v)
                m = min(roots)
                for v in roots:
                    parent[v] = m
                lab[y, x] = m
    remap = {}
    out = np.zeros((h, w), np.int32)
    for y in range(h):
        for x in range(w):
            v = lab[y, x]
            if v:
                while parent[v] != v:
                    v = parent[v]
                if v not in remap:
                    remap[v] = len(remap) + 1
                out[y, x] = remap[v]
    return out


def compress_range(arr):
    """misc/imutils.py:182-190: renumber the distinct values of arr to 0..K-1 in ascending order."""
    uniq = np.unique(arr)
    lut = np.zeros(int(uniq.max()) + 1, np.int32)
    lut[uniq] = np.arange(uniq.shape[0])
    out = lut[arr]
    return out - out.min()


def to_one_hot(ids, maximum_val=None):
    """misc/pyutils.py:86-101 -> bool [K, *ids.shape]."""
    ids = np.asarray(ids)
    k = int(ids.max()) + 1 if maximum_val is None else int(maximum_val)
    return (np.arange(k).reshape((k,) + (1,) * ids.ndim) == ids[None]).astype(bool)


def cluster_centroids(centroids, dp, thres=2.5):
    """step/make_ins_seg_labels.py:58-75."""
    dp = np.asarray(dp, np.float32)
    strength = np.sqrt(dp[1] ** 2 + dp[0] ** 2)
    weak = strength < thres
    lab = label4(weak)
    h, w = weak.shape
    picked = lab.reshape(-1)[centroids[0] * w + centroids[1]]
    return to_one_hot(compress_range(picked.reshape(h, w) + 1))


def detect_instance(score_map, mask, class_id, max_fragment_size=0):
    """step/make_ins_seg_labels.py:82-105."""
    scores, labels, masks = [], [], []
    for sc, mk, cl in zip(score_map, mask, class_id):
        if mk.sum() < 1:
            continue
        cc = label4(mk)
        for k in range(1, int(cc.max()) + 1):
            seg = cc == k
            scores.append(0 if seg.sum() < max_fragment_size else np.max(sc * seg))
            labels.append(cl)
            masks.append(seg)
    return {"score": np.stack(scores, 0), "mask": np.stack(masks, 0), "class": np.stack(labels, 0)}


def instance_labels(cams, keys, edge, dp, out_hw, beta=10, exp_times=8, radius=5, bg_thres=0.25,
                    walk=propagate_to_edge_stencil):
    """step/make_ins_seg_labels.py:131-150 end to end for one image (walk in fp64 stencil form)."""
    cams = np.asarray(cams, np.float32)
    cen = find_centroids_with_refinement(dp)
    inst = cluster_centroids(cen, dp)
    icam = cams[:, None] * inst[None].astype(np.float32)
    rw = walk(icam, edge, radius=radius, beta=beta, exp_times=exp_times).astype(np.float32)
    up, _, idx = sem_seg_epilogue(rw, out_hw, np.zeros(rw.shape[0], np.int64), bg_thres)
    nc, ni = len(keys), inst.shape[0]
    shape = to_one_hot(idx, maximum_val=ni * nc + 1)[1:]
    cls = np.repeat(np.asarray(keys), ni)
    det = detect_instance(up, shape, cls, max_fragment_size=out_hw[0] * out_hw[1] * 0.01)
    return cen, inst, rw, idx, det


# ------------------------------------------------------------------------------------------
# Elementwise tails of the trunk and the heads  (reference net/resnet50.py:11-14, :34-54, :94-97;
# net/resnet50_irn.py:36-48, :72-84) — what irn_bn_act / irn_stem_pool / irn_upsample_bilinear compute.
# Pinned on tests/golden/trunk_ops.npz, written by the reference's own FixedBatchNorm and the torch
# modules it instantiates.
# ------------------------------------------------------------------------------------------

def fold_batch_norm(weight, bias, mean, var, eps=1e-5):
    """FixedBatchNorm (net/resnet50.py:11-14: F.batch_norm with the running statistics) as y = x * scale + shift:
    scale = w / sqrt(var + eps), shift = b - mean * scale, in fp64, rounded to fp32 once."""
    w, b, mean, var = (np.asarray(t, np.float64) for t in (weight, bias, mean, var))
    scale = w / np.sqrt(var + eps)
    return scale.astype(np.float32), (b - mean * scale).astype(np.float32)


def bn_act(x, scale, shift, res=None, relu=True, res_affine=None):
    """Bottleneck.forward's tail (net/resnet50.py:34-54): bn -> (+ residual | + bn_d(residual)) -> ReLU with the batch
    norms folded: one fp32 fused multiply-add per operand, one fp32 addition."""
    x = np.asarray(x, np.float32)
    bc = (1, -1) + (1,) * (x.ndim - 2)
    y = _fma32(x, np.asarray(scale, np.float32).reshape(bc), np.asarray(shift, np.float32).reshape(bc))
    if res is not None:
        r = np.asarray(res, np.float32)
        if res_affine is not None:
            r = _fma32(r, np.asarray(res_affine[0], np.float32).reshape(bc), np.asarray(res_affine[1], np.float32).reshape(bc))
        y = (y + r).astype(np.float32)
    return np.maximum(y, np.float32(0)) if relu else y


def stem_pool(x, scale, shift):
    """conv1's tail (net/resnet50.py:94-97): bn1 -> ReLU -> MaxPool2d(3, stride 2, padding 1) (:66); padding taps never win."""
    y = bn_act(x, scale, shift, relu=True)
    n, c, h, w = y.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    pad = np.full((n, c, 2 * ho + 1, 2 * wo + 1), -np.inf, np.float32)
    pad[:, :, 1:h + 1, 1:w + 1] = y
    out = np.full((n, c, ho, wo), -np.inf, np.float32)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, pad[:, :, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2])
    return out


def head_upsample_relu(x, factor):
    """nn.Upsample(scale_factor, 'bilinear', align_corners=False) -> ReLU of the IRNet heads (net/resnet50_irn.py:36-48,
    :72-84) with the multiply-add pattern of `_bilinear_taps`; ATen's own pattern varies with the loop specialisation it
    picks for a shape, so this is within 2 ulp of it, not bit-equal (tests/golden/trunk_ops.npz)."""
    x = np.asarray(x, np.float32)
    lead, (h, w) = x.shape[:-2], x.shape[-2:]
    up = upsample_bilinear(x.reshape((-1, 1, h, w)), scale=int(factor))
    return np.maximum(up.reshape(lead + (h * factor, w * factor)), np.float32(0))


# ------------------------------------------------------------------------------------------
# CAM merge  (reference step/make_cam.py:32-52)
# ------------------------------------------------------------------------------------------

def _resize_axis(n_in, n_out):
    """torch interpolate(size=...) align_corners=False: scale = n_in / n_out (fp32)."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = np.maximum(scale * (dst + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (np.float32(1) - l1).astype(np.float32), l1


def resize_bilinear(x, size):
    x = np.asarray(x, np.float32)
    y0, y1, ly0, ly1 = _resize_axis(x.shape[-2], size[0])
    x0, x1, lx0, lx1 = _resize_axis(x.shape[-1], size[1])
    return _bilinear_taps(x, y0, y1, ly0, ly1, x0, x1, lx0, lx1)


def cam_merge(outputs, size, label):
    """step/make_cam.py:32-52: sum over scales of bilinear resizes to the stride-4 grid and to the
    stride-16-rounded full size (cropped), keep present classes, divide by (channel max + 1e-5)."""
    H, W = size
    ss = ((H - 1) // 4 + 1, (W - 1) // 4 + 1)
    us = (((H - 1) // 16 + 1) * 16, ((W - 1) // 16 + 1) * 16)
    lo = np.zeros((outputs[0].shape[0],) + ss, np.float32)
    hi = np.zeros((outputs[0].shape[0],) + us, np.float32)
    for o in outputs:
        lo = lo + resize_bilinear(o, ss)
        hi = hi + resize_bilinear(o, us)
    hi = hi[:, :H, :W]
    keys = np.nonzero(np.asarray(label))[0]
    lo, hi = lo[keys], hi[keys]
    lo = lo / (lo.max(axis=(1, 2), keepdims=True) + np.float32(1e-5))
    hi = hi / (hi.max(axis=(1, 2), keepdims=True) + np.float32(1e-5))
    return keys.astype(np.int64), lo.astype(np.float32), hi.astype(np.float32)
