"""Operator tier of the pseudo-label hot path — API mirror of reference misc/indexing.py, backed by
libirn_hip.so (hand-written gfx950 kernels, C ABI in include/irn_hip.h).

Same names, argument meaning and return shapes as the reference:

    PathIndex(radius, default_size)                      misc/indexing.py:6-88
    edge_to_affinity(edge, paths_indices)                misc/indexing.py:91-109
    affinity_sparse2dense(aff, ind_from, ind_to, n)      misc/indexing.py:112-129
    to_transition_matrix(affinity_dense, beta, times)    misc/indexing.py:132-139
    propagate_to_edge(x, edge, radius, beta, exp_times)  misc/indexing.py:141-165

plus the batched form the steps and bench use (``RandomWalk``).  Tensors must live on the GPU; there
is no CPU implementation here (the CPU restatement is test infrastructure under ``oracle/``).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .._lib import check, i32_array, lib, ptr_array


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise ValueError("%s must be a GPU tensor: the HIP path has no CPU fallback" % what)


# ------------------------------------------------------------------------------------------------
# PathIndex
# ------------------------------------------------------------------------------------------------

def _path_table(radius, order):
    nd, nc = C.c_int(), C.c_int()
    check(lib.irn_path_count(int(radius), C.byref(nd), C.byref(nc)))
    dst = np.empty((nd.value, 2), np.int32)
    start = np.empty(nd.value + 1, np.int32)
    cells = np.empty((nc.value, 2), np.int32)
    check(lib.irn_path_table(int(radius), int(order), dst.ctypes.data_as(_lib.pi32),
                             start.ctypes.data_as(_lib.pi32), cells.ctypes.data_as(_lib.pi32)))
    return dst, start, cells


class _PathIndexList(list):
    """``path_indices`` as the reference exposes it (a list of int64 arrays [n_paths, L, Ns]) that
    also remembers which (radius, grid) it belongs to, so ``edge_to_affinity`` can run the HIP
    kernel without ever touching the index arrays."""
    radius = None
    size = None


class PathIndex:
    """Radial path tables (reference misc/indexing.py:6-88).

    ``search_paths`` / ``search_dst`` come from the library (irn_path_table, reference channel
    order).  The flat int64 index tensors ``path_indices / src_indices / dst_indices`` exist for
    API compatibility (training code indexes with them); they are built lazily with numpy and the
    kernels never read them."""

    def __init__(self, radius, default_size):
        self.radius = radius
        self.radius_floor = int(np.ceil(radius) - 1)
        self.default_size = tuple(int(v) for v in default_size)
        dst, start, cells = _path_table(radius, 0)
        lens = np.diff(start)
        self.search_dst = dst.astype(np.int64)
        self.search_paths = []
        for length in np.unique(lens):                      # groups ascending by path length
            sel = np.nonzero(lens == length)[0]
            self.search_paths.append(np.stack([cells[start[i]:start[i + 1]] for i in sel]).astype(np.int64))
        self._indices = None

    def _build_indices(self):
        hp, wp = self.default_size
        rf = self.radius_floor
        ch, cw = hp - rf, wp - 2 * rf
        grid = np.arange(hp * wp, dtype=np.int64).reshape(hp, wp)

        def window(dy, dx):
            return grid[dy:dy + ch, rf + dx:rf + dx + cw].reshape(-1)

        plist = _PathIndexList()
        plist.radius, plist.size = self.radius, self.default_size
        for group in self.search_paths:
            plist.append(np.stack([np.stack([window(int(dy), int(dx)) for dy, dx in path]) for path in group]))
        src = window(0, 0)
        dst = np.concatenate([p[:, 0] for p in plist], axis=0)
        self._indices = (plist, src, dst)

    @property
    def path_indices(self):
        if self._indices is None:
            self._build_indices()
        return self._indices[0]

    @property
    def src_indices(self):
        if self._indices is None:
            self._build_indices()
        return self._indices[1]

    @property
    def dst_indices(self):
        if self._indices is None:
            self._build_indices()
        return self._indices[2]


class _EdgeToAffinity(torch.autograd.Function):
    """irn_edge_to_affinity with irn_edge_to_affinity_backward as its vector-Jacobian product (the
    gradient of aff[b,d,s] goes, negated, to the first path cell attaining the maximum — what autograd
    does through the reference's index_select + max_pool2d, net/resnet50_irn.py:162-175)."""

    @staticmethod
    def forward(ctx, e, radius, hp, wp):
        rf = int(np.ceil(radius) - 1)
        nd, nc = C.c_int(), C.c_int()
        check(lib.irn_path_count(int(radius), C.byref(nd), C.byref(nc)))
        out = torch.empty((e.size(0), nd.value, (hp - rf) * (wp - 2 * rf)), device=e.device, dtype=torch.float32)
        with torch.cuda.device(e.device):
            check(lib.irn_edge_to_affinity(e.data_ptr(), e.size(0), hp, wp, int(radius), out.data_ptr(), _stream()))
        ctx.save_for_backward(e)
        ctx.geom = (int(radius), hp, wp)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (e,) = ctx.saved_tensors
        radius, hp, wp = ctx.geom
        g = grad_out.contiguous().float()
        grad_edge = torch.empty_like(e)
        with torch.cuda.device(e.device):
            check(lib.irn_edge_to_affinity_backward(e.data_ptr(), g.data_ptr(), e.size(0), hp, wp, radius,
                                                    grad_edge.data_ptr(), _stream()))
        return grad_edge, None, None, None


def edge_to_affinity(edge, paths_indices=None, radius=None, size=None):
    """aff[b, d, s] = 1 - max over path(d) of edge (reference misc/indexing.py:91-109 and, under
    autograd, AffinityDisplacementLoss.to_affinity net/resnet50_irn.py:162-175).

    ``edge``: GPU float tensor [B, Hp*Wp] or [B, 1, Hp, Wp] (viewed as [B, -1] like the reference).
    ``paths_indices``: ``PathIndex.path_indices`` (carries radius and grid size) — or pass
    ``radius=`` and ``size=(Hp, Wp)`` directly.  Returns [B, |S|, (Hp-rf)*(Wp-2rf)] in the
    reference's channel order.  Differentiable w.r.t. ``edge``."""
    _need_cuda(edge, "edge")
    if paths_indices is not None and radius is None:
        radius, size = getattr(paths_indices, "radius", None), getattr(paths_indices, "size", None)
    if radius is None or size is None:
        raise ValueError("edge_to_affinity needs PathIndex.path_indices or explicit radius= and size=")
    hp, wp = int(size[0]), int(size[1])
    e = edge.reshape(edge.size(0), -1).contiguous().float()
    if e.size(1) != hp * wp:
        raise ValueError("edge has %d elements per item, grid is %dx%d" % (e.size(1), hp, wp))
    return _EdgeToAffinity.apply(e, radius, hp, wp)


class _PairDisplacement(torch.autograd.Function):
    """irn_pair_displacement / irn_pair_displacement_backward."""

    @staticmethod
    def forward(ctx, disp, radius):
        b, c, hp, wp = disp.shape
        rf = int(np.ceil(radius) - 1)
        nd, nc = C.c_int(), C.c_int()
        check(lib.irn_path_count(int(radius), C.byref(nd), C.byref(nc)))
        out = torch.empty((b, c, nd.value, (hp - rf) * (wp - 2 * rf)), device=disp.device, dtype=torch.float32)
        with torch.cuda.device(disp.device):
            check(lib.irn_pair_displacement(disp.data_ptr(), b, c, hp, wp, int(radius), out.data_ptr(), _stream()))
        ctx.geom = (int(radius), b, c, hp, wp)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        radius, b, c, hp, wp = ctx.geom
        g = grad_out.contiguous().float()
        grad = torch.empty((b, c, hp, wp), device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            check(lib.irn_pair_displacement_backward(g.data_ptr(), b, c, hp, wp, radius, grad.data_ptr(), _stream()))
        return grad, None


def pair_displacement(disp, radius):
    """pair_disp[b,c,d,s] = disp at the source cell minus disp at the cell (dy_d, dx_d) away, over the cropped grid
    (AffinityDisplacementLoss.to_pair_displacement, net/resnet50_irn.py:177-193).  ``disp``: GPU float
    [B, C, Hp, Wp]; returns [B, C, |S|, (Hp-rf)*(Wp-2rf)] in the reference's channel order.  Differentiable."""
    _need_cuda(disp, "disp")
    if disp.dim() != 4:
        raise ValueError("pair_displacement: [B, C, Hp, Wp] expected")
    return _PairDisplacement.apply(disp.contiguous().float(), radius)


def affinity_sparse2dense(affinity_sparse, ind_from, ind_to, n_vertices):
    """Dense symmetric affinity with unit diagonal (reference misc/indexing.py:112-129).
    Kept for API completeness only — the walk never densifies; built on-device with index_put
    instead of the reference's device->host->to_dense->device round trip."""
    dev = affinity_sparse.device
    a = affinity_sparse.reshape(-1)
    f = torch.as_tensor(np.asarray(ind_from), device=dev).repeat(np.asarray(ind_to).shape[0]).reshape(-1)
    t = torch.as_tensor(np.asarray(ind_to), device=dev).reshape(-1)
    dense = torch.zeros((n_vertices, n_vertices), device=dev, dtype=a.dtype)
    dense[f, t] = a
    dense[t, f] = a
    d = torch.arange(n_vertices, device=dev)
    dense[d, d] = 1.0
    return dense


def to_transition_matrix(affinity_dense, beta, times):
    """Reference misc/indexing.py:132-139 verbatim semantics (dense; API completeness only)."""
    s = torch.pow(affinity_dense, beta)
    t = s / torch.sum(s, dim=0, keepdim=True)
    for _ in range(times):
        t = torch.matmul(t, t)
    return t


# ------------------------------------------------------------------------------------------------
# Random walk
# ------------------------------------------------------------------------------------------------

def _poll_delay_file(device):
    from ..step import _common
    base = os.environ.get("IRN_TUNING_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "irn_amd", "walk")
    return os.path.join(base, "%s-dev%d.json" % (_common.miopen_cache_key(), torch.device(device).index or 0))


def _lib_stamp():
    from .._lib import LIB_PATH
    st = os.stat(LIB_PATH)
    return [st.st_mtime_ns, st.st_size]


def _saved_poll_delay(device):
    """The poll delay an earlier process's start-up probe picked on this device with THIS build of the library (0: none).
    IRN_POLL_DELAY_CACHE=0 ignores the file."""
    if os.environ.get("IRN_POLL_DELAY_CACHE", "1") == "0":
        return 0
    try:
        import json
        with open(_poll_delay_file(device)) as f:
            d = json.load(f)
        return int(d["poll_delay"]) if d.get("lib") == _lib_stamp() else 0
    except Exception:
        return 0


def _save_poll_delay(device, delay):
    if os.environ.get("IRN_POLL_DELAY_CACHE", "1") == "0":
        return
    try:
        import json
        path = _poll_delay_file(device)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump({"lib": _lib_stamp(), "poll_delay": int(delay)}, f)
        os.replace(tmp, path)
    except Exception:
        pass


class RandomWalk:
    """Batched affinity random walk on one GPU (wraps an ``irn_walk_ctx``).

        rw = RandomWalk(radius=5)
        outs = rw(edges, cams, beta=10, exp_times=8)

    ``edges[i]`` [h,w] (or [1,h,w]) and ``cams[i]`` [C,h,w] are GPU fp32 tensors of one image;
    ``outs[i]`` is [C,1,h,w] like ``propagate_to_edge``.  With ``inst_maps`` (int32 [h,w] cluster maps,
    ``k_inst[i]`` instances) channel cls*K+k starts from cam[cls]*(inst==k)
    (reference step/make_ins_seg_labels.py:77-80,:133) and ``outs[i]`` is [C*K,1,h,w]."""

    _warned_fallback = False

    def __init__(self, radius=5, device=None):
        self.radius = int(radius)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.irn_walk_create(self.radius, C.byref(self._ctx)))
        self._sig = None
        self._ws = None
        self._ws_bytes = 0
        self._live = None      # inputs of the last run: the re-run of `sync()` reads them again (include/irn_hip.h)
        # The schedule is a deliberate deviation from the reference's computation (misc/indexing.py:136-137 squares the
        # matrix exp_times times; this applies a polynomial of the operator): the user-reachable switch.  IRN_WALK_ACCEL=0
        # restores the plain 2^exp_times applications everywhere (run_sample.py --walk_accel 0 sets it for the steps and
        # their worker processes), IRN_WALK_ACCEL_TOL_EXP = e moves the series' truncation bound to 10^-e.
        env = os.environ.get("IRN_WALK_ACCEL")
        if env not in (None, ""):
            self.set_option("accel", 1 if int(env) else 0)
        env = os.environ.get("IRN_WALK_ACCEL_TOL_EXP")
        if env not in (None, ""):
            self.set_option("accel_tol_exp", int(env))
        env = os.environ.get("IRN_POLL_DELAY")            # pins the single-channel poll delay (no start-up probe)
        self._poll_pinned = env not in (None, "")
        self.poll_delay_source = "library default / start-up probe"
        if self._poll_pinned:
            self.set_option("poll_delay", int(env))
            self.poll_delay_source = "IRN_POLL_DELAY"
        elif self.radius == 10:
            # what an earlier process measured on this device with this build of the library: every pool worker would
            # otherwise run its first representative batch 8 extra times, blocking its host thread
            d = _saved_poll_delay(self.device)
            if d:
                self.set_option("poll_delay", d)
                self._poll_pinned = True
                self.poll_delay_source = "per-device cache"

    def close(self):
        if self._ctx:
            lib.irn_walk_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        check(lib.irn_walk_set_option(self._ctx, name.encode(), int(value)))
        self._sig = None

    def sync(self):
        """Wait for the last run and make its outputs valid: if the persistent kernel (variant 2) gave up its bounded
        wait for a neighbouring tile — the grid was not co-resident in time — the batch is run again on the streaming
        sweeps (irn_walk_sync).  Returns True when that happened.  Call before consuming the outputs."""
        fell = C.c_int()
        with torch.cuda.device(self.device):
            check(lib.irn_walk_sync(self._ctx, C.byref(fell)))
        self._live = None
        if fell.value and not RandomWalk._warned_fallback:
            RandomWalk._warned_fallback = True
            import warnings
            warnings.warn("irn_amd: a weights-stationary walk launch gave up waiting for a neighbouring tile (the grid was not co-resident "
                          "in time: another tenant on the GPU, or two workers sharing a device) and the batch was re-run on the streaming "
                          "sweeps, ~11x slower; results are unaffected.  `fallback_runs` counts these (this warning is shown once).")
        return bool(fell.value)

    def steps(self, n_sweeps):
        """Operator applications the context spends on x . T^n_sweeps (irn_walk_steps): n_sweeps itself for the plain
        powers, ~sqrt(2 n ln(1/tol)) with the truncated Chebyshev series (option accel=1, the default)."""
        k = C.c_int()
        check(lib.irn_walk_steps(self._ctx, int(n_sweeps), C.byref(k)))
        return int(k.value)

    @property
    def fallback_runs(self):
        return int(lib.irn_walk_fallback_runs(self._ctx))

    def tuning(self):
        """{'poll_delay', 'placement' (0 unchecked / 1 block -> XCD round robin holds / 2 it does not), 'probe_ms' (launch
        times the start-up probe measured for poll delays 8, 10, 12, 14, or None)} — irn_walk_tuning."""
        d, pl = C.c_int(), C.c_int()
        ms = (C.c_float * 4)()
        check(lib.irn_walk_tuning(self._ctx, C.byref(d), C.byref(pl), ms))
        probed = any(v > 0 for v in ms)
        return {"poll_delay": int(d.value), "placement": int(pl.value), "probe_ms": [float(v) for v in ms] if probed else None}

    def check(self):
        """Raise if a weights-stationary launch (option variant=2) gave up waiting for a neighbouring
        tile and `sync()` has not repaired it.  Synchronises the device first (the kernel reports through a
        pinned word).  For benchmarks, where a silent re-run on the slower path would misreport."""
        torch.cuda.synchronize(self.device)
        check(lib.irn_walk_check(self._ctx))

    def read_profile(self):
        """int64 [2,256,4] per-sweep time stamps (10 ns ticks) of two workgroups (option profile=1)."""
        import numpy as np
        buf = np.zeros((2, 256, 4), np.int64)
        torch.cuda.synchronize(self.device)
        check(lib.irn_walk_read_profile(self._ctx, buf.ctypes.data))
        return buf

    def enable_timing(self, on=True):
        with torch.cuda.device(self.device):
            check(lib.irn_walk_enable_timing(self._ctx, 1 if on else 0))

    def last_sweep_ms(self):
        ms, n = C.c_float(), C.c_int()
        check(lib.irn_walk_last_sweep_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def configure(self, shapes):
        """shapes: list of (h, w, c).  Sizes the workspace; cached while the shapes stay the same."""
        sig = tuple((int(h), int(w), int(c)) for h, w, c in shapes)
        if sig == self._sig:
            return
        need = C.c_size_t()
        with torch.cuda.device(self.device):
            check(lib.irn_walk_configure(self._ctx, len(sig), i32_array([s[0] for s in sig]),
                                         i32_array([s[1] for s in sig]), i32_array([s[2] for s in sig]),
                                         C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        self._ws_bytes = need.value
        self._sig = sig

    @property
    def workspace_bytes(self):
        return self._ws_bytes

    def __call__(self, edges, cams, beta=10, exp_times=8, inst_maps=None, k_inst=None, outs=None, n_sweeps=None):
        n = len(edges)
        if n_sweeps is None:
            n_sweeps = 2 ** int(exp_times)
        es, cs, shapes = [], [], []
        for i in range(n):
            _need_cuda(edges[i], "edge")
            _need_cuda(cams[i], "cam")
            c = cams[i].reshape((-1,) + tuple(cams[i].shape[-2:])).contiguous().float()
            h, w = c.shape[-2:]
            e = edges[i].reshape(h, w).contiguous().float()
            k = 1
            if inst_maps is not None and inst_maps[i] is not None:
                k = int(k_inst[i])
            es.append(e)
            cs.append(c)
            shapes.append((h, w, c.shape[0] * k))
        self.configure(shapes)
        if outs is None:
            outs = [torch.empty((s[2], 1, s[0], s[1]), device=self.device, dtype=torch.float32) for s in shapes]
        im_ptrs, ks = None, None
        keep = []
        if inst_maps is not None:
            ims = [None if m is None else m.reshape(shapes[i][0], shapes[i][1]).to(torch.int32).contiguous()
                   for i, m in enumerate(inst_maps)]
            keep = ims
            im_ptrs = ptr_array([None if m is None else m.data_ptr() for m in ims])
            ks = i32_array([1 if (k_inst is None or k_inst[i] is None) else k_inst[i] for i in range(n)])
        with torch.cuda.device(self.device):
            check(lib.irn_walk_run(self._ctx, ptr_array([e.data_ptr() for e in es]),
                                   ptr_array([c.data_ptr() for c in cs]), im_ptrs, ks,
                                   ptr_array([o.data_ptr() for o in outs]), float(beta), int(n_sweeps),
                                   self._ws.data_ptr(), self._ws.numel(), _stream()))
        # the edge / CAM / cluster-map tensors must outlive the run AND a possible re-run by `sync()` (which reads them
        # again, walk.hip x0_kernel): callers routinely drop theirs right after the call, and the caching allocator would
        # hand the blocks to the next batch's uploads
        self._live = (es, cs, keep)
        if not self._poll_pinned and self.radius == 10:
            t = self.tuning()
            if t["probe_ms"] is not None:                 # this run carried the start-up probe: remember its answer
                _save_poll_delay(self.device, t["poll_delay"])
                self._poll_pinned = True
        return outs

    def export_weights(self, image, n_dirs):
        """(weights [|S|,h,w] in raster direction order, inv_deg fp64 [h,w]) of the last run."""
        h, w, _ = self._sig[image]
        wt = torch.empty((n_dirs, h, w), device=self.device, dtype=torch.float32)
        dg = torch.empty((h, w), device=self.device, dtype=torch.float64)
        with torch.cuda.device(self.device):
            check(lib.irn_walk_export_weights(self._ctx, int(image), wt.data_ptr(), dg.data_ptr(),
                                              self._ws.data_ptr(), _stream()))
        return wt, dg


_WALKERS = {}


def _walker(device, radius):
    key = (torch.device(device).index, int(radius))
    w = _WALKERS.get(key)
    if w is None:
        w = _WALKERS[key] = RandomWalk(radius, device)
    return w


def propagate_to_edge(x, edge, radius=5, beta=10, exp_times=8):
    """Drop-in for reference misc/indexing.py:141-165.

    x: [C,h,w] or [C,K,h,w] GPU tensor, edge: [1,h,w]; returns [C' ,1,h,w] with C' = prod of the
    leading dims of x:  (x * (1-edge)) . T^(2^exp_times), T the column-normalised beta-powered
    path-max affinity of `edge`."""
    _need_cuda(x, "x")
    _need_cuda(edge, "edge")
    beta = float(beta)                 # run_sample.py passes CLI strings for --beta/--exp_times
    exp_times = int(exp_times)
    walker = _walker(x.device, radius)
    out = walker([edge], [x], beta=beta, exp_times=exp_times)[0]
    walker.sync()          # the reference's call is synchronous too; a launch that gave up is re-run on the streaming sweeps
    return out
