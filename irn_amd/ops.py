"""Host-side wrappers of the label epilogue and instance front-end kernels in libirn_hip.so.

Reference functions mirrored (names kept where the reference has one):
    find_centroids_with_refinement(displacement, iterations=300)   step/make_ins_seg_labels.py:18-56
    cluster_centroids(centroids, displacement, thres=2.5)          step/make_ins_seg_labels.py:58-75
    label4(mask)                = skimage.measure.label(mask, connectivity=1, background=0)  (:66,:92)
    label_epilogue(...)         = step/make_sem_seg_labels.py:43-49, step/make_ins_seg_labels.py:137-145
    detect_instance(...)        = step/make_ins_seg_labels.py:82-105
    bicubic_resize(img, size)   = misc/imutils.py:8-17 pil_resize(img, size, order=3)
    msf_pack(img, scales)       = voc12/dataloader.py:191-201 (rescale, normalise, CHW, flip pair)
GPU tensors in, GPU tensors out; no CPU fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import check, i32_array, lib, ptr_array


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise ValueError("%s must be a GPU tensor: the HIP path has no CPU fallback" % what)


def label_epilogue(rws, out_sizes, bg_thres, keys=None, want_labels=True, want_argmax=False, want_rw_up=False, packed=False):
    """Batched x4-upsample / normalise / background / argmax.

    rws[i]: GPU fp32 [C,1,h,w] (or [C,h,w]); out_sizes[i] = (H, W) with H <= 4h, W <= 4w;
    keys[i]: GPU int64 [C] (0-based class ids, the CAM dict's ``keys``) when labels are wanted.
    Returns dict of lists: 'labels' uint8 [H,W] (0 = background, else key+1), 'argmax' int32 [H,W],
    'rw_up' fp32 [C,H,W] (divided by the global max) — each present only if requested.  With `packed` the label maps are
    views of ONE uint8 buffer, returned as 'labels_flat' (a step brings a whole batch to the host with one copy)."""
    n = len(rws)
    dev = rws[0].device
    rs, cs, hs, ws, ohs, ows = [], [], [], [], [], []
    for i in range(n):
        _need_cuda(rws[i], "rw")
        r = rws[i].reshape((-1,) + tuple(rws[i].shape[-2:])).contiguous().float()
        rs.append(r)
        cs.append(r.shape[0]); hs.append(r.shape[1]); ws.append(r.shape[2])
        ohs.append(int(out_sizes[i][0])); ows.append(int(out_sizes[i][1]))
    labels = flat = None
    if want_labels and packed:
        offs = np.concatenate([[0], np.cumsum([ohs[i] * ows[i] for i in range(n)])])
        flat = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
        labels = [flat[int(offs[i]):int(offs[i + 1])].view(ohs[i], ows[i]) for i in range(n)]
    elif want_labels:
        labels = [torch.empty((ohs[i], ows[i]), dtype=torch.uint8, device=dev) for i in range(n)]
    argmax = [torch.empty((ohs[i], ows[i]), dtype=torch.int32, device=dev) for i in range(n)] if want_argmax else None
    rw_up = [torch.empty((cs[i], ohs[i], ows[i]), dtype=torch.float32, device=dev) for i in range(n)] if want_rw_up else None
    ks = None
    if want_labels:
        if keys is None:
            raise ValueError("labels need keys")
        ks = [torch.as_tensor(k, device=dev).to(torch.int64).contiguous() for k in keys]
    scratch = torch.empty(max(n, 64), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.irn_label_epilogue(
            n, ptr_array([r.data_ptr() for r in rs]), i32_array(cs), i32_array(hs), i32_array(ws),
            i32_array(ohs), i32_array(ows), float(bg_thres),
            None if ks is None else ptr_array([k.data_ptr() for k in ks]),
            None if labels is None else ptr_array([t.data_ptr() for t in labels]),
            None if argmax is None else ptr_array([t.data_ptr() for t in argmax]),
            None if rw_up is None else ptr_array([t.data_ptr() for t in rw_up]),
            scratch.data_ptr(), _stream()))
    out = {}
    if want_labels:
        out["labels"] = labels
        if flat is not None:
            out["labels_flat"] = flat
    if want_argmax:
        out["argmax"] = argmax
    if want_rw_up:
        out["rw_up"] = rw_up
    return out


def cam_merge(outputs, size, label):
    """Multi-scale CAM merge of reference step/make_cam.py:38-52 (irn_cam_merge).

    outputs: list of GPU fp32 [n_classes, hs, ws] (one per scale); size = (H, W) of the image;
    label: [n_classes] multi-hot image-level label (pass it as a HOST tensor to keep the call asynchronous).  Returns (keys int64 [K] on the same device,
    cam fp32 [K, ceil(H/4), ceil(W/4)], high_res fp32 [K, H, W]), each channel divided by its max + 1e-5."""
    for o in outputs:
        _need_cuda(o, "CAM output")
    dev = outputs[0].device
    outs = [o.contiguous().float() for o in outputs]
    n_cls = outs[0].shape[0]
    H, W = int(size[0]), int(size[1])
    label = torch.as_tensor(label)
    if label.is_cuda:
        keys = torch.nonzero(label.to(dev))[:, 0].to(torch.int64).contiguous()          # synchronises (data-dependent size)
    else:
        # the loader hands the image-level label over on the host: the present classes are found there and only the
        # key list travels, so the call never waits for the device (a device-side nonzero needs its result size)
        keys = torch.nonzero(label)[:, 0].to(torch.int64).contiguous().to(dev, non_blocking=True)
    k = int(keys.numel())
    cam = torch.empty((k, (H - 1) // 4 + 1, (W - 1) // 4 + 1), dtype=torch.float32, device=dev)
    hi = torch.empty((k, H, W), dtype=torch.float32, device=dev)
    scratch = torch.empty(2 * k, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.irn_cam_merge(len(outs), ptr_array([o.data_ptr() for o in outs]), i32_array([o.shape[1] for o in outs]),
                                i32_array([o.shape[2] for o in outs]), n_cls, keys.data_ptr(), k, H, W, cam.data_ptr(),
                                hi.data_ptr(), scratch.data_ptr(), _stream()))
    return keys, cam, hi


def bn_act_(x, scale, shift, residual=None, relu=True, residual_affine=None):
    """Inference batch norm (+ residual) (+ ReLU) in one pass, IN PLACE on a convolution's output (irn_bn_act):
    ``x = act(x * scale[c] + shift[c] (+ r))`` — the elementwise tail of reference net/resnet50.py:34-54.  ``r`` is the
    residual, or ``residual * rs[c] + rb[c]`` with ``residual_affine = (rs, rb)`` (the projection shortcut's batch norm).
    x, residual: GPU fp32 [N, C, ...] contiguous; scale, shift, rs, rb: GPU fp32 [C].  Returns x."""
    _need_cuda(x, "x")
    nhwc = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    if x.dtype != torch.float32 or not (x.is_contiguous() or nhwc) or x.dim() < 2:
        raise ValueError("bn_act_: x must be a contiguous (or channels-last) fp32 [N, C, ...] tensor, got %s %s" % (x.dtype, tuple(x.shape)))
    n_ch = int(x.shape[1])
    if nhwc and n_ch % 4:
        raise ValueError("bn_act_: a channels-last tensor needs a multiple of 4 channels, got %d" % n_ch)
    consts = [("scale", scale), ("shift", shift)]
    if residual_affine is not None:
        if residual is None:
            raise ValueError("bn_act_: residual_affine without a residual")
        consts += [("residual scale", residual_affine[0]), ("residual shift", residual_affine[1])]
    for name, t in consts:
        if t.device != x.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n_ch:
            raise ValueError("bn_act_: %s must be a contiguous fp32 [%d] tensor on %s" % (name, n_ch, x.device))
    if residual is not None and (residual.shape != x.shape or residual.dtype != torch.float32 or residual.device != x.device
                                 or not (residual.is_contiguous(memory_format=torch.channels_last) if nhwc else residual.is_contiguous())):
        raise ValueError("bn_act_: residual must match x (shape %s, fp32, same memory format, same device)" % (tuple(x.shape),))
    n_img = int(x.shape[0])
    if nhwc:
        rs, rb = (None, None) if residual_affine is None else (residual_affine[0].data_ptr(), residual_affine[1].data_ptr())
        px = int(x.shape[2]) * int(x.shape[3])
        per = max(1, (2 ** 31 - 1) // max(1, n_ch * px))
        with torch.cuda.device(x.device):
            for i in range(0, n_img, per):
                xi = x[i:i + per]
                ri = None if residual is None else residual[i:i + per]
                check(lib.irn_bn_act_nhwc(xi.data_ptr(), None if ri is None else ri.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                          rs, rb, int(xi.shape[0]) * px, n_ch, 1 if relu else 0, _stream()))
        return x
    plane = x[0, 0].numel() if n_img else 0
    rs, rb = (None, None) if residual_affine is None else (residual_affine[0].data_ptr(), residual_affine[1].data_ptr())
    # the entry point takes at most 2^31 - 1 elements: larger batches go image group by image group
    per = max(1, (2 ** 31 - 1) // max(1, n_ch * plane))
    with torch.cuda.device(x.device):
        for i in range(0, n_img, per):
            xi = x[i:i + per]
            ri = None if residual is None else residual[i:i + per]
            check(lib.irn_bn_act(xi.data_ptr(), None if ri is None else ri.data_ptr(), scale.data_ptr(), shift.data_ptr(), rs, rb,
                                 int(xi.shape[0]), n_ch, plane, 1 if relu else 0, _stream()))
    return x


_GEMM_WS = {}          # (device index, stream) -> workspace tensor of the fused 1x1 convolutions (stream-ordered use)
_GEMM_NALGOS = {}      # problems with a non-zero table rank -> length of hipBLASLt's heuristic list in this process
_GEMM_RANK_WARNED = False
_GEMM_RANKS = None     # (m, cin, cout, bias, residual, relu) -> rank in hipBLASLt's heuristic list, measured once per device


def gemm_ranks():
    """The shipped table `irn_amd/data/gemm/<device>-hip<version>.json` (written by tools/conv1x1_tune.py on a GPU box):
    for the problems listed, which entry of hipBLASLt's heuristic list was fastest.  Problems not listed use entry 0.  The
    table is data, not a timing: every process picks the same kernel for the same problem."""
    global _GEMM_RANKS
    if _GEMM_RANKS is None:
        ranks = {}
        try:
            import json
            import os
            from .step import _common
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gemm", _common.miopen_cache_key() + ".json")
            if os.path.exists(path) and os.environ.get("IRN_GEMM_TABLE", "1") != "0":
                ranks = {tuple(int(v) for v in k.split(",")): int(r) for k, r in json.load(open(path))["ranks"].items()}
        except Exception:
            ranks = {}
        _GEMM_RANKS = ranks
    return _GEMM_RANKS


def conv1x1_nhwc(x, weight, bias=None, residual=None, relu=False, out=None, algo_rank=None):
    """1x1 convolution (stride 1) of a channels-last activation with bias, residual add and ReLU in the GEMM's epilogue
    (irn_conv1x1_nhwc; reference net/resnet50.py:34-54 with FixedBatchNorm folded into `weight` / `bias`).
    x: GPU fp32 [N, cin, H, W] in torch.channels_last; weight: GPU fp32 [cout, cin] (or [cout, cin, 1, 1]) contiguous;
    bias: GPU fp32 [cout] or None; residual: like the result, or None; out: a channels-last [N, cout, H, W] tensor to write
    (may be `residual`), else a fresh one.  -> act(conv(x, weight) + bias (+ residual))."""
    _need_cuda(x, "x")
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("conv1x1_nhwc: x must be a channels-last fp32 [N, C, H, W] tensor, got %s %s %s" % (x.dtype, tuple(x.shape), x.stride()))
    n, cin, h, w_ = (int(v) for v in x.shape)
    cout = int(weight.shape[0])
    if weight.dtype != torch.float32 or weight.device != x.device or not weight.is_contiguous() or weight.numel() != cout * cin:
        raise ValueError("conv1x1_nhwc: weight must be a contiguous fp32 [%d-out, %d] tensor on %s" % (cout, cin, x.device))
    if bias is not None and (bias.dtype != torch.float32 or bias.device != x.device or not bias.is_contiguous() or bias.numel() != cout):
        raise ValueError("conv1x1_nhwc: bias must be a contiguous fp32 [%d] tensor on %s" % (cout, x.device))
    shape = (n, cout, h, w_)
    for name, t in (("residual", residual), ("out", out)):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or t.device != x.device
                              or not t.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("conv1x1_nhwc: %s must be a channels-last fp32 %s tensor on %s" % (name, shape, x.device))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    m = n * h * w_
    if m == 0:
        return out
    # one workspace per (device, stream): GEMMs enqueued on different streams may overlap on the device
    dev = (x.device.index if x.device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(x.device).cuda_stream)
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = _GEMM_WS[dev] = torch.empty(int(lib.irn_conv1x1_workspace_bytes()), dtype=torch.uint8, device=x.device)
    if algo_rank is None:
        prob = (m, cin, cout, int(bias is not None), int(residual is not None), int(bool(relu)))
        algo_rank = gemm_ranks().get(prob, 0)
        if algo_rank:
            # the table is keyed by architecture / CU count / HIP version only: another hipBLASLt build or workspace size may
            # offer a shorter list — then the first pick, with one warning, instead of an error in the middle of a forward
            n_algos = _GEMM_NALGOS.get(prob)
            if n_algos is None:
                with torch.cuda.device(x.device):
                    n_algos = _GEMM_NALGOS[prob] = conv1x1_algo_count(*prob)
            if algo_rank >= n_algos:
                global _GEMM_RANK_WARNED
                if not _GEMM_RANK_WARNED:
                    _GEMM_RANK_WARNED = True
                    import warnings
                    warnings.warn("irn_amd: the shipped GEMM rank table names entry %d for the 1x1 convolution m=%d cin=%d cout=%d but "
                                  "hipBLASLt offers %d here (another library build?): using its first pick for such problems; "
                                  "tools/conv1x1_tune.py rewrites the table" % (algo_rank, m, cin, cout, n_algos), RuntimeWarning)
                algo_rank = 0
    with torch.cuda.device(x.device):
        check(lib.irn_conv1x1_nhwc(x.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                   None if residual is None else residual.data_ptr(), out.data_ptr(), m, cin, cout,
                                   1 if relu else 0, int(algo_rank), ws.data_ptr(), ws.numel(), _stream()))
    return out


# ---- split-precision 1x1 convolutions (round 6): fp16 hi/lo operands, fp32 accumulation ----
_SPLIT_FLAGS = {}      # device index -> uint32 [1] device tensor: bit 0 = an activation left fp16's range in irn_split16


def _split_flag(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    f = _SPLIT_FLAGS.get(idx)
    if f is None:
        f = _SPLIT_FLAGS[idx] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


def split_overflowed(reset=True):
    """Did any activation handed to `split16` since the last call exceed fp16's range (|x| > 65504) or hold a NaN, on any
    device of this process?  Reads the device flags (synchronises).  The steps check it at the end of every step and raise:
    such a network needs IRN_SPLIT_GEMM=0 (trained ResNet-50 activations stay below a few hundred)."""
    bad = False
    for f in _SPLIT_FLAGS.values():
        if int(f.item()) != 0:
            bad = True
            if reset:
                f.zero_()
    return bad


def split16(x, scale=None, shift=None, relu=False):
    """fp32 channels-last activation [N, C, H, W] -> fp16 [N*H*W, 3C] = [hi | hi | lo'] per pixel (irn_split16), the A operand
    of `gemm16_nhwc`; with `scale` / `shift` (fp32 [C]) the inference batch norm (+ ReLU) of the convolution that produced
    `x` is applied on the way (reference net/resnet50.py:40-42) and `x` is never written back."""
    _need_cuda(x, "x")
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("split16: x must be a channels-last fp32 [N, C, H, W] tensor, got %s %s %s" % (x.dtype, tuple(x.shape), x.stride()))
    n, c, h, w_ = (int(v) for v in x.shape)
    if c % 8:
        raise ValueError("split16: %d channels; a multiple of 8 is needed" % c)
    m = n * h * w_
    out = torch.empty((m, 3 * c), dtype=torch.float16, device=x.device)
    if m == 0:
        return out
    for name, t in (("scale", scale), ("shift", shift)):
        if t is not None and (t.dtype != torch.float32 or t.device != x.device or not t.is_contiguous() or t.numel() != c):
            raise ValueError("split16: %s must be a contiguous fp32 [%d] tensor on %s" % (name, c, x.device))
    with torch.cuda.device(x.device):
        per = max(1, (2 ** 31 - 1) // c)
        for i in range(0, m, per):                       # at most 2^31 - 1 elements per call
            rows = min(per, m - i)
            check(lib.irn_split16(x.data_ptr() + 4 * i * c, None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(),
                                  1 if relu else 0, out.data_ptr() + 2 * i * 3 * c, rows, c, _split_flag(x.device).data_ptr(), _stream()))
    return out


def gemm16_algo_count(m, k, cout, bias, residual, relu):
    """How many kernels hipBLASLt's heuristic offers for the split-precision problem (tools/gemm16_tune.py times each once)."""
    n = C.c_int(0)
    check(lib.irn_gemm16_algo_count(int(m), int(k), int(cout), int(bool(bias)), int(bool(residual)), int(bool(relu)),
                                    int(lib.irn_conv1x1_workspace_bytes()), C.byref(n)))
    return n.value


_GEMM_RANKS16 = None


def gemm_ranks16():
    """`ranks16` of the shipped table (gemm_ranks): (m, k, cout, bias, residual, relu) -> entry of hipBLASLt's list for the fp16
    operand problems, measured once per device by tools/gemm16_tune.py; unlisted problems use entry 0."""
    global _GEMM_RANKS16
    if _GEMM_RANKS16 is None:
        ranks = {}
        try:
            import json
            import os
            from .step import _common
            path = os.path.join(_common.gemm_table_root(), _common.miopen_cache_key() + ".json")
            if os.path.exists(path) and os.environ.get("IRN_GEMM_TABLE", "1") != "0":
                ranks = {tuple(int(v) for v in k.split(",")): int(r) for k, r in json.load(open(path)).get("ranks16", {}).items()}
        except Exception:
            ranks = {}
        _GEMM_RANKS16 = ranks
    return _GEMM_RANKS16


_GEMM_RANKS3X3 = None


def gemm_ranks3x3():
    """`ranks3x3` of the shipped table: (rows of the bordered operand, cin, cout) -> entry of hipBLASLt's list for the row-fused 3x3
    split convolution (irn_conv3x3_split_gemm), measured by tools/gemm16_tune.py; unlisted problems use entry 0."""
    global _GEMM_RANKS3X3
    if _GEMM_RANKS3X3 is None:
        ranks = {}
        try:
            import json
            import os
            from .step import _common
            path = os.path.join(_common.gemm_table_root(), _common.miopen_cache_key() + ".json")
            if os.path.exists(path) and os.environ.get("IRN_GEMM_TABLE", "1") != "0":
                ranks = {tuple(int(v) for v in k.split(",")): int(r) for k, r in json.load(open(path)).get("ranks3x3", {}).items()}
        except Exception:
            ranks = {}
        _GEMM_RANKS3X3 = ranks
    return _GEMM_RANKS3X3


def gemm16_nhwc(a16, b16, shape, bias=None, residual=None, relu=False, alpha=1.0, out=None, algo_rank=None):
    """act(alpha * a16 . b16^T + bias (+ residual)) as a channels-last fp32 [N, cout, H, W] tensor of `shape` (irn_gemm16_nhwc):
    a16 fp16 [N*H*W, k] from `split16`, b16 fp16 [cout, k] = [w_hi | w_lo | w_hi 2^-11] of the weight scaled by 1 / alpha."""
    _need_cuda(a16, "a16")
    n, cout, h, w_ = (int(v) for v in shape)
    m, k = int(a16.shape[0]), int(a16.shape[1])
    if a16.dtype != torch.float16 or not a16.is_contiguous() or m != n * h * w_:
        raise ValueError("gemm16_nhwc: a16 must be a contiguous fp16 [%d, k] matrix" % (n * h * w_))
    if b16.dtype != torch.float16 or b16.device != a16.device or not b16.is_contiguous() or tuple(b16.shape) != (cout, k):
        raise ValueError("gemm16_nhwc: b16 must be a contiguous fp16 [%d, %d] matrix on %s" % (cout, k, a16.device))
    if bias is not None and (bias.dtype != torch.float32 or bias.device != a16.device or not bias.is_contiguous() or bias.numel() != cout):
        raise ValueError("gemm16_nhwc: bias must be a contiguous fp32 [%d] tensor on %s" % (cout, a16.device))
    for name, t in (("residual", residual), ("out", out)):
        if t is not None and (tuple(t.shape) != (n, cout, h, w_) or t.dtype != torch.float32 or t.device != a16.device
                              or not t.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("gemm16_nhwc: %s must be a channels-last fp32 %s tensor on %s" % (name, (n, cout, h, w_), a16.device))
    if out is None:
        out = torch.empty((n, cout, h, w_), dtype=torch.float32, device=a16.device, memory_format=torch.channels_last)
    if m == 0:
        return out
    dev = (a16.device.index if a16.device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(a16.device).cuda_stream)
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = _GEMM_WS[dev] = torch.empty(int(lib.irn_conv1x1_workspace_bytes()), dtype=torch.uint8, device=a16.device)
    if algo_rank is None:
        prob = (m, k, cout, int(bias is not None), int(residual is not None), int(bool(relu)))
        algo_rank = gemm_ranks16().get(prob, 0)
        if algo_rank:
            n_algos = _GEMM_NALGOS.get(("f16",) + prob)
            if n_algos is None:
                with torch.cuda.device(a16.device):
                    n_algos = _GEMM_NALGOS[("f16",) + prob] = gemm16_algo_count(*prob)
            if algo_rank >= n_algos:          # another hipBLASLt build than the one the table was measured with
                algo_rank = 0
    with torch.cuda.device(a16.device):
        check(lib.irn_gemm16_nhwc(a16.data_ptr(), b16.data_ptr(), None if bias is None else bias.data_ptr(),
                                  None if residual is None else residual.data_ptr(), out.data_ptr(), m, k, cout,
                                  1 if relu else 0, float(alpha), int(algo_rank), ws.data_ptr(), ws.numel(), _stream()))
    return out


def split_weight(w64, p=None):
    """Weight [cout, cin] (float64, batch norm folded in) -> (b16 fp16 [cout, 3 cin] = [w_hi | w_lo | w_hi 2^-11] of w 2^p, alpha =
    2^-p): p puts the largest |w 2^p| into [2^13, 2^14), so that w_lo = fp16(w 2^p - w_hi) <= 8 stays a normal fp16 number for
    every weight above 2^-17 of the largest (smaller ones contribute below the fp32 rounding of the sum).  `p` given: the
    exponent of a larger tensor this one is a slice of (the taps of a 3x3 weight share one)."""
    import math
    w64 = w64.detach().double()
    if p is None:
        top = float(w64.abs().max())
        p = 13 - int(math.floor(math.log2(top))) if top > 0 else 0
    ws = w64 * (2.0 ** p)
    hi = ws.to(torch.float16)
    lo = (ws - hi.double()).to(torch.float16)
    hi_s = (hi.double() * 2.0 ** -11).to(torch.float16)
    return torch.cat([hi, lo, hi_s], dim=1).contiguous(), 2.0 ** -p


# three GEMMs over 9 cin (row-fused) instead of nine over 3 cin: IRN_CONV3X3_ROW_FUSED=0 keeps the nine
CONV3X3_ROW_FUSED = os.environ.get("IRN_CONV3X3_ROW_FUSED", "1") != "0"


def split_weight_3x3(w64):
    """3x3 weight [cout, cin, 3, 3] (float64) -> (fp16 [9, cout, 3 cin]: one `split_weight` operand per tap (ky, kx) in raster
    order, one common exponent, alpha)."""
    import math
    w64 = w64.detach().double()
    top = float(w64.abs().max())
    p = 13 - int(math.floor(math.log2(top))) if top > 0 else 0
    taps = [split_weight(w64[:, :, ky, kx], p)[0] for ky in range(3) for kx in range(3)]
    if CONV3X3_ROW_FUSED:      # [3, cout, 9 cin]: the three taps of a kernel row side by side (irn_conv3x3_split_gemm row_fused)
        return torch.stack([torch.cat(taps[3 * ky:3 * ky + 3], dim=1) for ky in range(3)]).contiguous(), 2.0 ** -p
    return torch.stack(taps).contiguous(), 2.0 ** -p


def split16_pad(x, shape, scale=None, shift=None, relu=False, in_padded=False, out=None):
    """`split16` between a dense map and its zero-bordered form (irn_split16_pad).  shape = (n, c, h, w) of the DENSE map.
    in_padded: `x` is the bordered fp32 form [n (h+2)(w+2), c] (a `conv3x3_split` result), the result is the dense fp16
    [n h w, 3c]; `out` given (a bordered fp16 buffer's interior view, borders already zero): `x` is a dense channels-last fp32
    tensor and the split goes into the bordered form, zero border rows included."""
    n, c, h, w_ = (int(v) for v in shape)
    _need_cuda(x, "x")
    if c % 8:
        raise ValueError("split16_pad: %d channels; a multiple of 8 is needed" % c)
    if in_padded:
        if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != n * (h + 2) * (w_ + 2) * c:
            raise ValueError("split16_pad: x must be the contiguous bordered fp32 form of a %s map" % (shape,))
    elif x.dtype != torch.float32 or tuple(x.shape) != (n, c, h, w_) or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("split16_pad: x must be a channels-last fp32 %s tensor" % (shape,))
    out_padded = out is not None
    if out is None:
        out = torch.empty((n * h * w_, 3 * c), dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or not out.is_contiguous() or out.numel() < n * (h + 2) * (w_ + 2) * 3 * c:
        raise ValueError("split16_pad: out must be a contiguous fp16 buffer of the bordered form (its border rows are written too)")
    if n * h * w_ == 0:
        return out
    with torch.cuda.device(x.device):
        check(lib.irn_split16_pad(x.data_ptr(), None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(),
                                  1 if relu else 0, out.data_ptr(), n, h, w_, c, 1 if in_padded else 0, 1 if out_padded else 0,
                                  _split_flag(x.device).data_ptr(), _stream()))
    return out


_ROW_FUSED_REFUSED = False


def conv3x3_split(x, w16, alpha, algo_rank=None):
    """3x3 / stride 1 / pad 1 convolution (no bias) of a channels-last fp32 activation [N, C, H, W] in the split-precision form,
    WITHOUT materialising an im2col operand: the activation is split once into a zero-bordered fp16 matrix [N (H+2)(W+2), 3C]
    (irn_split16_pad writes the borders too); there tap (ky, kx) is the same matrix shifted by (ky-1)(W+2) + (kx-1) rows, so the
    convolution is nine fp16 GEMMs accumulating in fp32 in a fixed order (irn_conv3x3_split_gemm, one call).  w16 =
    `split_weight_3x3` fp16 [9, cout, 3C].  -> the result in the bordered fp32 form [N (H+2)(W+2), cout] (border rows hold
    garbage; `split16_pad(..., in_padded=True)` reads the interior).  Reference: conv2 of Bottleneck.forward, net/resnet50.py:40."""
    _need_cuda(x, "x")
    n, c, h, w_ = (int(v) for v in x.shape)
    cout = int(w16.shape[1])
    fused = int(w16.shape[0]) == 3
    if tuple(w16.shape) not in ((9, cout, 3 * c), (3, cout, 9 * c)) or w16.dtype != torch.float16 or not w16.is_contiguous() or w16.device != x.device:
        raise ValueError("conv3x3_split: w16 must be a contiguous fp16 [9, cout, %d] or [3, cout, %d] tensor on %s" % (3 * c, 9 * c, x.device))
    m_pad, guard = n * (h + 2) * (w_ + 2), w_ + 3
    a_buf = torch.empty((m_pad + 2 * guard, 3 * c), dtype=torch.float16, device=x.device)        # guard rows: valid memory, any content
    out = torch.empty((m_pad, cout), dtype=torch.float32, device=x.device)
    if m_pad == 0:
        return out
    split16_pad(x, (n, c, h, w_), out=a_buf[guard:])
    dev = (x.device.index if x.device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(x.device).cuda_stream)
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = _GEMM_WS[dev] = torch.empty(int(lib.irn_conv1x1_workspace_bytes()), dtype=torch.uint8, device=x.device)
    rank = algo_rank
    if rank is None:
        rank = gemm_ranks3x3().get((m_pad, c, cout), 0) if fused else gemm_ranks16().get((m_pad, 3 * c, cout, 0, 1, 0), 0)
    global _ROW_FUSED_REFUSED
    with torch.cuda.device(x.device):
        if fused and not _ROW_FUSED_REFUSED:
            rc = lib.irn_conv3x3_split_gemm(a_buf[guard:].data_ptr(), w16.data_ptr(), out.data_ptr(), n, h, w_, c, cout, float(alpha), 1, int(rank),
                                            ws.data_ptr(), ws.numel(), _stream())
            if rc == 0:
                return out
            # this hipBLASLt build does not take an operand with overlapping rows: the nine-GEMM form computes the same sums
            _ROW_FUSED_REFUSED = True
            import warnings
            warnings.warn("irn_amd: hipBLASLt refused the row-fused 3x3 operand (%s); using nine GEMMs per 3x3 convolution (~7 %% slower "
                          "backbones)" % lib.irn_last_error().decode(errors="replace"), RuntimeWarning)
        if fused:                                   # [3, cout, 9c] -> [9, cout, 3c]: the same taps, one per GEMM
            w16 = w16.view(3, cout, 3, 3 * c).permute(0, 2, 1, 3).reshape(9, cout, 3 * c).contiguous()
            rank = 0
        check(lib.irn_conv3x3_split_gemm(a_buf[guard:].data_ptr(), w16.data_ptr(), out.data_ptr(), n, h, w_, c, cout, float(alpha), 0, int(rank),
                                         ws.data_ptr(), ws.numel(), _stream()))
    return out


def conv1x1_algo_count(m, cin, cout, bias, residual, relu):
    """How many kernels hipBLASLt's heuristic offers for the problem (tools/conv1x1_tune.py times each of them once)."""
    n = C.c_int(0)
    check(lib.irn_conv1x1_algo_count(int(m), int(cin), int(cout), int(bool(bias)), int(bool(residual)), int(bool(relu)),
                                     int(lib.irn_conv1x1_workspace_bytes()), C.byref(n)))
    return n.value


def _need_f32_contig(t, what, min_dim):
    _need_cuda(t, what)
    if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() < min_dim:
        raise ValueError("%s must be a contiguous fp32 tensor of >= %d dimensions, got %s %s" % (what, min_dim, t.dtype, tuple(t.shape)))


def stem_pool(x, scale, shift):
    """Batch norm + ReLU + 3x3 / stride 2 / pad 1 max pool of the trunk's stem in one pass (irn_stem_pool; reference
    net/resnet50.py:94-97).  x: GPU fp32 [N, C, H, W] (conv1's output, left untouched) -> [N, C, (H-1)//2+1, (W-1)//2+1]."""
    _need_f32_contig(x, "stem_pool: x", 4)
    n, c, h, w = (int(v) for v in x.shape)
    for name, t in (("scale", scale), ("shift", shift)):
        if t.device != x.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != c:
            raise ValueError("stem_pool: %s must be a contiguous fp32 [%d] tensor on %s" % (name, c, x.device))
    out = torch.empty((n, c, (h - 1) // 2 + 1 if h else 0, (w - 1) // 2 + 1 if w else 0), dtype=torch.float32, device=x.device)
    if out.numel():
        with torch.cuda.device(x.device):
            check(lib.irn_stem_pool(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), n, c, h, w, out.data_ptr(), _stream()))
    return out


def upsample_bilinear(x, factor, relu=False):
    """nn.Upsample(scale_factor=factor, mode='bilinear', align_corners=False) (+ ReLU) of the IRNet heads in one pass
    (irn_upsample_bilinear; reference net/resnet50_irn.py:36-48, :72-84).  x: GPU fp32 [..., h, w] -> [..., h*factor, w*factor]."""
    _need_f32_contig(x, "upsample_bilinear: x", 2)
    if int(factor) != factor or not 1 <= factor <= 64:
        raise ValueError("upsample_bilinear: integer factor in 1..64 expected, got %r" % (factor,))
    factor = int(factor)
    h, w = int(x.shape[-2]), int(x.shape[-1])
    out = torch.empty(tuple(x.shape[:-2]) + (h * factor, w * factor), dtype=torch.float32, device=x.device)
    if out.numel():
        with torch.cuda.device(x.device):
            check(lib.irn_upsample_bilinear(x.data_ptr(), x.numel() // (h * w), h, w, factor, 1 if relu else 0, out.data_ptr(),
                                            _stream()))
    return out


_LUTS = {}


def rescale_size(h, w, scale):
    """misc/imutils.py:19-22: target size of pil_rescale (np.round: half to even)."""
    return int(np.round(h * scale)), int(np.round(w * scale))


def bicubic_plan(in_size, out_size):
    """Host-only: Pillow's fixed-point bicubic tap table of one axis -> (lo [out], count [out], weights [out, ksize])."""
    ks = C.c_int32()
    check(lib.irn_bicubic_plan(int(in_size), int(out_size), C.byref(ks), None, None, None, 0))
    lo = np.empty(out_size, np.int32)
    cnt = np.empty(out_size, np.int32)
    k = np.empty((out_size, ks.value), np.int32)
    as_p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    check(lib.irn_bicubic_plan(int(in_size), int(out_size), C.byref(ks), as_p(lo), as_p(cnt), as_p(k), k.size))
    return lo, cnt, k


def bicubic_resize(img, size):
    """GPU uint8 [H,W,C] (C in 1,3,4) or [H,W] -> GPU uint8 resized to size=(h,w); bit-identical to
    np.asarray(Image.fromarray(img).resize(size[::-1], Image.BICUBIC)) (misc/imutils.py:8-17)."""
    _need_cuda(img, "img")
    if img.dtype != torch.uint8:
        raise ValueError("bicubic_resize: uint8 image expected (Pillow's 8-bit path)")
    squeeze = img.dim() == 2
    src = (img[..., None] if squeeze else img).contiguous()
    h, w, ch = src.shape
    hs, ws = int(size[0]), int(size[1])
    if (hs, ws) == (h, w):
        return img
    out = torch.empty((hs, ws, ch), dtype=torch.uint8, device=src.device)
    scratch = torch.empty(max(lib.irn_bicubic_scratch_bytes(h, w, hs, ws, ch), 1), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        check(lib.irn_bicubic_resize_u8(src.data_ptr(), h, w, ch, hs, ws, out.data_ptr(), scratch.data_ptr(), _stream()))
    return out[..., 0] if squeeze else out


def normalize_lut(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """fp32 [3,256]: TorchvisionNormalize (voc12/dataloader.py:65-78) of every byte value, computed in float64
    and stored as float32 exactly like the reference's `proc_img[..., c] = (imgarr[..., c] / 255. - mean) / std`."""
    v = np.arange(256, dtype=np.uint8)
    return np.stack([((v / 255. - mean[c]) / std[c]).astype(np.float32) for c in range(3)])


def msf_pack(img, scales, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """GPU uint8 [H,W,3] -> list over scales of GPU fp32 [2,3,Hs,Ws]: the `img` entry of a
    VOC12ClassificationDatasetMSF item (voc12/dataloader.py:191-201), bit-identical to the PIL/numpy path."""
    _need_cuda(img, "img")
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise ValueError("msf_pack: uint8 [H,W,3] image expected")
    src = img.contiguous()
    dev = src.device
    h, w = int(src.shape[0]), int(src.shape[1])
    sizes = [(h, w) if s == 1 else rescale_size(h, w, s) for s in scales]
    key = ("msf_lut", tuple(mean), tuple(std), str(dev))
    lut = _LUTS.get(key)
    if lut is None:
        lut = _LUTS[key] = torch.from_numpy(normalize_lut(mean, std)).to(dev).contiguous()
    outs = [torch.empty((2, 3, hs, ws), dtype=torch.float32, device=dev) for hs, ws in sizes]
    nbytes = max([lib.irn_bicubic_scratch_bytes(h, w, hs, ws, 3) for hs, ws in sizes] + [1])
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.irn_msf_pack(src.data_ptr(), h, w, len(sizes), i32_array([s[0] for s in sizes]),
                               i32_array([s[1] for s in sizes]), lut.data_ptr(), ptr_array([o.data_ptr() for o in outs]),
                               scratch.data_ptr(), _stream()))
    return outs


def find_centroids_with_refinement(displacement, iterations=300):
    """dp GPU fp32 [2,h,w] -> GPU int32 [2,h,w] (cy, cx); bit-identical to the reference's numpy
    (step/make_ins_seg_labels.py:18-56)."""
    _need_cuda(displacement, "displacement")
    dp = displacement.contiguous().float()
    _, h, w = dp.shape
    out = torch.empty((2, h, w), dtype=torch.int32, device=dp.device)
    with torch.cuda.device(dp.device):
        check(lib.irn_find_centroids(dp.data_ptr(), h, w, int(iterations), out.data_ptr(), _stream()))
    return out


def cluster_centroids(centroids, displacement, thres=2.5, as_one_hot=False):
    """-> (cluster_map GPU int32 [h,w] with values 0..K-1, K).  With ``as_one_hot`` returns the
    reference's bool [K,h,w] instead (step/make_ins_seg_labels.py:58-75)."""
    _need_cuda(centroids, "centroids")
    dp = displacement.contiguous().float()
    cen = centroids.to(torch.int32).contiguous()
    _, h, w = dp.shape
    cmap = torch.empty((h, w), dtype=torch.int32, device=dp.device)
    scratch = torch.empty(lib.irn_cluster_scratch_bytes(h, w), dtype=torch.uint8, device=dp.device)
    k = C.c_int()
    with torch.cuda.device(dp.device):
        check(lib.irn_cluster_centroids(cen.data_ptr(), dp.data_ptr(), h, w, float(thres), cmap.data_ptr(),
                                        C.byref(k), scratch.data_ptr(), _stream()))
    if as_one_hot:
        return (cmap[None] == torch.arange(k.value, device=dp.device, dtype=torch.int32)[:, None, None])
    return cmap, k.value


def find_centroids_batch(displacements, iterations=300):
    """Batched find_centroids_with_refinement: list of GPU fp32 [2,h,w] -> list of GPU int32 [2,h,w]; one launch for
    the whole batch (irn_find_centroids_batch), nothing synchronises."""
    dps = []
    for d in displacements:
        _need_cuda(d, "displacement")
        dps.append(d.contiguous().float())
    dev = dps[0].device
    outs = [torch.empty((2,) + tuple(d.shape[1:]), dtype=torch.int32, device=dev) for d in dps]
    with torch.cuda.device(dev):
        check(lib.irn_find_centroids_batch(len(dps), ptr_array([d.data_ptr() for d in dps]),
                                           i32_array([d.shape[1] for d in dps]), i32_array([d.shape[2] for d in dps]),
                                           int(iterations), ptr_array([o.data_ptr() for o in outs]), _stream()))
    return outs


def cluster_centroids_batch(centroids, displacements, thres=2.5, k_on_device=False):
    """Batched cluster_centroids: -> (list of GPU int32 [h,w] cluster maps with values 0..K_i-1, list of K_i).
    One launch sequence for the batch and ONE device-to-host transfer for all the K (irn_cluster_centroids_batch).
    `k_on_device`: return the K as the GPU int32 [n] tensor instead — nothing waits for the device, the caller reads them
    (`.cpu()`) when it needs them (the instance step enqueues a batch's front end and goes on loading the next batch)."""
    n = len(centroids)
    dps = [d.contiguous().float() for d in displacements]
    cens = []
    for c in centroids:
        _need_cuda(c, "centroids")
        cens.append(c.to(torch.int32).contiguous())
    dev = dps[0].device
    hs, ws = i32_array([d.shape[1] for d in dps]), i32_array([d.shape[2] for d in dps])
    cmaps = [torch.empty(tuple(d.shape[1:]), dtype=torch.int32, device=dev) for d in dps]
    k_dev = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = _cached("cluster_scratch", dev, lib.irn_cluster_batch_scratch_bytes(n, hs, ws), torch.uint8)
    with torch.cuda.device(dev):
        check(lib.irn_cluster_centroids_batch(n, ptr_array([c.data_ptr() for c in cens]),
                                              ptr_array([d.data_ptr() for d in dps]), hs, ws, float(thres),
                                              ptr_array([m.data_ptr() for m in cmaps]), k_dev.data_ptr(),
                                              scratch.data_ptr(), _stream()))
    if k_on_device:
        return cmaps, k_dev
    return cmaps, [int(k) for k in k_dev.cpu().tolist()]


def label4(mask):
    """4-connected components of a GPU mask [n,h,w] or [h,w] (non-zero = foreground): int32 ids 1..
    per image in raster order of each component's first pixel, 0 background; and counts [n]."""
    _need_cuda(mask, "mask")
    squeeze = mask.dim() == 2
    m = (mask != 0).to(torch.uint8).contiguous()
    if squeeze:
        m = m[None]
    n, h, w = m.shape
    labels = torch.empty((n, h, w), dtype=torch.int32, device=m.device)
    counts = torch.empty(n, dtype=torch.int32, device=m.device)
    scratch = torch.empty(lib.irn_ccl_scratch_bytes(n, h, w), dtype=torch.uint8, device=m.device)
    with torch.cuda.device(m.device):
        check(lib.irn_label4(m.data_ptr(), n, h, w, labels.data_ptr(), counts.data_ptr(), scratch.data_ptr(), _stream()))
    return (labels[0], counts[0]) if squeeze else (labels, counts)


def detect_instance(rw_up, argmax, class_ids, n_channels, max_fragment_size=0):
    """Pixel-wise instance ids -> detections (reference step/make_ins_seg_labels.py:82-105), on GPU.

    rw_up: fp32 [C',H,W] normalised scores; argmax: int32 [H,W] (0 = bg, c+1 = channel c);
    class_ids: int64 [C'] (np.repeat(keys, K)).  Every 4-connected component of every channel's
    mask becomes a detection: score = max(rw_up[c] over the component), or 0 when it has fewer than
    max_fragment_size pixels.  Labelling, areas, scores and the [N,H,W] masks are produced by
    libirn_hip.so (irn_detect_instance_count / _emit): one pass over the class map, one 4-byte sync
    for N, one transfer of the result.  Returns the reference's numpy dict {'score','mask','class'}
    in its order (channel ascending, component ascending by first pixel).  Raises ValueError when
    nothing is detected (the reference crashes in np.stack([]) — SURVEY.md §3.5)."""
    _need_cuda(rw_up, "rw_up")
    _need_cuda(argmax, "argmax")
    dev = rw_up.device
    sc = rw_up.contiguous().float()
    am = argmax.to(torch.int32).contiguous()
    n_channels = int(n_channels)
    h, w = am.shape
    if sc.shape != (n_channels, h, w):
        raise ValueError("rw_up must be [%d,%d,%d], got %s" % (n_channels, h, w, tuple(sc.shape)))
    class_ids = np.asarray(class_ids)
    npx = h * w
    scratch = _cached("det_scratch", dev, lib.irn_detect_scratch_bytes(n_channels, h, w), torch.uint8)
    n = C.c_int()
    with torch.cuda.device(dev):
        check(lib.irn_detect_instance_count(sc.data_ptr(), am.data_ptr(), n_channels, h, w, C.byref(n),
                                            scratch.data_ptr(), _stream()))
        nd = n.value
        if nd == 0:
            raise ValueError("detect_instance: no foreground pixel in any channel")
        # one packed device buffer [score fp32 | channel int32 | masks uint8] -> one transfer into pinned memory
        # (a pageable .cpu() of the masks plus fresh allocations cost 3.6 ms per 512^2 image: 10x the rest of the step)
        head = 8 * nd
        packed = _cached("det_out", dev, head + nd * npx, torch.uint8)
        base = packed.data_ptr()
        check(lib.irn_detect_instance_emit(sc.data_ptr(), am.data_ptr(), n_channels, h, w, nd, float(max_fragment_size),
                                           base, base + 4 * nd, base + head, scratch.data_ptr(), _stream()))
        host = _cached("det_host", "pinned", head + nd * npx, torch.uint8)
        host[:head + nd * npx].copy_(packed[:head + nd * npx], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    raw = host.numpy()
    score = raw[:4 * nd].view(np.float32).copy()
    chan = raw[4 * nd:head].view(np.int32)
    mask = raw[head:head + nd * npx].view(np.bool_).reshape(nd, h, w).copy()
    return {"score": score, "mask": mask, "class": class_ids[chan]}


class PendingDetections:
    """Detections of a batch whose packed transfer to the host may still be in flight (`detect_instance_batch(...,
    deferred=True)`): `result()` waits for it and returns the list `detect_instance_batch` returns.  The transfer runs on
    a copy stream of its own, so the caller can enqueue the next batch's kernels before collecting this one."""

    def __init__(self, n, host, done, nds, offs, hs, ws, class_ids, timings, t_emit):
        self._n, self._host, self._done, self._nds, self._offs = n, host, done, nds, offs
        self._hs, self._ws, self._class_ids, self._timings, self._t_emit = hs, ws, class_ids, timings, t_emit
        self._out = None

    def result(self):
        if self._out is not None:
            return self._out
        import time
        out = []
        if self._host is None:                      # no foreground pixel in any image of the batch
            out = [ValueError("detect_instance: no foreground pixel in any channel") for _ in range(self._n)]
        else:
            self._done.synchronize()                                               # host round trip 2
            t_done = time.perf_counter()
            raw = self._host.numpy()
            for i in range(self._n):
                nd = self._nds[i]
                if nd == 0:
                    out.append(ValueError("detect_instance: no foreground pixel in any channel"))
                    continue
                o_sc, o_ch, o_mk = self._offs[i]
                score = raw[o_sc:o_sc + 4 * nd].view(np.float32)
                chan = raw[o_ch:o_ch + 4 * nd].view(np.int32)
                mask = raw[o_mk:o_mk + nd * self._hs[i] * self._ws[i]].view(np.bool_).reshape(nd, self._hs[i], self._ws[i])
                out.append({"score": score, "mask": mask, "class": np.asarray(self._class_ids[i])[chan]})
            if self._timings is not None:            # seconds: emit + packed transfer (as far as the caller waited for it), unpacking
                self._timings["emit_d2h"] = self._timings.get("emit_d2h", 0.0) + t_done - self._t_emit
                self._timings["unpack"] = self._timings.get("unpack", 0.0) + time.perf_counter() - t_done
        self._out = out
        return out


_COPY_STREAMS = {}


def _copy_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


def detect_instance_batch(rw_ups, argmaxes, class_ids, n_channels, max_fragment_sizes, timings=None, deferred=False):
    """detect_instance for a batch of images with two host round trips in total (the per-image form has three per
    image): one 4-byte-per-image transfer of the detection counts, one packed transfer of every image's
    {score, channel, masks} (irn_detect_instance_batch_count / _emit).  Arguments are lists (one entry per image) of
    what `detect_instance` takes.  Returns a list with, per image, the reference's numpy dict or — for an image without
    any foreground pixel — the ValueError `detect_instance` would raise.  With `deferred=True` the packed transfer is
    left in flight on a copy stream and a `PendingDetections` is returned: call its `result()` after the next batch has
    been enqueued and the 2 MB of masks per image cross PCIe under that batch's kernels."""
    import time
    t_start = time.perf_counter()
    n = len(rw_ups)
    dev = rw_ups[0].device
    scs, ams, hs, ws = [], [], [], []
    for i in range(n):
        _need_cuda(rw_ups[i], "rw_up")
        _need_cuda(argmaxes[i], "argmax")
        am = argmaxes[i].to(torch.int32).contiguous()
        sc = rw_ups[i].contiguous().float()
        h, w = am.shape
        if sc.shape != (int(n_channels[i]), h, w):
            raise ValueError("rw_up[%d] must be [%d,%d,%d], got %s" % (i, n_channels[i], h, w, tuple(sc.shape)))
        scs.append(sc); ams.append(am); hs.append(h); ws.append(w)
    cs_a, hs_a, ws_a = i32_array(n_channels), i32_array(hs), i32_array(ws)
    sc_p, am_p = ptr_array([t.data_ptr() for t in scs]), ptr_array([t.data_ptr() for t in ams])
    scratch = _cached("det_scratch_b", dev, lib.irn_detect_batch_scratch_bytes(n, cs_a, hs_a, ws_a), torch.uint8)
    n_det_dev = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.irn_detect_instance_batch_count(n, sc_p, am_p, cs_a, hs_a, ws_a, n_det_dev.data_ptr(),
                                                  scratch.data_ptr(), _stream()))
        nds = [int(v) for v in n_det_dev.cpu().tolist()]                       # host round trip 1
        t_count = time.perf_counter()
        if timings is not None:          # seconds: labelling + count transfer
            timings["count"] = timings.get("count", 0.0) + t_count - t_start
        # packed output: per image [score fp32 x nd | channel int32 x nd | pad to 16 | masks uint8 nd x h x w | pad to 16]
        offs, total = [], 0
        for i in range(n):
            head = (8 * nds[i] + 15) // 16 * 16
            offs.append((total, total + 4 * nds[i], total + head))
            total += head + (nds[i] * hs[i] * ws[i] + 15) // 16 * 16
        if total == 0:
            pending = PendingDetections(n, None, None, nds, offs, hs, ws, class_ids, timings, t_count)
            return pending if deferred else pending.result()
        # the device-side staging buffer: the cached one when the call waits for its transfer, one of the batch's own
        # when the transfer is left in flight (the next batch must not write into it)
        packed = (torch.empty(total + 16, dtype=torch.uint8, device=dev) if deferred
                  else _cached("det_out_b", dev, total + 16, torch.uint8))
        base = (packed.data_ptr() + 15) // 16 * 16
        shift = base - packed.data_ptr()
        live = [nd > 0 for nd in nds]
        check(lib.irn_detect_instance_batch_emit(
            n, sc_p, am_p, cs_a, hs_a, ws_a, i32_array(nds), (C.c_double * n)(*[float(v) for v in max_fragment_sizes]),
            ptr_array([base + o[0] if l else None for o, l in zip(offs, live)]),
            ptr_array([base + o[1] if l else None for o, l in zip(offs, live)]),
            ptr_array([base + o[2] if l else None for o, l in zip(offs, live)]), scratch.data_ptr(), _stream()))
        # a page-locked buffer of its own for every batch: the detections are handed out as VIEWS of it (no second copy of
        # 2 MB of masks per image) and it goes back to torch's caching host allocator when the last of them is dropped
        host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        done = torch.cuda.Event()
        if deferred:
            side, cur = _copy_stream(dev), torch.cuda.current_stream()
            emitted = torch.cuda.Event()
            emitted.record(cur)
            side.wait_event(emitted)
            with torch.cuda.stream(side):
                host.copy_(packed[shift:shift + total], non_blocking=True)
                done.record(side)
            packed.record_stream(side)               # the allocator may hand the block out again only behind the copy
        else:
            host.copy_(packed[shift:shift + total], non_blocking=True)
            done.record(torch.cuda.current_stream())
    if timings is not None:
        timings["bytes"] = timings.get("bytes", 0) + total
    pending = PendingDetections(n, host, done, nds, offs, hs, ws, class_ids, timings, t_count)
    return pending if deferred else pending.result()


_CACHE = {}


def _cached(tag, dev, nbytes, dtype):
    """Grow-only scratch buffers (device) / pinned staging (dev == "pinned"), one per tag and device."""
    key = (tag, str(dev))
    buf = _CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        n = int(nbytes * 1.25) + 256
        buf = torch.empty(n, dtype=dtype, pin_memory=True) if dev == "pinned" else torch.empty(n, dtype=dtype, device=dev)
        _CACHE[key] = buf
    return buf
