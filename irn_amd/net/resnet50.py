"""ResNet-50 trunk for the CAM / IRNet backbones, host side on PyTorch-ROCm (MIOpen convolutions).

Mirrors the parameter naming of the reference trunk (reference net/resnet50.py:11-108) so that
checkpoints written by the reference (`res50_cam.pth`, `res50_irn.pth`) load unchanged:
``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``.
Differences by design:
  * construction never touches the network (the reference downloads ImageNet weights in the
    constructor, net/resnet50.py:111-118); weights are an *input* to the hot path;
  * batch-norm layers always use their running statistics (reference `FixedBatchNorm`,
    net/resnet50.py:11-14) — here that is simply the module's forward;
  * on the inference path (GPU tensor, autograd off) the elementwise tail of every unit — batch norm, the residual add
    and the ReLU (net/resnet50.py:34-54, :94-97) — is ONE in-place pass of a hand-written kernel over the convolution's
    output (`ops.bn_act_`, irn_amd/csrc/bn_act.hip) instead of three kernels and seven tensor transfers; the
    convolutions stay on MIOpen.  With autograd on (the training seam) or on the CPU the composed PyTorch ops run.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_PLANES = (64, 128, 256, 512)


# IRN_FUSED_EPILOGUE=0 keeps the composed PyTorch ops everywhere (A/B runs)
FUSED_EPILOGUE = os.environ.get("IRN_FUSED_EPILOGUE", "1") != "0"


# Layout of the trunk's four stages.  MIOpen's NHWC solvers beat its NCHW ones on this network — `cam` 97.4 -> 104.6,
# `e2e` 81.1 -> 86.6 images/s (profiles/r04_s3_channels_last_ab.txt) — but ONLY for problems its find database was TUNED
# for: an untuned NHWC problem falls to an immediate-mode pick that is 2.2x SLOWER than NCHW's (43.6 images/s).  So the
# layout is chosen per input shape:
#   IRN_CHANNELS_LAST = "auto" (default)  channels-last for the network-input shapes [n, H, W] listed in the find database
#                                         shipped for this device (irn_amd/data/miopen/<key>/nhwc_shapes.json, written by
#                                         tools/miopen_warmup.py --channels-last 1; step/_common.miopen_setup merges that
#                                         database into the process's user database), NCHW for every other shape;
#                       "1" / "0"          always / never.
# The stem stays NCHW (a 3-channel input); its pooled output is converted once, and whoever consumes a stage's output in
# NCHW (the IRNet heads, the CAM merge) converts it back.  Convolution weights stay as they are: PyTorch hands MIOpen a
# re-laid-out copy per call (94 MB per forward, < 1 % of it).
CHANNELS_LAST_MODE = os.environ.get("IRN_CHANNELS_LAST", "auto")
_TUNED_SHAPES = {}
# The reproducible mode (step/_common.deterministic_backbones; set by `apply_deterministic_setting` in every process that sets
# MIOpen up): None = not managed here (a caller's own torch.backends.cudnn.deterministic is respected), True / False = managed.
DETERMINISTIC = None


def tuned_nhwc_shapes():
    """Network-input shapes (n, H, W) the shipped find database OF THIS PROCESS'S MODE holds tuned NHWC solvers for (empty
    without a GPU or a database for this device / HIP version)."""
    try:
        from ..step import _common
        key = _common.miopen_mode_key()
    except Exception:
        return set()
    if key not in _TUNED_SHAPES:
        shapes = set()
        try:
            import json
            path = os.path.join(_common.miopen_seed_root(), key, "nhwc_shapes.json")
            if os.path.exists(path) and os.environ.get("IRN_MIOPEN_SEED", "1") != "0":
                shapes = {tuple(int(v) for v in s) for s in json.load(open(path))}
            elif torch.cuda.is_available() and CHANNELS_LAST_MODE == "auto":
                _common.warn_missing_shipped_data()
        except Exception:
            shapes = set()
        _TUNED_SHAPES[key] = shapes
    return _TUNED_SHAPES[key]


def _tuned(x):
    return (int(x.shape[0]), int(x.shape[2]), int(x.shape[3])) in tuned_nhwc_shapes()


def channels_last_for(x):
    """Does the trunk run channels-last for the network input `x` [n, 3, H, W]?  (inference path only.)  Called once at the
    start of every trunk pass; in the reproducible mode it also sets MIOpen's deterministic attribute for the pass: off for a
    channels-last pass of a shape the filtered database covers (it has no order-dependent solver left), on otherwise —
    also for a channels-last pass forced by IRN_CHANNELS_LAST=1 on a shape the database does not know, where MIOpen would
    otherwise search and could pick a split-K solver; `end_trunk_pass` puts it back on."""
    if CHANNELS_LAST_MODE == "0" or not x.is_cuda or x.dim() != 4 or torch.is_grad_enabled():
        if DETERMINISTIC and x.is_cuda:
            torch.backends.cudnn.deterministic = True
        return False
    if DETERMINISTIC is None and torch.backends.cudnn.deterministic:
        # a caller's own deterministic flag: MIOpen's attribute leaves no fast NHWC fp32 solver, the NCHW trunk it is
        return False
    # auto: only in a process whose MIOpen user database has been completed from the shipped one (step/_common.miopen_setup:
    # the steps' workers, the in-process step path, bench.py) — anywhere else the NHWC problems would be untuned
    covered = bool(os.environ.get("IRN_MIOPEN_DB_SET")) and _tuned(x)
    cl = CHANNELS_LAST_MODE == "1" or covered
    if DETERMINISTIC:
        torch.backends.cudnn.deterministic = not (cl and covered)
    return cl


# Trunk passes of this process by layout, rows added to fill a partial batch, and the NCHW passes by network-input size
# (what `untuned_report` prints at the end of a step)
PASS_STATS = {"channels_last": 0, "nchw": 0, "pad_rows": 0, "nchw_sizes": {}}


def pass_rows(x):
    """Reproducible mode: how many rows ([image, flip] pairs back to back) ONE trunk pass over network inputs like `x`
    [n, 3, H, W] carries — 16 (the batch the shipped database is tuned for) when H x W is a tuned size, else 2 (one pair,
    the reference's own batch, step/make_cam.py:32-33); None outside the mode (the caller's batch is the pass).  MIOpen
    resolves a solver per PROBLEM and the batch is part of the problem, so a row's bits would otherwise depend on how many
    others travel with it — on the shard split, the tail of a shard, an early flush."""
    if not DETERMINISTIC or not x.is_cuda or x.dim() != 4 or torch.is_grad_enabled():
        return None
    if CHANNELS_LAST_MODE != "0" and bool(os.environ.get("IRN_MIOPEN_DB_SET")):
        rows = [n for (n, h, w) in tuned_nhwc_shapes() if (h, w) == (int(x.shape[2]), int(x.shape[3]))]
        if rows:
            return max(rows)
    return 2


def run_rows(fn, x):
    """`fn(x)` for a row-wise independent `fn` ([n, 3, H, W] -> tensor or tuple of tensors with n leading rows).  In the
    reproducible mode the rows go through in passes of exactly `pass_rows(x)` — a short last pass is filled with zero images
    whose outputs are dropped — so that the kernels a row meets are a function of its own size only."""
    rows = pass_rows(x)
    n = int(x.shape[0])
    if rows is None or n == rows:
        return fn(x)
    outs = []
    for i in range(0, n, rows):
        chunk = x[i:i + rows]
        valid = int(chunk.shape[0])
        if valid < rows:
            PASS_STATS["pad_rows"] += rows - valid
            chunk = torch.cat([chunk, chunk.new_zeros((rows - valid,) + tuple(chunk.shape[1:]))])
        y = fn(chunk)
        outs.append(tuple(t[:valid] for t in y) if isinstance(y, tuple) else y[:valid])
    if len(outs) == 1:
        return outs[0]
    if isinstance(outs[0], tuple):
        return tuple(torch.cat([o[k] for o in outs]) for k in range(len(outs[0])))
    return torch.cat(outs)


def count_pass(x, channels_last):
    PASS_STATS["channels_last" if channels_last else "nchw"] += 1
    if not channels_last and x.is_cuda:
        key = "%dx%d" % (int(x.shape[2]), int(x.shape[3]))
        PASS_STATS["nchw_sizes"][key] = PASS_STATS["nchw_sizes"].get(key, 0) + 1


def end_trunk_pass():
    """Behind the last trunk stage of a pass (before heads, classifier read-out in NCHW, merges): the reproducible mode's
    resting state — MIOpen's deterministic attribute on."""
    if DETERMINISTIC:
        torch.backends.cudnn.deterministic = True


def _dense(x):
    """contiguous in NCHW or (4-D) in channels-last order"""
    return x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))


def _fused(x):
    # irn_bn_act walks the tensor in 16-byte pieces: an offset view of another tensor takes the composed ops instead
    return (FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and _dense(x) and x.data_ptr() % 16 == 0
            and not torch.is_grad_enabled())


# IRN_FUSED_GEMM=0 keeps the trunk's 1x1 convolutions on MIOpen with a separate `bn_act_` pass behind each (A/B runs)
FUSED_GEMM = os.environ.get("IRN_FUSED_GEMM", "1") != "0"


# Split-precision form of those GEMMs (round 6, include/irn_hip.h irn_split16 / irn_gemm16_nhwc): fp16 hi/lo operands carry 22
# mantissa bits, the fp16 matrix pipe is ~2.5x faster over a 3x longer K, and the result is as close to fp64 as the fp32 GEMM's
# (tools/bf16x3_cam_error.py: 5.4e-6 / 9.2e-6 on the normalised CAM against 5.9e-6 / 7.6e-6).  IRN_SPLIT_GEMM=0 switches it off.
#   conv3's operand comes out of the 3x3 convolution's own tail (batch norm + ReLU + split in ONE pass, nothing extra): every
#       unit with at least SPLIT_MIN_PLANES planes;
#   conv1's / the stride-1 shortcut's operand is the block input, which must also stay fp32 (it is the residual): a split pass
#       of its own (10 bytes per element), worth it only where the GEMMs it feeds are large — cin * (sum of their cout) >=
#       SPLIT_MIN_INPUT (stage 4 of the trunk; profiles/r06_s2_split_gemm_fp16_note.txt).
SPLIT_GEMM = os.environ.get("IRN_SPLIT_GEMM", "1") != "0"
SPLIT_MIN_PLANES = int(os.environ.get("IRN_SPLIT_MIN_PLANES", "64"))
SPLIT_MIN_INPUT = int(os.environ.get("IRN_SPLIT_MIN_INPUT", str(1 << 20)))
#   conv2 (3x3, stride 1) of units with at least SPLIT_MIN_PLANES_3X3 planes on maps of at least SPLIT_MIN_ROWS_3X3 pixels per
#       pass: accumulating split GEMMs on row-shifted views of ONE zero-bordered operand (ops.conv3x3_split), no im2col — nine
#       over 3 cin, or (default) three over 9 cin: the three taps of a kernel row are consecutive memory rows, so the operand of a
#       kernel row is the same buffer read with overlapping rows (leading dimension 3 cin, 9 cin columns).  Against MIOpen's
#       fp32 convolution at 512 planes: 1.7-2.2x for the nine (profiles/r06_s6_conv3x3_split_probe.txt); end to end `cam` 133.6 ->
#       153.0 (nine) -> 164.6 (three) -> 166.0 with the 128-plane stage included, 158.5 with the 64-plane one too
#       (profiles/r06_s9_conv3x3_row_fused_ab.txt): stages 2-4.
SPLIT_MIN_PLANES_3X3 = int(os.environ.get("IRN_SPLIT_MIN_PLANES_3X3", "128"))
SPLIT_MIN_ROWS_3X3 = int(os.environ.get("IRN_SPLIT_MIN_ROWS_3X3", "8192"))


def _gemm_path(x):
    """A bottleneck's 1x1 convolutions run as hipBLASLt GEMMs with a fused epilogue when its input is a channels-last
    activation on the inference path (the layout in which the activation IS the GEMM's operand)."""
    return (FUSED_GEMM and FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_grad_enabled()
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def to_stage_format(x, channels_last):
    """Activation entering a trunk stage."""
    return x.contiguous(memory_format=torch.channels_last) if channels_last else x


def to_nchw(x):
    """A stage's output for a consumer that wants NCHW (heads, hand-written kernels, the CAM merge); ends the trunk pass."""
    end_trunk_pass()
    return x if x.is_contiguous() else x.contiguous()


def _version(t):
    try:
        return t._version
    except RuntimeError:          # tensors created under torch.inference_mode() carry no version counter
        return -1


class FrozenBatchNorm(nn.BatchNorm2d):
    """Inference-statistics batch norm regardless of train()/eval() (net/resnet50.py:11-14)."""

    _folded = None

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                            False, 0.0, self.eps)

    def folded(self):
        """(scale, shift) fp32 [C] with forward(x) = x * scale + shift, folded in double precision; cached until a
        parameter or statistic is written or moved."""
        src = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((t.data_ptr(), _version(t)) for t in src)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                w, b, mean, var = (t.detach().double() for t in src)
                scale = w / torch.sqrt(var + self.eps)
                shift = b - mean * scale
            self._folded = (key, scale.float().contiguous(), shift.float().contiguous())
        return self._folded[1], self._folded[2]

    def apply_(self, x, residual=None, relu=False, residual_bn=None):
        """act(forward(x) (+ r)) with r = residual, or residual_bn.forward(residual) when a second layer is given (the
        projection shortcut's batch norm); overwrites x on the inference path (x must be a tensor nobody else reads: the
        output of the convolution in front of this layer)."""
        if _fused(x) and (residual is None or (residual.dtype == x.dtype and residual.data_ptr() % 16 == 0 and residual.stride() == x.stride())):
            from .. import ops          # the HIP library; raises if it has not been built — there is no other GPU path
            scale, shift = self.folded()
            return ops.bn_act_(x, scale, shift, residual, relu, None if residual_bn is None else residual_bn.folded())
        y = self.forward(x)
        if residual is not None:
            y = y + (residual if residual_bn is None else residual_bn.forward(residual))
        return F.relu(y) if relu else y


def stem(conv1, bn1, maxpool, x):
    """conv1 -> bn1 -> ReLU -> maxpool (net/resnet50.py:94-97); on the inference path everything behind the convolution is
    one pass (`ops.stem_pool`) when the pool is the trunk's 3x3 / stride 2 / pad 1."""
    if DETERMINISTIC and x.is_cuda:
        # the stem is an NCHW problem in every layout: in the reproducible mode it always runs under MIOpen's attribute
        prev = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = True
        try:
            y = conv1(x)
        finally:
            torch.backends.cudnn.deterministic = prev
    else:
        y = conv1(x)
    if _fused(y) and y.is_contiguous() and (maxpool.kernel_size, maxpool.stride, maxpool.padding, maxpool.dilation, maxpool.ceil_mode) == (3, 2, 1, 1, False):
        from .. import ops
        return ops.stem_pool(y, *bn1.folded())
    return maxpool(bn1.apply_(y, relu=True))


class Stem(nn.Sequential):
    """conv1, bn1, relu, maxpool (+ following stages) as the nets register them (reference net/resnet50_cam.py:14-15,
    net/resnet50_irn.py:14): the same children and state-dict keys as a plain Sequential, with batch norm + ReLU taken
    and the pool taken in one pass."""

    def forward(self, x):
        cl = channels_last_for(x)
        count_pass(x, cl)
        x = stem(self[0], self[1], self[3], x)
        rest = list(self)[4:]
        if rest:
            x = to_stage_format(x, cl)
        for m in rest:
            x = m(x)
        return x


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride, dilation) -> 1x1(x4) residual unit (net/resnet50.py:17-57)."""
    expansion = 4

    def __init__(self, c_in, planes, stride=1, project=False, dilation=1):
        super().__init__()
        c_out = planes * self.expansion
        self.conv1 = nn.Conv2d(c_in, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, c_out, 1, bias=False)
        self.bn3 = FrozenBatchNorm(c_out)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride=stride, bias=False),
                                            FrozenBatchNorm(c_out))

    _gemm = None

    def gemm_params(self):
        """The unit's 1x1 convolutions as GEMM operands: weights [cout, cin] with their batch norm's scale folded in
        (double precision, rounded once), the shifts as biases; for a projection unit the shortcut's shift rides in
        conv3's bias (the shortcut GEMM has none).  Cached until a parameter or statistic is written or moved."""
        convs = [self.conv1, self.conv3] + ([] if self.downsample is None else [self.downsample[0]])
        w2 = self.conv2.weight
        bns = [self.bn1, self.bn3] + ([] if self.downsample is None else [self.downsample[1]])
        src = [c.weight for c in convs] + [w2] + [t for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var)]
        key = tuple((t.data_ptr(), _version(t)) for t in src)
        if self._gemm is None or self._gemm[0] != key:
            with torch.no_grad():
                folded, folded64 = [], []
                for c, b in zip(convs, bns):
                    scale = b.weight.detach().double() / torch.sqrt(b.running_var.detach().double() + b.eps)
                    shift = b.bias.detach().double() - b.running_mean.detach().double() * scale
                    folded64.append(c.weight.detach().double() * scale.view(-1, 1, 1, 1))
                    folded.append((folded64[-1].float().contiguous(), shift))
                p = {"w1": folded[0][0].flatten(1), "b1": folded[0][1].float().contiguous(), "w3": folded[1][0].flatten(1)}
                # the 3x3 weight in the layout MIOpen's NHWC solvers take: PyTorch otherwise re-lays it out on EVERY call
                # (1.7 % of the e2e GPU time in strided copy kernels, profiles/r05_s15_e2e_kernel_classes.txt)
                p["w2"] = self.conv2.weight.detach().contiguous(memory_format=torch.channels_last)
                if self.downsample is None:
                    p["b3"] = folded[1][1].float().contiguous()
                else:
                    p["wd"] = folded[2][0].contiguous(memory_format=torch.channels_last)   # [cout, cin, 1, 1]: also MIOpen's operand when strided
                    p["b3"] = (folded[1][1] + folded[2][1]).float().contiguous()
                if SPLIT_GEMM and self.conv1.weight.is_cuda:
                    # fp16 hi/lo operands of the split-precision GEMMs, from the weights folded in double precision; bn2's
                    # constants ride in the split pass that feeds conv3
                    from .. import ops
                    planes, cin = self.conv3.weight.shape[1], self.conv1.weight.shape[1]
                    couts = self.conv1.weight.shape[0] + (self.downsample[0].weight.shape[0] if self.downsample is not None and tuple(self.downsample[0].stride) == (1, 1) else 0)
                    if planes >= SPLIT_MIN_PLANES:
                        p["w3_16"], p["a3"] = ops.split_weight(folded64[1].flatten(1))
                        p["s2"], p["t2"] = self.bn2.folded()
                        c2 = self.conv2
                        if (planes >= SPLIT_MIN_PLANES_3X3 and planes % 8 == 0 and tuple(c2.kernel_size) == (3, 3) and tuple(c2.stride) == (1, 1)
                                and tuple(c2.padding) == (1, 1) and tuple(c2.dilation) == (1, 1) and c2.groups == 1):
                            p["w2_16"], p["a2"] = ops.split_weight_3x3(c2.weight.detach().double())
                    if cin * couts >= SPLIT_MIN_INPUT and cin % 8 == 0:
                        p["w1_16"], p["a1"] = ops.split_weight(folded64[0].flatten(1))
                        if self.downsample is not None and tuple(self.downsample[0].stride) == (1, 1):
                            p["wd_16"], p["ad"] = ops.split_weight(folded64[2].flatten(1))
            self._gemm = (key, p)
        return self._gemm[1]

    def forward(self, x):
        if _gemm_path(x):
            return self._forward_gemm(x)
        y = self.bn1.apply_(self.conv1(x), relu=True)
        y = self.bn2.apply_(self.conv2(y), relu=True)
        if self.downsample is None:
            return self.bn3.apply_(self.conv3(y), residual=x, relu=True)
        # the shortcut's batch norm rides in the same pass: its output never exists as a tensor
        return self.bn3.apply_(self.conv3(y), residual=self.downsample[0](x), relu=True, residual_bn=self.downsample[1])

    def _forward_gemm(self, x):
        """Channels-last inference path: conv1 and conv3 are GEMMs over the [pixels, channels] matrix the activation
        already is, with batch norm, residual and ReLU in their epilogue — fp32 (`ops.conv1x1_nhwc`) or, where `gemm_params`
        prepared the fp16 hi/lo weight operands, in split precision (`ops.gemm16_nhwc` on `ops.split16`'s operand; module
        comment above SPLIT_GEMM).  The 3x3 convolution is MIOpen's with one in-place `bn_act_` pass behind it, or — stride 1,
        >= SPLIT_MIN_PLANES_3X3 planes — three accumulating split GEMMs on a zero-bordered operand (`ops.conv3x3_split`) whose
        bordered result feeds bn2 + ReLU + the split for conv3 in one pass."""
        from .. import ops
        p = self.gemm_params()
        x3 = ops.split16(x) if "w1_16" in p else None                     # the block input as fp16 hi/lo (it stays fp32 too: the residual)
        if x3 is not None:
            y = ops.gemm16_nhwc(x3, p["w1_16"], (x.shape[0], p["w1"].shape[0], x.shape[2], x.shape[3]), p["b1"], relu=True, alpha=p["a1"])
        else:
            y = ops.conv1x1_nhwc(x, p["w1"], p["b1"], relu=True)
        c2 = self.conv2
        y_pad = None
        if "w2_16" in p and y.shape[0] * y.shape[2] * y.shape[3] >= SPLIT_MIN_ROWS_3X3:
            y_shape = tuple(int(v) for v in y.shape)
            y_pad = ops.conv3x3_split(y, p["w2_16"], p["a2"])        # bordered fp32 [N (H+2)(W+2), planes]
        else:
            y = F.conv2d(y, p["w2"], None, c2.stride, c2.padding, c2.dilation, c2.groups)
            y = y.contiguous(memory_format=torch.channels_last)
        ds = self.downsample[0] if self.downsample is not None else None
        sc = None
        if ds is not None:
            if tuple(ds.stride) != (1, 1):      # a strided shortcut is not a matrix view of x: MIOpen, with the folded weight
                sc = F.conv2d(x, p["wd"], None, ds.stride).contiguous(memory_format=torch.channels_last)
            elif "wd_16" in p:
                sc = ops.gemm16_nhwc(x3, p["wd_16"], (x.shape[0], p["wd"].shape[0], x.shape[2], x.shape[3]), alpha=p["ad"])
            else:
                sc = ops.conv1x1_nhwc(x, p["wd"].reshape(p["wd"].shape[0], -1))
        res = x if ds is None else sc
        if y_pad is not None:
            y3 = ops.split16_pad(y_pad, y_shape, p["s2"], p["t2"], relu=True, in_padded=True)
            return ops.gemm16_nhwc(y3, p["w3_16"], (y_shape[0], p["w3"].shape[0], y_shape[2], y_shape[3]), p["b3"], residual=res, relu=True,
                                   alpha=p["a3"], out=sc)
        if "w3_16" in p and y.shape[1] % 8 == 0:
            # bn2 + ReLU + split in one pass over the 3x3 convolution's output, then ONE fp16 GEMM with the unit's whole tail
            y3 = ops.split16(y, p["s2"], p["t2"], relu=True)
            return ops.gemm16_nhwc(y3, p["w3_16"], (y.shape[0], p["w3"].shape[0], y.shape[2], y.shape[3]), p["b3"], residual=res, relu=True,
                                   alpha=p["a3"], out=sc)
        y = self.bn2.apply_(y, relu=True)
        return ops.conv1x1_nhwc(y, p["w3"], p["b3"], residual=res, relu=True, out=sc)


def _stage(c_in, planes, n_blocks, stride, dilation):
    c_out = planes * Bottleneck.expansion
    units = [Bottleneck(c_in, planes, stride=stride, project=(stride != 1 or c_in != c_out), dilation=1)]
    units += [Bottleneck(c_out, planes, dilation=dilation) for _ in range(n_blocks - 1)]
    return nn.Sequential(*units)


class ResNet50Trunk(nn.Module):
    """conv1/bn1/maxpool + four bottleneck stages; no pooling head (the reference never uses it)."""

    def __init__(self, strides=(2, 2, 2, 2), dilations=(1, 1, 1, 1)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=strides[0], padding=3, bias=False)
        self.bn1 = FrozenBatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        stage_strides = (1,) + tuple(strides[1:])
        c_in = 64
        for i, (planes, n, s, d) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS, stage_strides, dilations)):
            setattr(self, "layer%d" % (i + 1), _stage(c_in, planes, n, s, d))
            c_in = planes * Bottleneck.expansion

    def forward(self, x):
        cl = channels_last_for(x)
        count_pass(x, cl)
        x = to_stage_format(stem(self.conv1, self.bn1, self.maxpool, x), cl)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnet50(pretrained=False, state_dict=None, **kw):
    """Build the trunk.  ``pretrained`` is accepted for signature compatibility with the reference
    (net/resnet50.py:111) but never downloads; pass ``state_dict`` to initialise explicitly."""
    net = ResNet50Trunk(**kw)
    if state_dict is not None:
        sd = {k: v for k, v in state_dict.items() if not k.startswith("fc.")}
        net.load_state_dict(sd)
    return net
