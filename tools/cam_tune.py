#!/usr/bin/env python3
"""Explore PyTorch-ROCm settings for the CAM backbone (plumbing, not a hand-written kernel):
memory format, BN folding, MIOpen find mode.  Prints images/s per variant and the max-abs deviation of the
CAM output from the plain fp32 NCHW forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
if mode == "fast":
    os.environ["MIOPEN_FIND_MODE"] = "2"
import torch
import torch.nn as nn
import torch.nn.functional as F

from irn_amd.net import resnet50_cam, weights

torch.backends.cudnn.benchmark = mode != "fast"
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "4"))
scales = (1.0, 0.5, 1.5, 2.0)
g = torch.Generator().manual_seed(0)
imgs = {s: torch.randn(B, 3, int(512 * s), int(512 * s), generator=g).to(dev) for s in scales}


def fold_bn(model):
    """conv -> frozen BN  ==>  conv with scaled weights + bias (exact up to fp32 rounding)."""
    from irn_amd.net.resnet50 import Bottleneck

    def fuse(conv, bn):
        w = conv.weight
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        conv.weight = nn.Parameter(w * scale.view(-1, 1, 1, 1), requires_grad=False)
        conv.bias = nn.Parameter(bn.bias - bn.running_mean * scale, requires_grad=False)

    t = model.resnet50
    fuse(t.conv1, t.bn1)
    t.bn1.forward = lambda x: x
    for m in t.modules():
        if isinstance(m, Bottleneck):
            for c, b in ((m.conv1, m.bn1), (m.conv2, m.bn2), (m.conv3, m.bn3)):
                fuse(c, b)
                b.forward = (lambda x: x)
            if m.downsample is not None:
                fuse(m.downsample[0], m.downsample[1])
                m.downsample[1].forward = (lambda x: x)
    return model


def build(fold=False, cl=False):
    net = resnet50_cam.CAM()
    net.load_state_dict(weights.random_cam_state(1))
    net = net.to(dev).eval()
    if fold:
        net = fold_bn(net)
    if cl:
        net = net.to(memory_format=torch.channels_last)
    return net


def run(net, cl=False):
    outs = []
    with torch.no_grad():
        for s in scales:
            x = imgs[s]
            x = torch.cat([x, x.flip(-1)], 0)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
            f = F.relu(F.conv2d(net.features(x), net.classifier.weight))
            outs.append(f[:B] + f[B:].flip(-1))
    return outs


ref = None
for name, fold, cl in (("nchw", False, False), ("nchw+foldbn", True, False), ("nhwc", False, True), ("nhwc+foldbn", True, True)):
    net = build(fold, cl)
    t0 = time.time()
    o = run(net, cl)
    torch.cuda.synchronize()
    t_first = time.time() - t0
    run(net, cl)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 3
    for _ in range(n):
        o = run(net, cl)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    if ref is None:
        ref = [x.clone() for x in o]
    err = max(float((a - b).abs().max()) for a, b in zip(o, ref))
    mx = max(float(b.abs().max()) for b in ref)
    print("%-6s %-12s B=%d  %.1f img/s  (%.1f ms/img)  first-call %.1fs  max|d| vs nchw %.2e (ref max %.2f)" %
          (mode, name, B, B / dt, 1e3 * dt / B, t_first, err, mx), flush=True)
    del net
