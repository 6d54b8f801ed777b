#!/usr/bin/env python3
"""Achieved HBM rate of the trunk's elementwise kernels (irn_bn_act, irn_stem_pool, irn_upsample_bilinear) at the shapes
of a batch of 8 image pairs at 512 x 512, beside the composed PyTorch ops they replace.
    python tools/epilogue_bench.py          (needs a GPU)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from irn_amd import ops

dev = torch.device("cuda", 0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def line(name, dt, nbytes, dt_ref):
    print("%-58s %7.3f ms  %6.2f TB/s  (= %4.1f %% of 8 TB/s)   composed torch ops: %7.3f ms  x%.2f" %
          (name, dt * 1e3, nbytes / dt / 1e12, 100 * nbytes / dt / 8e12, dt_ref * 1e3, dt_ref / dt), flush=True)


for shape in ((16, 256, 128, 128), (16, 512, 64, 64), (16, 2048, 32, 32), (16, 64, 125, 94)):
    x = torch.randn(shape, device=dev)
    res = torch.randn(shape, device=dev)
    c = shape[1]
    scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, var, w, b = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev), torch.randn(c, device=dev)
    nb = x.numel() * 4
    with torch.no_grad():
        line("bn_act %s" % (shape,), timed(lambda: ops.bn_act_(x, scale, shift, None, True)), 2 * nb,
             timed(lambda: F.relu(F.batch_norm(x, mean, var, w, b, False, 0.0, 1e-5), inplace=True)))
        line("bn_act + residual %s" % (shape,), timed(lambda: ops.bn_act_(x, scale, shift, res, True)), 3 * nb,
             timed(lambda: F.relu(F.batch_norm(x, mean, var, w, b, False, 0.0, 1e-5) + res, inplace=True)))
        line("bn_act + bn(residual) %s" % (shape,), timed(lambda: ops.bn_act_(x, scale, shift, res, True, (scale, shift))), 3 * nb,
             timed(lambda: F.relu(F.batch_norm(x, mean, var, w, b, False, 0.0, 1e-5) + F.batch_norm(res, mean, var, w, b, False, 0.0, 1e-5), inplace=True)))
for shape in ((16, 64, 256, 256), (16, 64, 512, 512)):
    x = torch.randn(shape, device=dev)
    c = shape[1]
    scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, var, w, b = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev), torch.randn(c, device=dev)
    with torch.no_grad():
        line("stem_pool %s" % (shape,), timed(lambda: ops.stem_pool(x, scale, shift)), int(x.numel() * 4 * 1.25),
             timed(lambda: F.max_pool2d(F.relu(F.batch_norm(x, mean, var, w, b, False, 0.0, 1e-5), inplace=True), 3, 2, 1)))
for shape, f in (((16, 256, 64, 64), 2), ((16, 256, 32, 32), 2), ((16, 32, 32, 32), 4)):
    x = torch.randn(shape, device=dev)
    up = torch.nn.Upsample(scale_factor=f, mode="bilinear", align_corners=False)
    with torch.no_grad():
        line("upsample x%d + relu %s" % (f, shape), timed(lambda: ops.upsample_bilinear(x, f, True)), x.numel() * 4 * (1 + f * f),
             timed(lambda: F.relu(up(x), inplace=True)))
