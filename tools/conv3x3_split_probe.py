"""Would the stage-4 3x3 convolutions (512 -> 512 at 16 x 32 x 32: 48 % of the trunk's 3x3 flops) gain from the split-precision form
too?  A 3x3 convolution is a GEMM over K = 9 cin; with fp16 hi/lo operands K = 27 cin.  hipBLASLt has no implicit GEMM, so the
operand would have to be materialised (im2col of [hi | hi | lo']: 54 bytes per input element).  This measures the two halves
against MIOpen's fp32 convolution as the trunk runs it (channels-last, the process's find database):
    conv fp32     F.conv2d on the channels-last activation
    gemm16        the fp16 x fp16 -> fp32 GEMM [m, 27 cin] x [27 cin, cout] alone (operand given)
    im2col bytes  what writing + reading the operand costs at 5 TB/s
Shapes: stage 4 (512 planes, 32x32) and stage 3 (256 planes, 32x32) of a 16-row pass at 512x512."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd.step import _common  # noqa: E402

dev = torch.device("cuda", 0)
_common.miopen_setup(0)
torch.backends.cudnn.deterministic = False


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for planes, hw, rows in ((512, 32, 16), (256, 32, 16), (128, 64, 16), (512, 64, 16), (512, 16, 16)):
    x = torch.relu(torch.randn(rows, planes, hw, hw, device=dev)).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(planes, planes, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    t_conv = bench(lambda: F.conv2d(x, w, None, 1, 1))
    m, k = rows * hw * hw, 27 * planes
    a16 = torch.randn(m, k, device=dev, dtype=torch.float16)
    b16 = torch.randn(k, planes, device=dev, dtype=torch.float16)
    t_gemm = bench(lambda: torch.mm(a16, b16, out_dtype=torch.float32))
    fl = 2.0 * m * planes * planes * 9
    t_mem = 2.0 * m * k * 2 / 5e12
    print("3x3 %4d planes at %dx%dx%d: conv fp32 %.3f ms (%5.1f TF) | gemm16 over 27 cin %.3f ms (%6.1f TF-eq) + operand traffic %.3f ms at 5 TB/s -> %.3f ms" % (
        planes, rows, hw, hw, t_conv * 1e3, fl / t_conv / 1e12, t_gemm * 1e3, fl / t_gemm / 1e12, t_mem * 1e3, (t_gemm + t_mem) * 1e3))

# ---- the form that needs no im2col: on a zero-PADDED channels-last activation [N, H+2, W+2, C] tap (ky, kx) of a 3x3 / pad 1
# convolution is the same matrix shifted by (ky-1)(W+2) + (kx-1) rows, so the convolution is nine accumulating GEMMs (beta = 1)
# over K = 3 cin each on row-offset views of ONE split operand (border rows compute garbage that nobody reads)
from irn_amd import ops  # noqa: E402

print()
for planes, hw, rows in ((512, 32, 16), (512, 64, 16), (512, 16, 16), (512, 48, 16), (256, 32, 16)):
    hp = hw + 2
    m_pad, k = rows * hp * hp, 3 * planes
    guard = hp + 1
    buf = torch.randn(m_pad + 2 * guard, k, device=dev, dtype=torch.float16)
    wts = [torch.randn(planes, k, device=dev, dtype=torch.float16) for _ in range(9)]
    out = torch.empty((1, planes, m_pad, 1), device=dev, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
    offs = [(ky - 1) * hp + (kx - 1) for ky in range(3) for kx in range(3)]

    def nine():
        for t, off in enumerate(offs):
            a = buf[guard + off:guard + off + m_pad]
            ops.gemm16_nhwc(a, wts[t], (1, planes, m_pad, 1), residual=out if t else None, out=out, alpha=1e-3)

    t9 = bench(nine)
    x = torch.relu(torch.randn(rows, planes, hw, hw, device=dev)).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(planes, planes, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    t_conv = bench(lambda: F.conv2d(x, w, None, 1, 1))
    fl = 2.0 * rows * hw * hw * planes * planes * 9
    print("3x3 %4d planes at %dx%dx%d: conv fp32 %.3f ms (%5.1f TF) | nine accumulating split GEMMs on the padded operand %.3f ms (%6.1f TF-eq)" % (
        planes, rows, hw, hw, t_conv * 1e3, fl / t_conv / 1e12, t9 * 1e3, fl / t9 / 1e12))
