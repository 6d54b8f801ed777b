"""Feasibility probe for VERDICT r2 item 5: the 1x1 convolutions of the backbones (39 % of the e2e kernel time, fp32
Tensile GEMMs at ~122 TFLOP/s) as split-precision bf16 MFMA GEMMs with fp32 accumulation: x = x_hi + x_lo, w = w_hi + w_lo
(bf16 each), w.x ~ w_hi.x_hi + w_hi.x_lo + w_lo.x_hi as ONE GEMM over a 3x longer K.  Measures time and error against
fp64 at the shapes of a ResNet-50 trunk pass (8 images + flips at 512^2)."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


have_out_dtype = True
try:
    a = torch.randn(2, 8, 16, device=dev, dtype=torch.bfloat16)
    b = torch.randn(2, 16, 8, device=dev, dtype=torch.bfloat16)
    torch.bmm(a, b, out_dtype=torch.float32)
except Exception as e:
    have_out_dtype = False
    print("torch.bmm(out_dtype=float32) unavailable:", repr(e)[:200])

# (C_in, C_out, H*W) of 1x1 convolutions in a 16-image (8 + flips) pass at 512^2: stage 1-4 bottlenecks
shapes = [(64, 64, 128 * 128), (64, 256, 128 * 128), (256, 64, 128 * 128), (256, 128, 64 * 64), (128, 512, 64 * 64),
          (512, 256, 32 * 32), (256, 1024, 32 * 32), (1024, 512, 32 * 32), (512, 2048, 32 * 32), (2048, 512, 32 * 32)]
B = 16
tot32 = tot3 = 0.0
for ci, co, hw in shapes:
    x = torch.relu(torch.randn(B, ci, hw, device=dev))            # activations after ReLU
    w = torch.randn(co, ci, device=dev) * (2.0 / ci) ** 0.5
    ref = torch.matmul(w.double(), x.double())
    y32 = torch.matmul(w, x)
    e32 = ((y32.double() - ref).abs().max() / ref.abs().max()).item()
    t32 = bench(lambda: torch.matmul(w, x))
    xh, xl = split(x)
    wh, wl = split(w)
    w3 = torch.cat([wh, wh, wl], 1).contiguous()                  # [co, 3ci]
    x3 = torch.cat([xh, xl, xh], 1).contiguous()                  # [B, 3ci, hw]
    w3b = w3[None].expand(B, -1, -1)
    if have_out_dtype:
        f3 = lambda: torch.bmm(w3b, x3, out_dtype=torch.float32)
    else:
        f3 = lambda: torch.bmm(w3b, x3).float()
    y3 = f3()
    e3 = ((y3.double() - ref).abs().max() / ref.abs().max()).item()
    t3 = bench(f3)
    tsplit = bench(lambda: split(x))
    # the full product (four terms, 4x K) for the error floor of the split itself
    w4 = torch.cat([wh, wh, wl, wl], 1).contiguous()[None].expand(B, -1, -1)
    x4 = torch.cat([xh, xl, xh, xl], 1).contiguous()
    y4 = torch.bmm(w4, x4, out_dtype=torch.float32) if have_out_dtype else torch.bmm(w4, x4).float()
    e4 = ((y4.double() - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * B * ci * co * hw
    print("C %4d -> %4d, HW %5d: fp32 %.3f ms (%5.1f TF, err %.1e) | bf16x3 one GEMM %.3f ms (%5.1f TF-equiv, err %.1e; x4 err %.1e) | "
          "split of x alone %.3f ms" % (ci, co, hw, t32 * 1e3, fl / t32 / 1e12, e32, t3 * 1e3, fl / t3 / 1e12, e3, e4, tsplit * 1e3))
    tot32 += t32
    tot3 += t3 + tsplit
print("sum over the shapes: fp32 %.3f ms, bf16x3 + split %.3f ms (out_dtype path: %s)" % (tot32 * 1e3, tot3 * 1e3, have_out_dtype))
