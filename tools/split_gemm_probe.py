"""GPU pass of the round-6 gate on the last backbone lever: the stride-1 1x1 convolutions (hipBLASLt fp32 GEMMs, 44 % of the
`e2e` GPU time at 0.76-0.91 of the fp32 matrix peak) as fp16 hi/lo split products with fp32 accumulation.

    x = x_hi + 2^-11 x_lo',  x_hi = fp16(x),  x_lo' = fp16((x - x_hi) 2^11)          (22 mantissa bits together)
    x.w ~ x_hi.w_hi + 2^-11 (x_hi.w_lo' + x_lo'.w_hi)

`tools/bf16x3_cam_error.py` (CPU emulation, every 1x1 layer of the CAM network) puts this form at 5.4e-6 / 9.2e-6 from fp64 on
the normalised CAM — fp32 itself is 5.9e-6 / 7.6e-6, bf16x3 was 1.05e-4 / 1.56e-4 — so the accuracy gate (2e-5) passes.
This script measures what it buys per layer, in the product's layout (channels-last: the activation IS the row-major
[pixels, C_in] matrix), at the shapes of a 16-image (8 + flips) trunk pass at 512x512:
  fp32         torch.mm (what irn_conv1x1_nhwc's GEMM costs without its epilogue)
  two GEMMs    y = x_hi.w_hi ; y += 2^-11 [x_hi | x_lo'].[w_lo' | w_hi]        (the form of the emulation)
  one GEMM     y = 2^-p [x_hi | x_hi | x_lo].[w_hi | w_lo | w_hi] over 3 C_in, operands pre-scaled by fixed powers of two
               (x by 2^3, w by 2^p) instead of scaling the low parts: same bits unless a low part is subnormal
  split        producing [x_hi | x_lo'] from fp32 x as a pass of its own (what a producer that cannot emit the pair costs)
Errors are max |y - fp64| / max |fp64|.  Reference: net/resnet50.py:34-54 (Bottleneck)."""
import time

import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
S11 = 2.0 ** 11


def split16(t, scaled=True):
    hi = t.to(torch.float16)
    lo = ((t - hi.float()) * (S11 if scaled else 1.0)).to(torch.float16)
    return hi, lo


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def mm32(a, b):         # fp16 x fp16 -> fp32
    return torch.mm(a, b, out_dtype=torch.float32)


try:
    mm32(torch.randn(16, 32, device=dev, dtype=torch.float16), torch.randn(32, 16, device=dev, dtype=torch.float16))
    HAVE = True
except Exception as e:          # noqa: BLE001
    HAVE = False
    print("torch.mm(out_dtype=float32) unavailable: %r" % (e,))
    def mm32(a, b):             # noqa: F811
        return torch.mm(a, b).float()

# (C_in, C_out, pixels per image) of the stride-1 1x1 convolutions of a ResNet-50 trunk (strides 2,2,2,1) at 512x512
shapes = [(64, 64, 128 * 128), (64, 256, 128 * 128), (256, 64, 128 * 128), (256, 128, 64 * 64), (128, 512, 64 * 64),
          (512, 256, 32 * 32), (256, 1024, 32 * 32), (1024, 512, 32 * 32), (512, 2048, 32 * 32), (2048, 512, 32 * 32)]
B = 16
tot = {"fp32": 0.0, "two": 0.0, "one": 0.0, "split": 0.0}
print("%-22s | %-22s | %-30s | %-30s | %s" % ("C_in -> C_out, rows", "fp32", "two GEMMs (K + 2K)", "one GEMM (3K)", "split pass"))
for ci, co, hw in shapes:
    m = B * hw
    x = torch.relu(torch.randn(m, ci, device=dev)) * 2.0                 # activations after ReLU
    w = torch.randn(co, ci, device=dev) * (2.0 / ci) ** 0.5
    ref = torch.mm(x.double(), w.double().t())
    scale = ref.abs().max()
    wt = w.t().contiguous()
    y32 = torch.mm(x, wt)
    e32 = ((y32.double() - ref).abs().max() / scale).item()
    t32 = bench(lambda: torch.mm(x, wt))
    # two GEMMs, scaled low parts
    xh, xl = split16(x)
    wh, wl = split16(w)
    x2 = torch.cat([xh, xl], 1).contiguous()                             # [m, 2ci]
    w1t = wh.t().contiguous()                                            # [ci, co]
    w2t = torch.cat([wl, wh], 1).t().contiguous()                        # [2ci, co]
    def two():
        y = mm32(xh, w1t)
        return y.add_(mm32(x2, w2t), alpha=1.0 / S11)
    def two_gemms_only():
        mm32(xh, w1t)
        mm32(x2, w2t)
    y2 = two()
    e2 = ((y2.double() - ref).abs().max() / scale).item()
    t2 = bench(two_gemms_only)
    # one GEMM over 3K, fixed operand scales
    p = 8
    xs, ws = x * 8.0, w * 2.0 ** p
    xh1, xl1 = split16(xs, scaled=False)
    wh1, wl1 = split16(ws, scaled=False)
    x3 = torch.cat([xh1, xh1, xl1], 1).contiguous()
    w3t = torch.cat([wh1, wl1, wh1], 1).t().contiguous()
    y1 = mm32(x3, w3t) * (2.0 ** -(p + 3))
    e1 = ((y1.double() - ref).abs().max() / scale).item()
    t1 = bench(lambda: mm32(x3, w3t))
    def do_split():
        hi = x.to(torch.float16)
        lo = ((x - hi.float()) * S11).to(torch.float16)
        return torch.cat([hi, lo], 1)
    ts = bench(do_split)
    fl = 2.0 * m * ci * co
    print("%4d -> %4d, %7d | %.3f ms %5.1f TF %.1e | %.3f ms %6.1f TF-eq %.1e | %.3f ms %6.1f TF-eq %.1e | %.3f ms (ideal %.3f at 5 TB/s)" % (
        ci, co, m, t32 * 1e3, fl / t32 / 1e12, e32, t2 * 1e3, fl / t2 / 1e12, e2, t1 * 1e3, fl / t1 / 1e12, e1, ts * 1e3,
        m * ci * 8 / 5e12 * 1e3))
    tot["fp32"] += t32
    tot["two"] += t2
    tot["one"] += t1
    tot["split"] += ts
print("sum over the ten shapes: fp32 %.3f ms | two GEMMs %.3f | one GEMM %.3f | split passes %.3f (out_dtype path: %s)" % (
    tot["fp32"] * 1e3, tot["two"] * 1e3, tot["one"] * 1e3, tot["split"] * 1e3, HAVE))
