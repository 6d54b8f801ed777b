#!/usr/bin/env python3
"""MIOpen's own fused ops as PyTorch exposes them (torch.miopen_convolution_relu / _add_relu) against what the trunk runs:
MIOpen convolution + one in-place `irn_bn_act` pass, and (1x1 layers) the hipBLASLt GEMM with the epilogue fused.
Per layer class of the CAM network at 512^2, 8 flip pairs; folded batch norm (scale in the weight, shift as bias).

    python tools/miopen_fused_probe.py
Reference: net/resnet50.py:17-60 (Bottleneck)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, n=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    import torch
    import torch.nn.functional as F
    from irn_amd import ops
    from irn_amd.step import _common
    dev = torch.device("cuda", 0)
    _common.miopen_setup(0)
    n = 16
    # (name, cin, cout, k, spatial, residual)
    layers = [("layer1 conv1 1x1", 256, 64, 1, 128, False), ("layer1 conv2 3x3", 64, 64, 3, 128, False), ("layer1 conv3 1x1 + res", 64, 256, 1, 128, True),
              ("layer3 conv1 1x1", 1024, 256, 1, 32, False), ("layer3 conv2 3x3", 256, 256, 3, 32, False), ("layer3 conv3 1x1 + res", 256, 1024, 1, 32, True),
              ("layer4 conv2 3x3", 512, 512, 3, 32, False), ("layer4 conv3 1x1 + res", 512, 2048, 1, 32, True)]
    print("%-26s %-13s | %10s %12s %12s | %s" % ("layer (16 x C x S x S)", "layout", "conv+bn_act", "miopen fused", "GEMM fused", "max |fused - composed|"))
    for name, cin, cout, k, s, res in layers:
        for cl in (False, True):
            g = torch.Generator(device=dev).manual_seed(cin + cout + k)
            x = torch.randn(n, cin, s, s, device=dev, generator=g)
            w = torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5
            bias = torch.randn(cout, device=dev, generator=g)
            z = torch.randn(n, cout, s, s, device=dev, generator=g) if res else None
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
                z = None if z is None else z.contiguous(memory_format=torch.channels_last)
            one = torch.ones(cout, device=dev)
            pad = k // 2

            def composed():
                y = F.conv2d(x, w, None, 1, pad)
                return ops.bn_act_(y, one, bias, z, True)

            def fused():
                if res:
                    return torch.miopen_convolution_add_relu(x, w, z, 1.0, bias, [1, 1], [pad, pad], [1, 1], 1)
                return torch.miopen_convolution_relu(x, w, bias, [1, 1], [pad, pad], [1, 1], 1)

            t_c = time_ms(composed)
            ref = composed()
            try:
                got = fused()
                err = float((got - ref).abs().max())
                t_f = "%10.4f" % time_ms(fused)
            except Exception as e:
                t_f, err = "   failed", float("nan")
                print("   (%s: %s)" % (name, repr(e)[:160]))
            t_g = "           -"
            if cl and k == 1:
                w2 = w.flatten(1).contiguous()
                t_g = "%12.4f" % time_ms(lambda: ops.conv1x1_nhwc(x, w2, bias, z, True))
            print("%-26s %-13s | %10.4f %12s %12s | %.2e" % (name, "channels-last" if cl else "NCHW", t_c, t_f, t_g, err), flush=True)


if __name__ == "__main__":
    main()
