#!/usr/bin/env python3
"""CAM passes (four scales, 8 flip pairs) at image sizes the shipped MIOpen database is and is not tuned for, by trunk layout
and MIOpen's deterministic attribute: which layout should an untuned size take now that the 1x1 convolutions are GEMMs?

    python tools/cam_layout_probe.py [--sizes 512x512,375x500,333x500,500x333,281x500] [--pairs 8]
Reference: step/make_cam.py:26-56 (one image + flip per pass there)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="512x512,375x500,333x500,500x333,281x500")
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    import torch
    from irn_amd.net import resnet50 as r50, resnet50_cam, weights
    from irn_amd.step import _common
    dev = torch.device("cuda", 0)
    _common.miopen_setup(0)
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(1))
    cam = cam.to(dev).eval()
    print("%-10s %-14s %-6s %-5s | %9s %9s  %s" % ("size", "layout", "gemm", "det", "ms/pass", "images/s", "tuned NHWC shape?"))
    for size in a.sizes.split(","):
        h, w = (int(v) for v in size.split("x"))
        xs = [torch.randn(2 * a.pairs, 3, int(round(h * s)), int(round(w * s)), device=dev) for s in (1.0, 0.5, 1.5, 2.0)]
        tuned = all((int(x.shape[0]), int(x.shape[2]), int(x.shape[3])) in r50.tuned_nhwc_shapes() for x in xs)
        ref = None
        for layout, gemm, det in (("0", 0, False), ("0", 0, True), ("1", 0, False), ("1", 1, False), ("1", 1, True)):
            r50.CHANNELS_LAST_MODE, r50.FUSED_GEMM, r50.DETERMINISTIC = layout, bool(gemm), None
            torch.backends.cudnn.deterministic = det
            with torch.no_grad():
                t0 = time.perf_counter()
                outs = [cam.forward_batch(x) for x in xs]             # first pass: MIOpen resolves its solvers
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    outs = [cam.forward_batch(x) for x in xs]
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.reps
                again = [cam.forward_batch(x) for x in xs]
                same = all(torch.equal(u, v) for u, v in zip(outs, again))
            if ref is None:
                ref = outs
            dev_max = max(float((u - v).abs().max() / v.abs().max()) for u, v in zip(outs, ref))
            print("%-10s %-14s %-6d %-5s | %9.2f %9.1f  %s  (first pass %.1f s; repeat bit-identical: %s; vs first row %.1e)" % (
                size, "channels-last" if layout == "1" else "NCHW", gemm, det, 1e3 * dt, a.pairs / dt, tuned, first, same, dev_max), flush=True)


if __name__ == "__main__":
    main()
