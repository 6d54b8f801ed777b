#!/usr/bin/env python3
"""Are the backbones' outputs a function of their inputs only — within a process, and from one process to the next?

    python tools/determinism_probe.py <out.json> [--sizes 96x128,113x150,512x512] [--pairs 1] [--repeat 3]

Runs CAM.forward_batch and EdgeDisplacement.forward_batch on seeded inputs `--repeat` times and records a BIT checksum
(sum of the int32 views) of every convolution / fused-GEMM output of the trunk, in call order.  Prints the first layer whose
checksum changes between repeats; writes the first repeat's checksums to <out.json> so that runs of OTHER processes
(another pid, a pool worker's environment, another MIOpen user database) can be compared with `--compare a.json b.json`.
The steps' claim under test: any worker layout writes the same files (reference step/make_cam.py:67-74)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(pa, pb):
    a, b = json.load(open(pa)), json.load(open(pb))
    bad = 0
    for tag in sorted(set(a) | set(b)):
        la, lb = a.get(tag, []), b.get(tag, [])
        first = next((i for i, (u, v) in enumerate(zip(la, lb)) if u[1] != v[1]), None)
        if len(la) != len(lb):
            print("%-28s %d vs %d layers recorded" % (tag, len(la), len(lb)))
            bad += 1
        elif first is None:
            print("%-28s identical bits in all %d layer outputs" % (tag, len(la)))
        else:
            n = sum(1 for u, v in zip(la, lb) if u[1] != v[1])
            print("%-28s %d of %d layer outputs differ; first: #%d %s" % (tag, n, len(la), first, la[first][0]))
            bad += 1
    return bad


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--compare":
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--sizes", default="96x128,113x150,512x512")
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--scales", default="1.0,0.5")
    ap.add_argument("--deterministic", type=int, default=-1, help="-1: the mode IRN_DETERMINISTIC selects; 0 / 1: the bare torch.backends.cudnn.deterministic flag, unmanaged")
    a = ap.parse_args()
    import torch
    import torch.nn as nn
    from irn_amd import ops
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    from irn_amd.step import _common
    dev = torch.device("cuda", 0)
    print("miopen db:", _common.miopen_setup(0), "find mode", os.environ.get("MIOPEN_FIND_MODE"),
          "| channels-last", os.environ.get("IRN_CHANNELS_LAST", "auto"), "| fused gemm", os.environ.get("IRN_FUSED_GEMM", "1"),
          "| IRN_DETERMINISTIC", os.environ.get("IRN_DETERMINISTIC", "1 (default)"))
    if a.deterministic >= 0:       # the bare MIOpen attribute, unmanaged (what rounds 1-4 could have switched on)
        from irn_amd.net import resnet50 as _r50
        _r50.DETERMINISTIC = None
        torch.backends.cudnn.deterministic = bool(a.deterministic)
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(1))
    cam = cam.to(dev).eval()
    irn = resnet50_irn.EdgeDisplacement()
    irn.load_state_dict(weights.random_irn_state(2), strict=False)
    irn = irn.to(dev).eval()

    rec = []

    def bits(t):
        return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())

    def hook(name):
        def fn(mod, inp, out):
            rec.append(("%s %s" % (name, tuple(out.shape)), bits(out)))
        return fn

    for net, tag in ((cam, "cam"), (irn, "irn")):
        for name, m in net.named_modules():
            if isinstance(m, nn.Conv2d):
                m.register_forward_hook(hook(tag + "." + name))
    real = ops.conv1x1_nhwc

    def spy(x, weight, *args, **kw):
        out = real(x, weight, *args, **kw)
        rec.append(("gemm %s->%d" % (tuple(x.shape), weight.shape[0]), bits(out)))
        return out

    ops.conv1x1_nhwc = spy
    real16 = ops.gemm16_nhwc

    def spy16(a16, b16, shape, *args, **kw):            # the split-precision form of the same layers (fp16 hi/lo operands)
        out = real16(a16, b16, shape, *args, **kw)
        rec.append(("gemm split %s x %d->%d" % (tuple(shape), a16.shape[1] // 3, b16.shape[0]), bits(out)))
        return out

    ops.gemm16_nhwc = spy16
    result = {}
    scales = [float(s) for s in a.scales.split(",")]
    with torch.no_grad():
        for size in a.sizes.split(","):
            h, w = (int(v) for v in size.split("x"))
            g = torch.Generator().manual_seed(h * 1000 + w)
            base = torch.rand(a.pairs, 1, 3, h, w, generator=g)
            runs = []
            for r in range(a.repeat):
                del rec[:]
                for s in scales:
                    hs, ws = int(round(h * s)), int(round(w * s))
                    xs = torch.nn.functional.interpolate(base.flatten(0, 1), size=(hs, ws), mode="bilinear", align_corners=False)
                    x = torch.stack([xs, xs.flip(-1)], 1).flatten(0, 1).to(dev)
                    out = cam.forward_batch(x)
                    rec.append(("cam output scale %g" % s, bits(out)))
                x0 = torch.stack([base[:, 0], base[:, 0].flip(-1)], 1).to(dev)
                for e, d in irn.forward_batch([x0[i] for i in range(a.pairs)]):
                    rec.append(("edge", bits(e)))
                    rec.append(("dp", bits(d)))
                runs.append(list(rec))
            tag = "%dx%d pairs %d" % (h, w, a.pairs)
            result[tag] = runs[0]
            for r in range(1, a.repeat):
                diff = [i for i, (u, v) in enumerate(zip(runs[0], runs[r])) if u[1] != v[1]]
                print("%-24s repeat %d vs 0: %s" % (tag, r, "identical bits in all %d outputs" % len(runs[0]) if not diff else
                                                  "%d of %d outputs differ; first: #%d %s" % (len(diff), len(runs[0]), diff[0], runs[0][diff[0]][0])))
    json.dump(result, open(a.out, "w"))


if __name__ == "__main__":
    main()
