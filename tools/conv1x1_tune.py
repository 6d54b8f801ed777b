#!/usr/bin/env python3
"""Per-layer-class A/B of the trunk's 1x1 convolutions and the rank table of their fused GEMMs.

    python tools/conv1x1_tune.py <out_dir> [--sizes 512x512,375x500,500x375] [--batch 8] [--write 1]

1. Collects every (m, cin, cout, bias, residual, relu) problem `ops.conv1x1_nhwc` is called with while the CAM network (four
   scales, `--batch` flip pairs per pass) and IRNet (`--batch` padded images) run channels-last on the given image sizes.
2. Per problem: checks the GEMM against the composed fp64 reference, times every entry of hipBLASLt's heuristic list (HIP
   events, 20 launches each) and, beside it, what the layer costs on MIOpen + `irn_bn_act_nhwc` (the round-4 path).
3. Writes `irn_amd/data/gemm/<device>-hip<version>.json` ({"ranks": {"m,cin,cout,bias,res,relu": k}} for the problems whose
   best entry is not the first, by more than 3 %) and a per-layer-class table (`<out_dir>/conv1x1_ab.txt`).
Reference: net/resnet50.py:11-60 (FixedBatchNorm + Bottleneck.forward)."""
import argparse
import json
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
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--sizes", default="512x512")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--write", type=int, default=1)
    ap.add_argument("--single", type=int, default=0)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    os.environ["IRN_CHANNELS_LAST"] = "1"
    os.environ["IRN_FUSED_GEMM"] = "1"
    os.environ["IRN_GEMM_TABLE"] = "0"
    import torch
    import torch.nn.functional as F
    from irn_amd import ops
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    from irn_amd.step import _common
    dev = torch.device("cuda", 0)
    _common.miopen_setup(0)
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(1))
    cam = cam.to(dev).eval()
    irn = resnet50_irn.EdgeDisplacement()
    irn.load_state_dict(weights.random_irn_state(2), strict=False)
    irn = irn.to(dev).eval()

    problems = {}
    real = ops.conv1x1_nhwc

    def spy(x, weight, bias=None, residual=None, relu=False, out=None, algo_rank=None):
        n, cin, h, w = x.shape
        key = (n * h * w, cin, int(weight.shape[0]), int(bias is not None), int(residual is not None), int(bool(relu)))
        problems[key] = problems.get(key, 0) + 1
        return real(x, weight, bias, residual, relu, out, algo_rank)

    ops.conv1x1_nhwc = spy
    with torch.no_grad():
        for size in a.sizes.split(","):
            h, w = (int(v) for v in size.split("x"))
            for b in sorted({a.batch} | ({1} if a.single else set())):
                for s in (1.0, 0.5, 1.5, 2.0):
                    cam.forward_batch(torch.randn(2 * b, 3, int(round(h * s)), int(round(w * s)), device=dev))
                irn.forward_batch([torch.randn(2, 3, h, w, device=dev) for _ in range(b)])
    torch.cuda.synchronize()
    ops.conv1x1_nhwc = real

    lines = ["# fused 1x1 convolution (hipBLASLt GEMM, bias/residual/ReLU in the epilogue) vs MIOpen NHWC convolution + irn_bn_act_nhwc",
             "# sizes %s, %d pairs per pass; ms per call; calls = per CAM(4 scales)+IRNet pass set" % (a.sizes, a.batch),
             "%9s %5s %5s %3s %3s %4s %5s | %8s %8s %8s | %8s %5s %7s | %9s" % (
                 "m", "cin", "cout", "b", "res", "relu", "calls", "miopen", "+bn_act", "sum", "gemm[0]", "best", "gemm[k]", "max|err|")]
    ranks = {}
    tot = {"miopen": 0.0, "gemm0": 0.0, "gemmk": 0.0}
    for key in sorted(problems):
        m, cin, cout, hb, hr, relu = key
        calls = problems[key]
        g = torch.Generator(device=dev).manual_seed(m + cin + cout)
        x2 = torch.randn(m, cin, device=dev, generator=g)
        wt = torch.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
        bias = torch.randn(cout, device=dev, generator=g) if hb else None
        res2 = torch.randn(m, cout, device=dev, generator=g) if hr else None
        # as [1, c, m, 1] channels-last tensors (the same matrices)
        x4 = x2.view(1, m, 1, cin).permute(0, 3, 1, 2)
        res4 = None if res2 is None else res2.view(1, m, 1, cout).permute(0, 3, 1, 2)
        want = x2.double() @ wt.double().t()
        if hb:
            want = want + bias.double()
        if hr:
            want = want + res2.double()
        if relu:
            want = want.clamp_min(0)
        n_algo = ops.conv1x1_algo_count(m, cin, cout, hb, hr, relu)
        times, err = [], 0.0
        for k in range(n_algo):
            try:
                got = ops.conv1x1_nhwc(x4, wt, bias, res4, relu, algo_rank=k)
                e = float((got.permute(0, 2, 3, 1).reshape(m, cout).double() - want).abs().max())
                err = max(err, e)
                if e > 1e-3:
                    times.append(float("inf"))
                    continue
                out = torch.empty_like(got)
                times.append(time_ms(lambda: ops.conv1x1_nhwc(x4, wt, bias, res4, relu, out=out, algo_rank=k)))
            except Exception as ex:
                print("problem %s rank %d failed: %r" % (key, k, ex))
                times.append(float("inf"))
        best = min(range(n_algo), key=lambda k: times[k])
        if times[best] < 0.97 * times[0]:
            ranks[",".join(str(v) for v in key)] = best
        # the round-4 path for the same layer: MIOpen convolution on the channels-last tensor, then one in-place pass
        w4 = wt.view(cout, cin, 1, 1)
        scale = torch.ones(cout, device=dev)
        shift = bias if hb else torch.zeros(cout, device=dev)
        # a 2-D image so that MIOpen sees a convolution problem like the network's (n = 1 here; the GEMM shape is what matters)
        side = 1
        for s in range(int(m ** 0.5), 0, -1):
            if m % s == 0:
                side = s
                break
        xi = x2.view(1, side, m // side, cin).permute(0, 3, 1, 2)
        ri = None if res2 is None else res2.view(1, side, m // side, cout).permute(0, 3, 1, 2)
        t_conv = time_ms(lambda: F.conv2d(xi, w4))
        y = F.conv2d(xi, w4).contiguous(memory_format=torch.channels_last)
        t_bn = time_ms(lambda: ops.bn_act_(y, scale, shift, ri, bool(relu)))
        lines.append("%9d %5d %5d %3d %3d %4d %5d | %8.4f %8.4f %8.4f | %8.4f %5d %7.4f | %9.2e" % (
            m, cin, cout, hb, hr, relu, calls, t_conv, t_bn, t_conv + t_bn, times[0], best, times[best], err))
        print(lines[-1], flush=True)
        tot["miopen"] += calls * (t_conv + t_bn)
        tot["gemm0"] += calls * times[0]
        tot["gemmk"] += calls * times[best]
    lines.append("# per pass set: MIOpen + bn_act %.2f ms, GEMM first pick %.2f ms, GEMM best rank %.2f ms" % (tot["miopen"], tot["gemm0"], tot["gemmk"]))
    print(lines[-1])
    with open(os.path.join(a.out, "conv1x1_ab.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    if a.write:
        dst = os.path.join(ROOT, "irn_amd", "data", "gemm")
        os.makedirs(dst, exist_ok=True)
        path = os.path.join(dst, _common.miopen_cache_key() + ".json")
        old = json.load(open(path))["ranks"] if os.path.exists(path) else {}
        old.update(ranks)
        json.dump({"ranks": old, "note": "rank in hipBLASLt's heuristic list per (m,cin,cout,bias,residual,relu); tools/conv1x1_tune.py"},
                  open(path, "w"), indent=0, sort_keys=True)
        # the GPU box's copy of the repo is scratch: what must come back goes through the output directory
        json.dump({"ranks": old}, open(os.path.join(a.out, "gemm_ranks_" + _common.miopen_cache_key() + ".json"), "w"), indent=0, sort_keys=True)
        print("rank table: %d problems with a non-default rank -> %s" % (len(old), path))


if __name__ == "__main__":
    main()
