#!/usr/bin/env python3
"""Rank table of the split-precision GEMMs (irn_gemm16_nhwc): which entry of hipBLASLt's heuristic list is fastest per problem.

    python tools/gemm16_tune.py <out_dir> [--sizes 512x512,375x500,500x375] [--batch 8] [--write 1]

Collects every (m, k, cout, bias, residual, relu) problem `ops.gemm16_nhwc` is called with while the CAM network (four scales,
`--batch` flip pairs per pass) and IRNet run on the given image sizes, checks every candidate against the first one (same
operands: the results must agree to fp32 accumulation order), times it (HIP events) and writes the problems whose best entry beats
the first by more than 3 % into `ranks16` of irn_amd/data/gemm/<key>.json — data, not a timing made in the product's process.
Reference: net/resnet50.py:34-54."""
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
    ap.add_argument("--sizes", default="512x512,375x500,500x375")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--write", type=int, default=1)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    os.environ["IRN_CHANNELS_LAST"] = "1"
    os.environ["IRN_GEMM_TABLE"] = "0"
    import torch
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
    real = ops.gemm16_nhwc

    def spy(a16, b16, shape, bias=None, residual=None, relu=False, alpha=1.0, out=None, algo_rank=None):
        key = (int(a16.shape[0]), int(a16.shape[1]), int(b16.shape[0]), int(bias is not None), int(residual is not None), int(bool(relu)))
        problems[key] = problems.get(key, 0) + 1
        return real(a16, b16, shape, bias, residual, relu, alpha, out, algo_rank)

    ops.gemm16_nhwc = spy
    convs = {}
    real3 = ops.conv3x3_split

    def spy3(x, w16, alpha, algo_rank=None):
        n, c, h, w = (int(v) for v in x.shape)
        convs[(n, c, h, w, int(w16.shape[1]))] = convs.get((n, c, h, w, int(w16.shape[1])), 0) + 1
        return real3(x, w16, alpha, algo_rank)

    ops.conv3x3_split = spy3
    with torch.no_grad():
        for size in a.sizes.split(","):
            h, w = (int(v) for v in size.split("x"))
            for s in (1.0, 0.5, 1.5, 2.0):
                cam.forward_batch(torch.randn(2 * a.batch, 3, int(round(h * s)), int(round(w * s)), device=dev))
            irn.forward_batch([torch.randn(2, 3, h, w, device=dev) for _ in range(a.batch)])
    torch.cuda.synchronize()
    ops.gemm16_nhwc = real
    ops.conv3x3_split = real3
    lines = ["# split-precision GEMMs (fp16 operands, fp32 accumulation): ms per call for hipBLASLt's first pick and for the best of its list",
             "%9s %6s %5s %3s %3s %4s %5s | %8s %5s %8s | %9s" % ("m", "k", "cout", "b", "res", "relu", "calls", "gemm[0]", "best", "gemm[k]", "max|diff|")]
    ranks, tot0, totk = {}, 0.0, 0.0
    for key in sorted(problems):
        m, k, cout, hb, hr, relu = key
        g = torch.Generator(device=dev).manual_seed(m + k + cout)
        a16 = torch.randn(m, k, device=dev, generator=g).to(torch.float16)
        b16 = (torch.randn(cout, k, device=dev, generator=g) * 30).to(torch.float16)
        bias = torch.randn(cout, device=dev, generator=g) if hb else None
        res = torch.randn(1, cout, m, 1, device=dev, generator=g).contiguous(memory_format=torch.channels_last) if hr else None
        shape = (1, cout, m, 1)
        n_algo = ops.gemm16_algo_count(m, k, cout, hb, hr, relu)
        ref = ops.gemm16_nhwc(a16, b16, shape, bias, res, relu, 1e-3, algo_rank=0)
        times, diff = [], 0.0
        for r in range(n_algo):
            try:
                got = ops.gemm16_nhwc(a16, b16, shape, bias, res, relu, 1e-3, algo_rank=r)
                d = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
                diff = max(diff, d)
                if d > 1e-5:
                    times.append(float("inf"))
                    continue
                out = torch.empty_like(got)
                times.append(time_ms(lambda: ops.gemm16_nhwc(a16, b16, shape, bias, res, relu, 1e-3, out=out, algo_rank=r)))
            except Exception as ex:      # noqa: BLE001
                print("problem %s rank %d failed: %r" % (key, r, ex))
                times.append(float("inf"))
        best = min(range(n_algo), key=lambda r: times[r])
        if times[best] < 0.97 * times[0]:
            ranks[",".join(str(v) for v in key)] = best
        tot0 += times[0] * problems[key]
        totk += times[best if times[best] < 0.97 * times[0] else 0] * problems[key]
        lines.append("%9d %6d %5d %3d %3d %4d %5d | %8.4f %5d %8.4f | %9.2e" % (m, k, cout, hb, hr, relu, problems[key], times[0], best, times[best], diff))
    lines.append("# %d problems; all calls with the first pick %.3f ms, with the table %.3f ms; %d table entries" % (len(problems), tot0, totk, len(ranks)))
    # the row-fused 3x3 split convolutions: the rank applies to its three GEMMs (the heuristic list of the accumulating one)
    ranks3, t30, t3k = {}, 0.0, 0.0
    lines.append("# row-fused 3x3 split convolutions: (n, cin, h, w, cout), calls | ms with the first pick, best rank, ms with it")
    for (n, c, h, w, cout), calls in sorted(convs.items()):
        g = torch.Generator(device=dev).manual_seed(n + c + h + w)
        x = torch.relu(torch.randn(n, c, h, w, device=dev, generator=g)).contiguous(memory_format=torch.channels_last)
        w16, alpha = ops.split_weight_3x3(torch.randn(cout, c, 3, 3, device=dev, generator=g).double() * 0.02)
        m_pad = n * (h + 2) * (w + 2)
        n_algo = ops.gemm16_algo_count(m_pad, 9 * c, cout, 0, 1, 0)          # (an upper bound: the overlapping-row layout may offer fewer)
        inner = lambda t: t.view(n, h + 2, w + 2, cout)[:, 1:-1, 1:-1]
        ref = inner(ops.conv3x3_split(x, w16, alpha, algo_rank=0)).clone()
        times = []
        for r in range(n_algo):
            try:
                got = inner(ops.conv3x3_split(x, w16, alpha, algo_rank=r))
                if float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) > 1e-5:
                    times.append(float("inf"))
                    continue
                times.append(time_ms(lambda: ops.conv3x3_split(x, w16, alpha, algo_rank=r), n=10))
            except Exception as ex:      # noqa: BLE001
                times.append(float("inf"))
        best = min(range(n_algo), key=lambda r: times[r])
        if times[best] < 0.97 * times[0]:
            ranks3["%d,%d,%d" % (m_pad, c, cout)] = best
        t30 += times[0] * calls
        t3k += times[best if times[best] < 0.97 * times[0] else 0] * calls
        lines.append("%s %3d | %8.4f %3d %8.4f" % ((n, c, h, w, cout), calls, times[0], best, times[best]))
    lines.append("# %d 3x3 problems; all calls with the first pick %.3f ms, with the table %.3f ms; %d table entries" % (len(convs), t30, t3k, len(ranks3)))
    open(os.path.join(a.out, "gemm16_tune.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-12:]))
    if a.write:
        path = os.path.join(_common.gemm_table_root(), _common.miopen_cache_key() + ".json")
        table = json.load(open(path)) if os.path.exists(path) else {"ranks": {}}
        table["ranks16"] = ranks
        table["ranks3x3"] = ranks3
        json.dump(table, open(path, "w"), indent=0, sort_keys=True)
        json.dump(table, open(os.path.join(a.out, os.path.basename(path)), "w"), indent=0, sort_keys=True)      # travels back with gpurun_out
        print("wrote", path, "(copy in", a.out + ")")


if __name__ == "__main__":
    main()
