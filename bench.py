#!/usr/bin/env python3
"""bench.py — throughput of the IRN pseudo-label hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (default `walk`, BASELINE.json configs[2], the case the north-star target is quoted on):
semantic pseudo-label generation for a batch of synthetic VOC12-shaped images — 512x512 images,
i.e. 128x128 stride-4 grids; per image an edge map [128,128] and K class activation maps
(K ~ VOC image-level label histogram: 60 % one class, 29 % two, 9 % three, 2 % four); affinity
random walk radius 10, beta 10, 2^8 = 256 sweeps; x4 bilinear upsample, /max, background 0.25,
argmax -> uint8 label map [512,512].  One "step" = that whole path for `--batch` images per GPU
with edge/CAM tensors already resident in HBM.  Images shard over ranks with no collective on the
data path (weak scaling: every rank processes its own `--batch` images per step).

One JSON line on rank 0: metric/value (images/s, whole job); `roofline` of the dominant kernel —
the weights-stationary resident walk by default (one launch = all sweeps of the batch), the
streaming sweep with --variant 1 (one launch = one sweep): algorithmic bytes per launch (SURVEY.md
§8d: weights streamed once per sweep) / HIP-event time of the launch on the launch stream vs the
8 TB/s HBM peak, plus the PMC-measured fabric traffic (profiles/traffic_walk.json) — and
`cpu_baseline` (oracle/walk_oracle.c, the fp64 C port of the same algorithm, timed on the host
cores at N=1 on 8 images).  Other legs: --workload walk_r5 | ins | coco | cam | e2e.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: fp32 vector (non-matrix) FMA peak, 256 CUs x 128 lanes x 2 flop x 2.4 GHz
N_DIRS = {5: 34, 10: 152}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="walk", choices=["walk", "walk_r5", "ins", "coco", "cam", "e2e"])
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = workload default)")
    ap.add_argument("--unique", type=int, default=96, help="distinct synthetic images per GPU")
    ap.add_argument("--variant", type=int, default=2, help="0 generic sweep, 1 blocked streaming sweep, 2 weights-stationary persistent walk")
    ap.add_argument("--xcd-map", type=int, default=1)
    ap.add_argument("--tile", type=int, default=8, help="sweep tile shape id (irn_walk_set_option 'tile')")
    ap.add_argument("--streams", type=int, default=1, help="channel-chunk classes on separate streams (merged=0)")
    ap.add_argument("--merged", type=int, default=0, help="all channel-chunk widths in one launch per sweep")
    ap.add_argument("--probe", type=int, default=0, help="diagnostic: time the weight-streaming skeleton instead")
    ap.add_argument("--walk-option", action="append", default=[], metavar="NAME=VALUE",
                    help="extra irn_walk_set_option settings (tuning experiments), e.g. poll_delay=8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=8)
    ap.add_argument("--json-out", default=None)
    return ap.parse_args()


WORKLOADS = {
    #            h    w    radius beta exp  out      default batch
    "walk":    (128, 128, 10, 10.0, 8, (512, 512), 192),     # BASELINE configs[2]
    "walk_r5": (128, 128, 5, 10.0, 8, (512, 512), 256),     # configs[0]'s operator setting at full batch
    "ins":     (128, 128, 10, 10.0, 8, (512, 512), 32),     # configs[3]: C*K instance channels
    "coco":    (256, 256, 10, 10.0, 8, (1024, 1024), 2),    # configs[4]: 80 classes, 1024^2
}


def make_inputs(workload, n_unique, seed0, device):
    from irn_amd import synth
    h, w, radius, beta, exp_times, out_hw, _ = WORKLOADS[workload]
    edges, cams, keys, insts, kinst = [], [], [], [], []
    for i in range(n_unique):
        seed = seed0 + i
        if workload == "coco":
            k = 80
        else:
            k = synth.voc_num_classes(seed)
        edges.append(torch.from_numpy(synth.edge_field(h, w, seed)).to(device))
        cams.append(torch.from_numpy(synth.cam_blobs(k, h, w, seed)).to(device))
        keys.append(torch.from_numpy(synth.voc_keys(min(k, 20), seed) if k <= 20 else np.arange(k)).to(device))
        if workload == "ins":
            ni = 1 + (seed % 4)                                    # 1-4 instances per image
            yy, xx = np.mgrid[0:h, 0:w]
            cmap = ((xx * ni) // w).astype(np.int32)               # vertical strips as stand-in clusters
            insts.append(torch.from_numpy(cmap).to(device))
            kinst.append(ni)
    return edges, cams, keys, insts, kinst


def algorithmic_bytes_per_sweep(shapes, n_dirs):
    """one sweep streams the |S| weight planes once, reads 1/deg, reads and writes
    the state: 4*N*(|S| + 1 + 2*C') bytes per image.  (This build keeps 1/deg in fp64, 8 B/pixel;
    the figure below uses the canonical 4 B so that fractions are comparable across builds.)"""
    return float(sum(4 * h * w * (n_dirs + 1 + 2 * c) for h, w, c in shapes))


def cpu_baseline(workload, n_images, seed0):
    """oracle/walk_oracle.c (fp64 C port of the same stencil algorithm, OpenMP over the host cores),
    rebuilt with -march=native on this box, timed on `n_images` images of the same workload."""
    from irn_amd import synth
    from oracle import build_oracle
    h, w, radius, beta, exp_times, out_hw, _ = WORKLOADS[workload]
    try:
        lib = build_oracle.load(native=True, out_dir="/tmp/irn_oracle_native")
    except Exception:
        lib = build_oracle.load(native=False)
    data = []
    for i in range(n_images):
        k = 80 if workload == "coco" else synth.voc_num_classes(seed0 + i)
        data.append((synth.cam_blobs(k, h, w, seed0 + i), synth.edge_field(h, w, seed0 + i)))
    build_oracle.walk(lib, data[0][0][:1], data[0][1], radius, beta, 2)          # warm
    t0 = time.perf_counter()
    for cam, edge in data:
        build_oracle.walk(lib, cam, edge, radius, beta, 2 ** exp_times)
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/s", "cores": int(lib.irn_oracle_threads()), "kind": "port",
            "sample": "%d images of the same workload (walk only, fp64 C stencil port oracle/walk_oracle.c, "
                      "OpenMP, %.1f s)" % (n_images, dt)}


def backbone_bench(a, rank, world, device, dist, parallel):
    """Secondary legs (no hand-written kernel dominates them, so no roofline object):
    `cam`  BASELINE configs[1]: multi-scale CAM inference — ResNet-50 CAM on {1.0,0.5,1.5,2.0}x512^2 with
           h-flip, merge to the stride-4 / full-resolution maps (step/make_cam.py:26-56), PyTorch-ROCm fp32.
    `e2e`  cam + EdgeDisplacement forward + random walk (radius 10, beta 10, 2^8) + label epilogue."""
    import torch.nn.functional as F
    from irn_amd import ops, synth
    from irn_amd.misc import indexing
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    from irn_amd.step import make_cam

    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    batch = a.batch or 8
    H = W = 512
    scales = (1.0, 0.5, 1.5, 2.0)
    cam_net = resnet50_cam.CAM()
    cam_net.load_state_dict(weights.random_cam_state(1))
    cam_net = cam_net.to(device).eval()
    # decoded uint8 images resident in HBM; the per-scale normalised (image, flip) pairs are built inside the timed
    # step by irn_msf_pack (Pillow-exact bicubic), like make_cam._work does for every loader item
    u8 = [torch.from_numpy(synth.photo(H, W, seed=1234 + rank * batch + i)).to(device) for i in range(batch)]
    labels = []
    for i in range(batch):
        lab = torch.zeros(20)
        lab[torch.from_numpy(synth.voc_keys(synth.voc_num_classes(i + 7), i + 7))] = 1
        labels.append(lab.to(device))
    irn = walker = None
    if a.workload == "e2e":
        irn = resnet50_irn.EdgeDisplacement()
        irn.load_state_dict(weights.random_irn_state(2), strict=False)
        irn = irn.to(device).eval()
        walker = indexing.RandomWalk(10, device)

    def cam_stage():
        packs = [ops.msf_pack(u8[i], scales) for i in range(batch)]         # per image: [2,3,Hs,Ws] per scale
        outs = []
        for si in range(len(scales)):
            x = torch.stack([p[si] for p in packs]).flatten(0, 1)             # [2B,3,Hs,Ws]: image, flip, image, ...
            f = F.relu(F.conv2d(cam_net.features(x), cam_net.classifier.weight))
            outs.append(f[0::2] + f[1::2].flip(-1))                           # [B,20,hs,ws]
        res = []
        for i in range(batch):
            res.append(make_cam.merge_scales([o[i] for o in outs], (H, W), labels[i]))
        return res, packs

    def step():
        with torch.no_grad():
            cams, packs = cam_stage()
            if a.workload == "cam":
                return cams
            edges = []
            for i in range(batch):
                e, _dp = irn(packs[i][0])
                edges.append(e)
            rws = walker(edges, [c[1] for c in cams], beta=10.0, exp_times=8)
            return ops.label_epilogue(rws, [(H, W)] * batch, 0.25, keys=[c[0] for c in cams])["labels"]

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dist, device)
    if rank == 0:
        res = {"metric": "images/sec for CAM+random-walk label gen, VOC12 512^2 (%s)" %
                         ("multi-scale CAM inference stage" if a.workload == "cam" else "CAM + IRNet + walk + labels, end to end"),
               "value": a.steps * batch * world / elapsed, "unit": "images/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s: synthetic 512x512 uint8 images resident in HBM, multi-scale inputs built on the GPU "
                                      "(irn_msf_pack), ResNet-50 CAM at scales %s + flip (random-init weights, fp32, MIOpen)%s" % (a.workload, scales, "" if a.workload == "cam" else
                                                           "; EdgeDisplacement forward; walk radius 10 beta 10 2^8; label epilogue"),
                          "images_per_gpu_per_step": batch},
               "roofline": None, "cpu_baseline": None}
        line = json.dumps(res)
        print(line, flush=True)
        if a.json_out:
            with open(a.json_out, "w") as f:
                f.write(line + "\n")
    if dist:
        dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from irn_amd import parallel
    dist = parallel.init_process_group(backend="nccl", device=device)      # nccl == RCCL on ROCm; None at N=1

    if a.workload in ("cam", "e2e"):
        return backbone_bench(a, rank, world, device, dist, parallel)

    from irn_amd import ops
    from irn_amd.misc import indexing

    h, w, radius, beta, exp_times, out_hw, default_batch = WORKLOADS[a.workload]
    batch = a.batch or default_batch
    n_unique = min(a.unique, batch)
    edges_u, cams_u, keys_u, insts_u, kinst_u = make_inputs(a.workload, n_unique, 1000 * (rank + 1), device)
    idx = [i % n_unique for i in range(batch)]
    edges = [edges_u[i] for i in idx]
    cams = [cams_u[i] for i in idx]
    keys = [keys_u[i] for i in idx]
    insts = [insts_u[i] for i in idx] if insts_u else None
    kinst = [kinst_u[i] for i in idx] if insts_u else None
    shapes = [(h, w, cams[i].shape[0] * (kinst[i] if kinst else 1)) for i in range(batch)]
    sizes = [out_hw] * batch

    walker = indexing.RandomWalk(radius, device)
    walker.set_option("variant", a.variant)
    walker.set_option("xcd_map", a.xcd_map)
    walker.set_option("tile", a.tile)
    walker.set_option("streams", a.streams)
    walker.set_option("merged", a.merged)
    walker.set_option("probe", a.probe)
    for kv in a.walk_option:
        name, value = kv.split("=")
        walker.set_option(name, int(value))
    walker.enable_timing(True)
    outs = [torch.empty((s[2], 1, h, w), device=device) for s in shapes]

    def step():
        rws = walker(edges, cams, beta=beta, exp_times=exp_times, inst_maps=insts, k_inst=kinst, outs=outs)
        if a.workload == "ins":
            return ops.label_epilogue(rws, sizes, 0.25, want_labels=False, want_argmax=True)["argmax"]
        return ops.label_epilogue(rws, sizes, 0.25, keys=keys)["labels"]

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if a.warmup > 0:
        walker.last_sweep_ms()                   # drop the warm-up steps' event pairs
    t0 = time.perf_counter()
    for _ in range(a.steps):
        labels = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dist, device)

    # HIP events recorded inside the timed region on the launch stream, read out after it
    sweep_ms, sweep_launches = walker.last_sweep_ms()
    walker.check()                               # resident walk: no tile gave up waiting
    checksum = int(sum(int(l.sum().item()) for l in labels[:4]))
    if rank == 0:
        n_dirs = N_DIRS[radius]
        per_sweep_bytes = algorithmic_bytes_per_sweep(shapes, n_dirs)
        n_sweeps = 2 ** exp_times
        avg_sweep_ms = sweep_ms / max(sweep_launches, 1)
        achieved = per_sweep_bytes / (avg_sweep_ms * 1e-3) / 1e9
        # one "launch" of the dominant kernel: the streaming variants launch once per sweep; the
        # weights-stationary walk is ONE launch for all 2^exp_times sweeps of the batch
        sweeps_per_launch = n_sweeps if a.variant == 2 else 1
        bytes_per_launch = per_sweep_bytes * sweeps_per_launch
        avg_launch_ms = avg_sweep_ms * sweeps_per_launch
        # the other ceiling: F = 2 * (2|S| + 1) flops per pixel, channel and sweep on the fp32 vector FMAs
        flops_per_launch = sweeps_per_launch * sum(2.0 * (2 * n_dirs + 1) * s[0] * s[1] * s[2] for s in shapes)
        fma_tflops = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == batch and tj.get("variant", 1) == a.variant:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "images/sec for CAM+random-walk label gen, VOC12 512^2 (random-walk label generation stage)",
            "value": a.steps * batch * world / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: VOC12-shaped %dx%d images (%dx%d stride-4 grids), affinity random walk "
                                   "radius=%d beta=%g 2^%d sweeps + x4 upsample/argmax label epilogue; K~VOC "
                                   "label histogram; inputs resident in HBM" %
                                   (a.workload, out_hw[0], out_hw[1], h, w, radius, beta, exp_times),
                       "images_per_gpu_per_step": batch, "sharding": "images strided over ranks, no collective",
                       "variant": a.variant, "tile": a.tile, "streams": a.streams, "merged": a.merged, "probe": a.probe, "mean_channels": float(np.mean([s[2] for s in shapes]))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("resident_kernel<%d> (weights-stationary persistent walk: one launch = all sweeps of the batch)" % radius)
                                   if a.variant == 2 else
                                   ("sweep_blocked_kernel<%d,CH> (one sweep over the batch = 1 launch per channel-chunk width)" % radius),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_ms,
                         "sweeps_per_launch": sweeps_per_launch, "launches_timed": sweep_launches // sweeps_per_launch,
                         "note": ("algorithmic bytes = SURVEY.md 8(d): weights streamed once per sweep, 4*N*(|S|+1+2C') per image "
                                  "and sweep.  This kernel keeps the weights in registers for all sweeps, so its HBM "
                                  "traffic (see `traffic`) is far below that figure and `frac` may exceed 1: it is faster than "
                                  "any kernel that re-reads the weights from HBM every sweep can be; it is bound by the "
                                  "tile-to-tile exchange latency and fp32 VALU issue, not by HBM") if a.variant == 2 else None,
                         "fp32_fma": {"achieved": fma_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": fma_tflops / FP32_VECTOR_PEAK_TFLOPS},
                         "sweep_share_of_step": sweep_ms / (1e3 * elapsed)},
            "label_checksum": checksum,
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_images, 1000)
        line = json.dumps(res)
        print(line, flush=True)
        if a.json_out:
            with open(a.json_out, "w") as f:
                f.write(line + "\n")
    walker.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
