#!/usr/bin/env python3
"""bench.py — throughput of the IRN pseudo-label hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: under torch.distributed.run, or it starts its own N ranks;
                                                              --rank-devices 0,0 = two ranks on one GPU)

Workload (default `walk`, BASELINE.json configs[2], the case the north-star target is quoted on):
semantic pseudo-label generation for a batch of synthetic VOC12-shaped images — 512x512 images,
i.e. 128x128 stride-4 grids; per image an edge map [128,128] and K class activation maps
(K ~ VOC image-level label histogram: 60 % one class, 29 % two, 9 % three, 2 % four); affinity
random walk radius 10, beta 10, 2^8 = 256 sweeps; x4 bilinear upsample, /max, background 0.25,
argmax -> uint8 label map [512,512].  One "step" = that whole path for `--batch` images per GPU
with edge/CAM tensors already resident in HBM.  Images shard over ranks with no collective on the
data path (weak scaling: every rank processes its own `--batch` images per step).

One JSON line on rank 0:
  metric/value     images/s, whole job
  roofline         of the dominant kernel, `resident_kernel<10>` (one launch = the whole walk of the batch).  Its binding
                   ceiling is the fp32 vector FMA rate: `bound` = "fp32_vector", achieved/peak in TFLOP/s from HIP events
                   around the launch on its own stream.  `achieved` / `frac` count the flops the kernel EXECUTES:
                   2*(2|S|+1) per pixel, channel and operator application, times the applications of the schedule (round 3:
                   x.T^256 as an 84-term Chebyshev series, `schedule`).  `power_equivalent` is SURVEY.md §8(d)'s
                   F = 2*(2|S|+1)*N*C'*2^exp_times over the same time: what a kernel that applies the operator 2^exp_times
                   times would have to sustain for this throughput.  The kernel reads the weights from HBM once per image and
                   keeps them in registers, so the streaming-kernel byte formula of §8(d) does not describe it; that figure
                   is kept as `hbm_equivalent`, next to `traffic` (PMC-measured HBM bytes per launch, from profiles/).
  cpu_baseline     kind "port": oracle/walk_oracle.c (fp64 stencil port, one image per host thread) on a bounded
                   sample of the same workload; `label_parity`: the GPU's label maps of the first images of the batch
                   against the port's walk + the oracle's epilogue (pixels differing, every one checked to be a top-2 tie
                   below 1e-4); `reference_algorithm`: the reference's own dense algorithm (oracle/dense_ref.py =
                   misc/indexing.py:91-165 op for op on torch CPU) timed IN FULL at the workload's 128x128 grid (and at
                   64x64 as the scaling check), per-phase seconds, same thread count as the port.
  legs             runs of the other configurations at N = 1 (>= 10 steps or >= 96 images each): cam (configs[1]), e2e (cam +
                   IRNet + walk + labels), steps (the run_sample.py step API through `run(args)` on a synthetic VOC directory:
                   make_cam -> make_ins_seg_labels -> make_sem_seg_labels, 256 images, per-pass seconds), walk_r5, walk_plain (the
                   default workload with the plain 2^exp_times iteration, option accel = 0), ins / ins_r10 (configs[3]), coco
                   (configs[4]) — images/s each.
Other main workloads: --workload walk_r5 | ins | coco | cam | e2e | steps.
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
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: fp32 vector FMA peak, 256 CUs x 128 lanes x 2 flop x 2.4 GHz
FP32_MATRIX_PEAK_TFLOPS = 157.3   # same guide: dense fp32 MFMA peak (equal to the vector peak on gfx950)
N_DIRS = {5: 34, 10: 152}
METRIC = "images/sec for CAM+random-walk label gen, VOC12 512^2"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="walk", choices=["walk", "walk_r5", "walk_plain", "walk_voc", "walk_voc_r5", "ins", "ins_r10", "coco", "cam", "e2e", "cam_fp32", "e2e_fp32", "steps", "steps_voc"])
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = workload default)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic images per GPU (0 = the batch: no image twice in a step)")
    ap.add_argument("--variant", type=int, default=2, help="0 generic sweep, 1 blocked streaming sweep, 2 weights-stationary persistent walk")
    ap.add_argument("--walk-option", action="append", default=[], metavar="NAME=VALUE",
                    help="extra irn_walk_set_option settings (tuning experiments), e.g. poll_delay=8")
    ap.add_argument("--accel", type=int, default=1,
                    help="1: x.T^n as a truncated Chebyshev series (the product's default); 0: the plain n-fold iteration")
    ap.add_argument("--ref-grids", default="64,128",
                    help="grid sizes at which the reference's dense algorithm is timed on the host (the last one is reported; 128 = "
                         "the workload's own grid, ~3 minutes of CPU; '64' alone = quick runs, extrapolated)")
    ap.add_argument("--backend", default="auto", help="process group of the barrier / max-over-ranks: nccl (= RCCL), gloo, or auto = "
                                                      "RCCL when every rank's start-up probe completes inside its deadline, else gloo "
                                                      "(the data path has no collective; irn_amd/parallel.py)")
    ap.add_argument("--rank-devices", default="", metavar="D0,D1,...",
                    help="device ordinal of every local rank (default: LOCAL_RANK).  '0,0' = two ranks sharing GPU 0: the N > 1 path "
                         "on a one-GPU box (RCCL needs one device per rank, so `auto` then uses gloo)")
    ap.add_argument("--launch-timeout-s", type=float, default=0.0,
                    help="self-launched N > 1 runs: kill the launcher's process group after this many seconds (0 = no limit)")
    ap.add_argument("--allow-walk-fallback", action="store_true",
                    help="e2e / steps: record persistent walks that were re-run on the streaming sweeps instead of failing the run (ranks "
                         "sharing one device, a co-tenant on the GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-measure roofline.traffic (two short rocprofv3 --pmc passes of this script, ~40 s): use profiles/traffic_<workload>.json")
    ap.add_argument("--no-legs", action="store_true", help="skip the short secondary runs (cam, e2e, steps, walk_r5, ins, coco)")
    ap.add_argument("--legs", default="walk_voc,walk_voc_r5,walk_r5,walk_plain,ins,ins_r10,coco,cam,e2e,cam_fp32,e2e_fp32,steps,steps_voc")
    ap.add_argument("--legs-budget-s", type=float, default=240.0,
                    help="stop starting new legs once the legs have used this much wall time (the rest are recorded as skipped)")
    ap.add_argument("--loader-workers", type=int, default=4, help="DataLoader workers of the `steps` workload")
    ap.add_argument("--ins-blocking", action="store_true",
                    help="ins / ins_r10: wait for every batch's detections before the next batch is enqueued (no PCIe overlap)")
    ap.add_argument("--cpu-images", type=int, default=0, help="images of the CPU port sample (0 = 2 per host thread, at most 256)")
    ap.add_argument("--json-out", default=None)
    return ap.parse_args(argv)


WORKLOADS = {
    #            h    w    radius beta exp  out      default batch
    "walk":    (128, 128, 10, 10.0, 8, (512, 512), 192),     # BASELINE configs[2]
    "walk_plain": (128, 128, 10, 10.0, 8, (512, 512), 192),  # the same with the plain 2^exp_times iteration (accel = 0)
    "walk_r5": (128, 128, 5, 10.0, 8, (512, 512), 256),     # configs[0]'s operator setting at full batch
    "ins":     (128, 128, 5, 10.0, 8, (512, 512), 128),     # configs[3]: instance labels, radius 5 = the reference's call site (step/make_ins_seg_labels.py:135)
    "ins_r10": (128, 128, 10, 10.0, 8, (512, 512), 128),    # the same at SURVEY.md 8(d)'s row 4 radius
    "coco":    (256, 256, 10, 10.0, 8, (1024, 1024), 2),    # configs[4]: 80 classes, 1024^2
    # SURVEY.md 8(d) "ragged variant" of configs[2]: image sizes drawn from the VOC12 size histogram (synth.VOC_SIZES:
    # 500x375, 375x500, 500x333, ...), i.e. 94x125, 125x94, 84x125, ... grids; h, w here are only the nominal grid
    "walk_voc":    (128, 128, 10, 10.0, 8, None, 192),
    "walk_voc_r5": (128, 128, 5, 10.0, 8, None, 256),
}
WALK_WORKLOADS = ("walk", "walk_r5", "walk_plain", "coco", "walk_voc", "walk_voc_r5")


def image_geometry(workload, seed):
    """(grid h, grid w, (H, W)) of synthetic image `seed` of a workload."""
    from irn_amd import synth
    h, w, _, _, _, out_hw, _ = WORKLOADS[workload]
    if out_hw is None:
        out_hw = synth.voc_image_size(seed)
        h, w = synth.grid_of(out_hw)
    return h, w, out_hw


def make_inputs(workload, n_unique, seed0, device):
    from irn_amd import synth
    edges, cams, keys, dps = [], [], [], []
    for i in range(n_unique):
        seed = seed0 + i
        h, w, _ = image_geometry(workload, seed)
        k = 80 if workload == "coco" else synth.voc_num_classes(seed)
        edges.append(torch.from_numpy(synth.edge_field(h, w, seed)).to(device))
        cams.append(torch.from_numpy(synth.cam_blobs(k, h, w, seed)).to(device))
        keys.append(torch.from_numpy(synth.voc_keys(min(k, 20), seed) if k <= 20 else np.arange(k)).to(device))
        if workload in ("ins", "ins_r10"):
            dps.append(torch.from_numpy(synth.displacement_field(h, w, seed=seed, strength=0.3)).to(device))
    return edges, cams, keys, dps


def algorithmic_bytes_per_sweep(shapes, n_dirs):
    """one sweep of a STREAMING kernel reads the |S| weight planes once, 1/deg, and reads and
    writes the state: 4*N*(|S| + 1 + 2*C') bytes per image."""
    return float(sum(4 * h * w * (n_dirs + 1 + 2 * c) for h, w, c in shapes))


def flops_per_sweep(shapes, n_dirs):
    """F = 2*(2|S|+1) flops per pixel, channel and sweep."""
    return float(sum(2.0 * (2 * n_dirs + 1) * h * w * c for h, w, c in shapes))


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------

def _label_ties(got, want, up, bg, lut):
    """(#pixels differing, largest top-2 gap of the oracle's score stack among them, #of them that are NOT ties below 1e-4)."""
    diff = got != want
    n = int(diff.sum())
    if n == 0:
        return 0, 0.0, 0
    stack = np.concatenate([np.full((1,) + up.shape[1:], bg, np.float32), up], 0)[:, diff]
    srt = np.sort(stack, axis=0)
    gap = srt[-1] - srt[-2]
    return n, float(gap.max()), int((gap >= 1e-4).sum())


def host_cpu_model():
    """The host CPU's model string (the "x CPU" ratio moves by +-40 % from one GPU box to the next; this says which box)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(a, workload, n_images, seed0, gpu_labels=None):
    from irn_amd import synth
    from oracle import build_oracle, irn_oracle
    h, w, radius, beta, exp_times, out_hw, _ = WORKLOADS[workload]
    try:
        lib = build_oracle.load(native=True, out_dir="/tmp/irn_oracle_native")
    except Exception:
        lib = build_oracle.load(native=False)
    threads = int(lib.irn_oracle_threads())
    if n_images <= 0:
        n_images = max(8, min(256, 2 * threads))
    cams = [synth.cam_blobs(80 if workload == "coco" else synth.voc_num_classes(seed0 + i), h, w, seed0 + i) for i in range(n_images)]
    edges = [synth.edge_field(h, w, seed0 + i) for i in range(n_images)]
    build_oracle.walk_batch(lib, cams[:threads], edges[:threads], radius, beta, 2)            # warm (threads, pages)
    t0 = time.perf_counter()
    rws = build_oracle.walk_batch(lib, cams, edges, radius, beta, 2 ** exp_times)
    dt = time.perf_counter() - t0
    res = {"value": n_images / dt, "unit": "images/s", "cores": threads, "cpu_model": host_cpu_model(), "kind": "port",
           "sample": "%d images of the same workload (walk only: oracle/walk_oracle.c irn_oracle_walk_batch, fp64 stencil port of "
                     "misc/indexing.py:141-165, one image per OpenMP thread, rows vectorised; %.1f s)" % (n_images, dt)}
    res["which_is_which"] = {
        "value": "kind 'port': the SAME stencil algorithm the GPU runs, restated for the CPU (oracle/walk_oracle.c, fp64, plain "
                 "2^exp_times iteration) - the fair-algorithm baseline the >= 50x target is checked against",
        "reference_algorithm.value": "the north star's 'reference PyTorch CPU path': the reference's own dense algorithm "
                                     "(misc/indexing.py:91-165 restated op for op on torch CPU tensors, oracle/dense_ref.py, pinned "
                                     "bit-for-bit on outputs of /root/reference), timed in full at the workload's grid"}
    if gpu_labels:
        # by-product: the label maps the GPU wrote in the timed run against the port's walk + the oracle's epilogue
        # (step/make_sem_seg_labels.py:43-49) for the first images of the batch — measured, not a tolerance
        tot = {"images": 0, "pixels": 0, "pixels_differing": 0, "max_top2_gap": 0.0, "not_a_tie": 0}
        for i, (lab, keys) in enumerate(gpu_labels[:n_images]):
            up, want, _ = irn_oracle.sem_seg_epilogue(rws[i], out_hw, keys, 0.25)
            n_diff, gap, bad = _label_ties(lab, want, up, 0.25, np.concatenate([[0], keys + 1]))
            tot["images"] += 1
            tot["pixels"] += int(lab.size)
            tot["pixels_differing"] += n_diff
            tot["max_top2_gap"] = max(tot["max_top2_gap"], gap)
            tot["not_a_tie"] += bad
        tot["criterion"] = "a pixel may differ only where the two best entries of the oracle's normalised score stack are closer than 1e-4"
        res["label_parity"] = tot
    try:
        res["cam"] = cpu_baseline_cam(threads)
    except Exception as e:
        res["cam"] = {"error": repr(e)[:200]}
    try:
        res["reference_algorithm"] = reference_algorithm_baseline(radius, beta, exp_times, h, seed0, threads,
                                                                  [int(g) for g in a.ref_grids.split(",") if g])
    except Exception as e:                                   # never lose the bench line to the baseline
        res["reference_algorithm"] = {"error": repr(e)[:200]}
    return res


def cpu_baseline_cam(threads):
    """The CAM half of the metric on the host cores: the reference's `CAM.forward` (net/resnet50_cam.py:55-70, restated as
    plain torch functional ops in oracle/cam_ref.py and pinned bit for bit on the reference's own output) over the four
    scales + flips of ONE 512x512 image — what step/make_cam.py:26-37 runs per image before its merge — with seeded random
    weights of the architecture.  Bounded: one image (974 GFLOP), a few seconds on a host with many cores."""
    from irn_amd import synth
    from irn_amd.net import weights
    from oracle import cam_ref
    torch.set_num_threads(threads)
    sd = weights.random_cam_state(1)
    img = torch.from_numpy(synth.photo(512, 512, seed=1234)).permute(2, 0, 1).float().div_(255.0)
    scales = (1.0, 0.5, 1.5, 2.0)
    pairs = []
    for sc in scales:
        hs = int(round(512 * sc))
        x = torch.nn.functional.interpolate(img[None], (hs, hs), mode="bicubic", align_corners=False)[0]
        pairs.append(torch.stack([x, x.flip(-1)]))
    with torch.no_grad():
        cam_ref.cam_forward(sd, pairs[1])                                    # warm (threads, allocator): the 256^2 pair
        t0 = time.perf_counter()
        for x in pairs:
            cam_ref.cam_forward(sd, x)
        dt = time.perf_counter() - t0
    gflop = sum(cam_ref.flops_per_pair(int(round(512 * sc)), int(round(512 * sc))) for sc in scales) / 1e9
    return {"value": 1.0 / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "seconds_per_image": dt, "gflop_per_image": gflop, "host_tflops": gflop / dt / 1e3,
            "sample": "1 image of 512x512: CAM.forward on 4 scales x (image, flip), torch CPU fp32, %d threads, %.2f s "
                      "(oracle/cam_ref.py = net/resnet50_cam.py:55-70 + net/resnet50.py:17-108); compare with legs.cam" % (threads, dt)}


def reference_algorithm_baseline(radius, beta, exp_times, grid_target, seed0, threads, grids):
    """The reference's own DENSE algorithm (misc/indexing.py:91-165: dense (hw x hw) matrix, `exp_times` sgemm
    squarings) restated op for op on torch CPU tensors (oracle/dense_ref.py; the reference tree itself does not travel
    to the GPU box), timed on this box's host cores with the thread count the port used.  The workload's own grid
    (128x128: 70 TFLOP of sgemm, ~1 GB matrices) is run IN FULL; the smaller grid before it is the scaling check
    (set-up ~N^2, squarings ~N^3).  Only when the target grid is not in `grids` is the figure extrapolated, and says so."""
    from irn_amd import synth
    from oracle import dense_ref
    torch.set_num_threads(threads)
    measured = {}
    last = None
    for g in grids:
        cam = synth.cam_blobs(2, g, g, seed0)
        edge = synth.edge_field(g, g, seed0)
        tm = {}
        t0 = time.perf_counter()
        dense_ref.propagate_to_edge(cam, edge[None], radius, beta, exp_times, timings=tm)
        tm["total"] = time.perf_counter() - t0
        measured["%dx%d" % (g, g)] = {k: round(float(v), 3) for k, v in tm.items()}
        last = (g, tm)
    g, tm = last
    full = g == grid_target
    scale = float(grid_target * grid_target) / (g * g)
    est = tm["total"] if full else tm["setup"] * scale ** 2 + tm["transition"] * scale ** 3
    out = {"value": 1.0 / est, "unit": "images/s", "cores": threads,
           "kind": "reference algorithm (dense, torch CPU restatement oracle/dense_ref.py of misc/indexing.py:91-165)",
           "extrapolated": not full, "seconds_per_image": est, "seconds_measured": measured,
           "sample": "1 image (2 classes) per grid size, radius %d, exp_times=%d, %d torch threads; %s" %
                     (radius, exp_times, threads, "the %dx%d grid of the workload measured in full" % (g, g) if full else
                      "set-up ~N^2 and squarings ~N^3 extrapolated from %dx%d to %dx%d" % (g, g, grid_target, grid_target))}
    if full and len(grids) > 1:
        g0 = grids[-2]
        t0m = measured["%dx%d" % (g0, g0)]
        sc = float(g * g) / (g0 * g0)
        out["scaling_check"] = {"from": "%dx%d" % (g0, g0), "predicted_seconds": t0m["setup"] * sc ** 2 + t0m["transition"] * sc ** 3,
                                "measured_seconds": tm["total"]}
    return out


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------

def timed_loop(step, steps, warmup, dist, parallel, device, after_warmup=None, before_stop=None):
    """The contract: W untimed steps, then exactly K steps bracketed by barrier + synchronize, MAX over ranks.
    `before_stop` runs inside the timed region behind the last step (a pipelined step collects its last batch there)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if after_warmup:
        after_warmup()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    if before_stop:
        out = before_stop()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    return parallel.max_over_ranks(time.perf_counter() - t0, dist, device), out


def run_walk(a, workload, rank, world, device, dist, parallel, steps, warmup, batch=0):
    """walk / walk_r5 / coco: affinity build + random walk + label epilogue on resident tensors."""
    from irn_amd import ops
    from irn_amd.misc import indexing
    h, w, radius, beta, exp_times, out_hw, default_batch = WORKLOADS[workload]
    batch = batch or default_batch
    n_unique = min(a.unique or batch, batch)
    edges_u, cams_u, keys_u, _ = make_inputs(workload, n_unique, 1000 * (rank + 1), device)
    idx = [i % n_unique for i in range(batch)]
    edges, cams, keys = [edges_u[i] for i in idx], [cams_u[i] for i in idx], [keys_u[i] for i in idx]
    shapes = [(int(cams[i].shape[1]), int(cams[i].shape[2]), int(cams[i].shape[0])) for i in range(batch)]
    sizes = [image_geometry(workload, 1000 * (rank + 1) + i)[2] for i in idx]

    walker = indexing.RandomWalk(radius, device)
    walker.set_option("variant", a.variant)
    walker.set_option("accel", 0 if workload == "walk_plain" else a.accel)
    if warmup == 0:
        walker.set_option("poll_delay_auto", 0)          # no untimed step for the start-up probe to run in (profiler passes)
    for kv in a.walk_option:
        name, value = kv.split("=")
        walker.set_option(name, int(value))
    walker.enable_timing(True)
    outs = [torch.empty((s[2], 1, s[0], s[1]), device=device) for s in shapes]

    def step():
        rws = walker(edges, cams, beta=beta, exp_times=exp_times, outs=outs)
        return ops.label_epilogue(rws, sizes, 0.25, keys=keys)["labels"]

    elapsed, labels = timed_loop(step, steps, warmup, dist, parallel, device,
                                 after_warmup=(lambda: walker.last_sweep_ms()) if warmup > 0 else None)
    sweep_ms, sweep_launches = walker.last_sweep_ms()      # HIP events recorded inside the timed region on the launch stream
    walker.check()                                         # a launch that gave up its bounded wait invalidates the run
    checksum = int(sum(int(l.sum().item()) for l in labels[:4]))
    n_sweeps = 2 ** exp_times
    n_applied = walker.steps(n_sweeps)                     # operator applications the schedule spends on T^n_sweeps
    head = [(labels[i].cpu().numpy(), keys[i].cpu().numpy()) for i in range(min(8, n_unique))]   # for cpu_baseline.label_parity
    tuning = walker.tuning()
    fallback_runs = walker.fallback_runs
    walker.close()
    n_dirs = N_DIRS[radius]
    rounds = None
    if a.variant == 2:                                   # how the persistent launch packed the batch (host arithmetic)
        import ctypes
        from irn_amd import _lib
        nr = ctypes.c_int(0)
        arr = lambda v: _lib.i32_array([int(x) for x in v])
        n_wg = torch.cuda.get_device_properties(device).multi_processor_count
        if _lib.lib.irn_walk_plan_rounds(radius, batch, arr(s[0] for s in shapes), arr(s[1] for s in shapes), arr(s[2] for s in shapes),
                                         n_wg, tuning["placement"] or 1, None, 0, ctypes.byref(nr)) == 0:
            rounds = int(nr.value)
    avg_sweep_ms = sweep_ms / max(sweep_launches, 1)
    # one "launch" of the dominant kernel: the streaming variants launch once per operator application; the
    # weights-stationary walk is ONE launch for the whole walk of the batch
    sweeps_per_launch = n_applied if a.variant == 2 else 1
    avg_launch_ms = avg_sweep_ms * sweeps_per_launch
    flops_per_launch = flops_per_sweep(shapes, n_dirs) * sweeps_per_launch
    bytes_per_launch = algorithmic_bytes_per_sweep(shapes, n_dirs) * sweeps_per_launch
    return {"value": steps * batch * world / elapsed, "ms_per_step": 1e3 * elapsed / steps, "batch": batch,
            "shapes": shapes, "radius": radius, "beta": beta, "exp_times": exp_times, "out_hw": out_hw, "h": h, "w": w,
            "avg_launch_ms": avg_launch_ms, "sweeps_per_launch": sweeps_per_launch, "n_applied": n_applied, "n_sweeps": n_sweeps,
            "launches_timed": sweep_launches // max(sweeps_per_launch, 1), "flops_per_launch": flops_per_launch,
            "bytes_per_launch": bytes_per_launch, "sweep_share_of_step": sweep_ms / (1e3 * elapsed),
            "label_checksum": checksum, "head_labels": head, "tuning": tuning, "fallback_runs": fallback_runs, "rounds": rounds,
            "grid_pixels": int(sum(s[0] * s[1] for s in shapes)), "sizes": sizes}


def run_ins(a, workload, rank, world, device, dist, parallel, steps, warmup, batch=0):
    """configs[3]: instance pseudo-labels from resident edge / displacement / CAM tensors — centroid refinement,
    clustering, per-instance CAM split + random walk, label epilogue, per-mask connected components, and the transfer
    of the detections to the host (step/make_ins_seg_labels.py:131-152 for every image of the batch)."""
    from irn_amd import synth
    from irn_amd.misc import indexing
    from irn_amd.step import make_ins_seg_labels as mis
    h, w, radius, beta, exp_times, out_hw, default_batch = WORKLOADS[workload]
    batch = batch or default_batch
    n_unique = min(a.unique or batch, batch)
    edges_u, cams_u, keys_u, dps_u = make_inputs(workload, n_unique, 1000 * (rank + 1), device)
    items = [{"edge": edges_u[i % n_unique][None], "dp": dps_u[i % n_unique], "cam": cams_u[i % n_unique],
              "keys": keys_u[i % n_unique].cpu(), "size": out_hw} for i in range(batch)]
    walker = indexing.RandomWalk(radius, device)
    for kv in a.walk_option:
        name, value = kv.split("=")
        walker.set_option(name, int(value))

    # as the step's loop does it (make_ins_seg_labels._flush): a batch is enqueued, then the batch before it is collected —
    # its masks cross PCIe under this batch's kernels.  Every batch enqueued inside the timed region is collected inside it.
    in_flight = [None]

    def step():
        cur = mis.instance_labels_batch(walker, items, beta, exp_times, 0.25, deferred=not a.ins_blocking)
        if a.ins_blocking:
            return cur
        prev, in_flight[0] = in_flight[0], cur
        return prev.result() if prev is not None else None

    def collect_last():
        prev, in_flight[0] = in_flight[0], None
        return prev.result() if prev is not None else None

    def after_warmup():
        collect_last()

    elapsed, dets = timed_loop(step, steps, warmup, dist, parallel, device, after_warmup=after_warmup,
                               before_stop=None if a.ins_blocking else collect_last)
    n_det = sum(0 if isinstance(d, Exception) else len(d["score"]) for d in dets)
    n_fallback = walker.fallback_runs
    walker.close()
    return {"value": steps * batch * world / elapsed, "ms_per_step": 1e3 * elapsed / steps, "batch": batch,
            "detections_per_image": n_det / float(batch), "fallback_runs": n_fallback,
            "radius": radius, "beta": beta, "exp_times": exp_times, "out_hw": out_hw, "h": h, "w": w}


def run_backbone(a, workload, rank, world, device, dist, parallel, steps, warmup, batch=0):
    """`cam`  BASELINE configs[1]: multi-scale CAM inference — ResNet-50 CAM on {1.0,0.5,1.5,2.0}x512^2 with
           h-flip, merge to the stride-4 / full-resolution maps (step/make_cam.py:26-56), PyTorch-ROCm fp32.
    `e2e`  cam + EdgeDisplacement forward + random walk (radius 10, beta 10, 2^8) + label epilogue.
    No hand-written kernel dominates these legs (MIOpen convolutions do), so they carry no roofline object."""
    from irn_amd import ops, synth
    from irn_amd.misc import indexing
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    from irn_amd.step import make_cam

    from irn_amd.step import _common
    _common.miopen_setup(device.index or 0)     # what the steps run with: find mode 2, the stable / shipped find database
    batch = batch or 8
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
        labels.append(lab)            # host-side, like the loader hands it over: the merge then never waits for the device
    irn = walker = None
    if workload == "e2e":
        irn = resnet50_irn.EdgeDisplacement()
        irn.load_state_dict(weights.random_irn_state(2), strict=False)
        irn = irn.to(device).eval()
        walker = indexing.RandomWalk(10, device)

    def step():
        with torch.no_grad():
            packs = [ops.msf_pack(u8[i], scales) for i in range(batch)]         # per image: [2,3,Hs,Ws] per scale
            outs = [cam_net.forward_batch(torch.cat([p[si] for p in packs])) for si in range(len(scales))]
            cams = [make_cam.merge_scales([o[i] for o in outs], (H, W), labels[i]) for i in range(batch)]
            if workload == "cam":
                return cams
            edges = [e for e, _dp in irn.forward_batch([p[0] for p in packs])]
            rws = walker(edges, [c[1] for c in cams], beta=10.0, exp_times=8)
            return ops.label_epilogue(rws, [(H, W)] * batch, 0.25, keys=[c[0] for c in cams])["labels"]

    elapsed, _ = timed_loop(step, steps, warmup, dist, parallel, device)
    _common.check_split_overflow("bench.py %s" % workload)        # an activation beyond fp16's range would invalidate the run
    out = {"value": steps * batch * world / elapsed, "ms_per_step": 1e3 * elapsed / steps, "batch": batch, "scales": scales}
    # convolution flops per image (counted on meta tensors: nothing is computed) against the fp32 matrix peak: the backbones
    # are library kernels (MIOpen, hipBLASLt), so this is a utilisation figure, not a roofline claim of a kernel of ours — and
    # an fp32-EQUIVALENT one: with the split-precision 1x1 convolutions part of the work runs on the fp16 matrix pipe (3 fp16
    # flops per fp32 flop counted here), which is how it can approach 1
    try:
        gflop = backbone_gflop_per_image(scales, H, workload == "e2e")
        out["gflop_per_image"] = gflop
        out["matrix_fp32_frac"] = gflop * 1e9 * (steps * batch / elapsed) / (FP32_MATRIX_PEAK_TFLOPS * 1e12)
    except Exception as e:                  # noqa: BLE001
        out["gflop_per_image"] = {"error": repr(e)[:120]}
    # which trunk ran (a silent fall-back to NCHW / to the unfused 1x1 convolutions would just look slow)
    from irn_amd.net import resnet50 as _r50
    with torch.no_grad():
        probe = torch.empty(2 * batch, 3, H, W, device=device)
        cl = _r50.channels_last_for(probe)
    out["trunk"] = {"layout": "channels_last" if cl else "nchw", "fused_1x1_gemm": bool(cl and _r50.FUSED_GEMM and _r50.FUSED_EPILOGUE),
                    "split_precision_1x1": bool(cl and _r50.FUSED_GEMM and _r50.FUSED_EPILOGUE and _r50.SPLIT_GEMM),
                    "arithmetic": "fp32" + ("; 1x1 convolutions of the bottlenecks as fp16 hi/lo split products (2 x 11-bit operands, fp32 accumulation: "
                                            "as close to fp64 as the fp32 GEMM, tests/test_gpu_split_gemm.py); the *_fp32 legs run without them"
                                            if (cl and _r50.FUSED_GEMM and _r50.FUSED_EPILOGUE and _r50.SPLIT_GEMM) else ""),
                    "deterministic": bool(_r50.DETERMINISTIC), "tuned_nhwc_shapes": len(_r50.tuned_nhwc_shapes()),
                    "miopen_db": os.environ.get("MIOPEN_USER_DB_PATH"), "miopen_key": _common.miopen_mode_key()}
    if walker is not None:
        walker.check()                  # raises when a persistent launch gave up (nothing here calls sync(), so nothing re-ran)
        out["walk_fallback_runs"] = walker.fallback_runs
        walker.close()
        if out["walk_fallback_runs"] and not a.allow_walk_fallback:
            raise RuntimeError("e2e leg: %d walk batch(es) fell back to the streaming sweeps" % out["walk_fallback_runs"])
    return out


_GFLOP = {}


def backbone_gflop_per_image(scales, size, with_irn):
    """Multiply-add flops (2 per MAC) of the product's own networks per image: CAM on every scale x (image, flip), plus the
    EdgeDisplacement pass on the 512x512 crop pair for `e2e` — 974.0 + 149.6 GFLOP.  Counted by PyTorch's flop counter on META
    tensors (shapes only, no arithmetic, no GPU)."""
    key = (tuple(scales), size, with_irn)
    if key not in _GFLOP:
        from torch.utils.flop_counter import FlopCounterMode
        from irn_amd.net import resnet50_cam, resnet50_irn
        total = 0
        with torch.no_grad():
            with torch.device("meta"):
                cam = resnet50_cam.CAM().eval()
                irn = resnet50_irn.EdgeDisplacement().eval() if with_irn else None
            for sc in scales:
                hs = int(round(size * sc))
                with FlopCounterMode(display=False) as fc:
                    cam(torch.empty(2, 3, hs, hs, device="meta"))
                total += fc.get_total_flops()
            if irn is not None:
                with FlopCounterMode(display=False) as fc:
                    irn(torch.empty(2, 3, size, size, device="meta"))
                total += fc.get_total_flops()
        _GFLOP[key] = total / 1e9
    return _GFLOP[key]


def run_steps(a, rank, world, device, dist, parallel, steps, warmup, batch=0, voc_sizes=False):
    """The drop-in step API itself, in run_sample.py's order (reference run_sample.py:91-131): make_cam.run(args) ->
    make_ins_seg_labels.run(args) -> make_sem_seg_labels.run(args) on a synthetic VOC-shaped directory of 512x512 JPEGs
    with random-init checkpoints: DataLoader + JPEG decode, multi-scale CAM, CAM hand-off, IRNet, walk, detections, PNG /
    .npy writing.  The semantic step walks at radius 10 (= BASELINE configs[2]), the instance step at the reference's own
    radius 5 (step/make_ins_seg_labels.py:135).  One "step" = the three passes over `batch` images."""
    import shutil
    import tempfile
    from PIL import Image
    from irn_amd import synth
    from irn_amd.net import weights
    from irn_amd.step import _common, make_cam, make_ins_seg_labels, make_sem_seg_labels
    _common.miopen_setup(device.index or 0)
    batch = batch or 32
    t_setup = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="irn_steps_%d_" % rank)
    try:
        root = os.path.join(tmp, "voc")
        os.makedirs(os.path.join(root, "JPEGImages"))
        names, labels = [], {}
        for i in range(batch):
            name = "2009_%06d" % (rank * batch + i + 1)
            ih, iw = synth.voc_image_size(7000 + rank * batch + i) if voc_sizes else (512, 512)
            Image.fromarray(synth.photo(ih, iw, seed=7000 + rank * batch + i)).save(
                os.path.join(root, "JPEGImages", name + ".jpg"), quality=92)
            lab = np.zeros(20, np.float32)
            lab[synth.voc_keys(synth.voc_num_classes(i + 11), i + 11)] = 1
            names.append(name)
            labels[int(name.replace("_", ""))] = lab
        with open(os.path.join(tmp, "train.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        np.save(os.path.join(tmp, "cls_labels.npy"), labels)
        torch.save(weights.random_cam_state(1), os.path.join(tmp, "res50_cam.pth"))
        torch.save(weights.random_irn_state(2), os.path.join(tmp, "res50_irn.pth"))
        args = argparse.Namespace(
            num_workers=int(a.loader_workers), voc12_root=root, train_list=os.path.join(tmp, "train.txt"),
            infer_list=os.path.join(tmp, "train.txt"), cam_network="net.resnet50_cam",
            cam_weights_name=os.path.join(tmp, "res50_cam"), cam_scales=(1.0, 0.5, 1.5, 2.0), irn_network="net.resnet50_irn",
            irn_weights_name=os.path.join(tmp, "res50_irn.pth"), beta=10, exp_times=8, sem_seg_bg_thres=0.25,
            ins_seg_bg_thres=0.25, cam_out_dir=os.path.join(tmp, "cam"), sem_seg_out_dir=os.path.join(tmp, "sem"),
            ins_seg_out_dir=os.path.join(tmp, "ins"), radius=10, walk_batch=64)
        for d in (args.cam_out_dir, args.sem_seg_out_dir, args.ins_seg_out_dir):
            os.makedirs(d)
        # the steps shard over the visible GPUs by themselves (one worker process per GPU); inside a torch.distributed job
        # every rank must stay on its own device, so each rank runs the steps with a one-entry device list (in-process)
        args.worker_devices = str(device.index or 0)

        class _Quiet:
            def write(self, s):
                return len(s)

            def flush(self):
                pass

        passes = []

        def step():
            real = sys.stdout
            sys.stdout = _Quiet()                      # progress ticks of the steps would break the one-JSON-line contract
            try:
                # the step API itself, as run_sample.py calls it (checkpoints loaded, shards made, workers run, files written).
                # Every pass starts like a fresh run_sample.py invocation as far as results of earlier passes go: the edge /
                # displacement maps an earlier pass left on the device are dropped, so that only the hand-off INSIDE a pass
                # (make_ins_seg_labels -> make_sem_seg_labels) is measured, never one across passes
                _common.EDGE_STORE.clear()
                t0 = time.perf_counter()
                make_cam.run(args)
                t1 = time.perf_counter()
                args.radius, args.walk_batch = 5, 32       # the reference's instance call site
                make_ins_seg_labels.run(args)
                t2 = time.perf_counter()
                args.radius, args.walk_batch = 10, 64      # configs[2]
                make_sem_seg_labels.run(args)
                t3 = time.perf_counter()
                passes.append({"make_cam": t1 - t0, "make_ins_seg_labels": t2 - t1, "make_sem_seg_labels": t3 - t2})
            finally:
                sys.stdout = real

        t_setup = time.perf_counter() - t_setup
        hits0, misses0 = _common.CAM_STORE.hits, _common.CAM_STORE.misses
        e_hits0 = _common.EDGE_STORE.hits
        for _ in range(warmup):                      # (here, not in timed_loop: the counters below must cover the timed passes only)
            step()
        fb0 = _common.WALK_STATS["fallback_runs"]
        cam0 = dict(_common.CAM_STATS)
        elapsed, _ = timed_loop(step, steps, 0, dist, parallel, device)
        fallback_runs = _common.WALK_STATS["fallback_runs"] - fb0
        if fallback_runs and not a.allow_walk_fallback:
            # a persistent walk that lost its bounded wait is re-run on the streaming sweeps (11x slower): correct files, but
            # not a measurement of the default path
            raise RuntimeError("steps leg: %d walk batch(es) fell back to the streaming sweeps inside the timed passes" % fallback_runs)
        n_png = len([f for f in os.listdir(args.sem_seg_out_dir) if f.endswith(".png")])
        n_ins = len([f for f in os.listdir(args.ins_seg_out_dir) if f.endswith(".npy")])
        if n_png != batch:
            raise RuntimeError("steps leg: %d label maps written for %d images" % (n_png, batch))
        return {"value": steps * batch * world / elapsed, "ms_per_step": 1e3 * elapsed / steps, "batch": batch,
                "cam_store_hits": _common.CAM_STORE.hits - hits0, "cam_store_misses": _common.CAM_STORE.misses - misses0,
                "edge_store_hits": _common.EDGE_STORE.hits - e_hits0, "loader_workers": args.num_workers, "instance_files": n_ins,
                "walk_fallback_runs": fallback_runs, "voc_sizes": bool(voc_sizes),
                "cam_trunk_passes": {k: _common.CAM_STATS[k] - cam0[k] for k in cam0},
                "through": "make_cam.run(args) + make_ins_seg_labels.run(args) + make_sem_seg_labels.run(args)",
                "pass_seconds": [{k: round(v, 3) for k, v in p.items()} for p in passes[warmup:]], "setup_seconds": t_setup}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------

def describe(workload, r):
    if workload in WALK_WORKLOADS and r.get("out_hw"):
        return ("%s: VOC12-shaped %dx%d images (%dx%d stride-4 grids), affinity random walk radius=%d beta=%g 2^%d sweeps "
                "+ x4 upsample/argmax label epilogue; K~VOC label histogram%s; inputs resident in HBM" %
                (workload, r["out_hw"][0], r["out_hw"][1], r["h"], r["w"], r["radius"], r["beta"], r["exp_times"],
                 " (80 classes)" if workload == "coco" else ""))
    if workload in ("walk_voc", "walk_voc_r5"):
        import collections
        top = collections.Counter((s[0], s[1]) for s in r["shapes"]).most_common(4)
        return ("%s: ragged variant of configs[2] (SURVEY.md 8d): image sizes drawn from the VOC12 size histogram (synth.VOC_SIZES), "
                "grids %s ..., affinity random walk radius=%d beta=%g 2^%d sweeps + x4 upsample/argmax label epilogue cropped to each "
                "image; K~VOC label histogram; inputs resident in HBM" %
                (workload, ", ".join("%dx%d (%d)" % (g[0], g[1], n) for g, n in top), r["radius"], r["beta"], r["exp_times"]))
    if workload in ("ins", "ins_r10"):
        return ("ins: VOC12-shaped %dx%d images (%dx%d grids), instance labels: displacement-field centroids + clustering + "
                "per-instance random walk radius=%d beta=%g 2^%d + epilogue + connected-component detections copied to the "
                "host; edge / displacement / CAM tensors resident in HBM" %
                (r["out_hw"][0], r["out_hw"][1], r["h"], r["w"], r["radius"], r["beta"], r["exp_times"]))
    if workload == "steps_voc":
        return ("steps_voc: run_sample.py step API (make_cam -> make_ins_seg_labels at radius 5 -> make_sem_seg_labels at radius 10) "
                "on a synthetic VOC directory of JPEGs whose sizes follow the VOC12 size histogram (synth.VOC_SIZES: 500x375, 375x500, "
                "500x333, ...), random-init checkpoints, files written")
    if workload == "steps":
        return ("steps: run_sample.py step API (make_cam -> make_ins_seg_labels at radius 5 -> make_sem_seg_labels at radius 10) "
                "on a synthetic VOC directory of 512x512 JPEGs, random-init checkpoints, files written")
    return ("%s: synthetic 512x512 uint8 images resident in HBM, multi-scale inputs built on the GPU (irn_msf_pack), ResNet-50 CAM "
            "at scales %s + flip (random-init weights, fp32, MIOpen)%s" %
            (workload, r.get("scales"), "" if workload.startswith("cam") else "; EdgeDisplacement forward; walk radius 10 beta 10 2^8; label epilogue"))


def run_workload(a, workload, rank, world, device, dist, parallel, steps, warmup, batch=0):
    if workload.endswith("_fp32"):
        # the backbone legs with the split-precision 1x1 convolutions switched off (IRN_SPLIT_GEMM=0): plain fp32 GEMMs everywhere
        from irn_amd.net import resnet50 as _r50
        saved, _r50.SPLIT_GEMM = _r50.SPLIT_GEMM, False
        try:
            return run_backbone(a, workload[:-5], rank, world, device, dist, parallel, steps, warmup, batch)
        finally:
            _r50.SPLIT_GEMM = saved
    if workload in WALK_WORKLOADS:
        return run_walk(a, workload, rank, world, device, dist, parallel, steps, warmup, batch)
    if workload in ("ins", "ins_r10"):
        return run_ins(a, workload, rank, world, device, dist, parallel, steps, warmup, batch)
    if workload in ("steps", "steps_voc"):
        return run_steps(a, rank, world, device, dist, parallel, steps, warmup, batch, voc_sizes=(workload == "steps_voc"))
    return run_backbone(a, workload, rank, world, device, dist, parallel, steps, warmup, batch)


LEG_RUNS = {   # short runs for the `legs` object of the default line: (steps, warmup, batch)
    # (two warm-up steps for the backbone legs: the caching allocator is emptied between legs and still grows in the second step)
    "cam": (12, 2, 8), "e2e": (12, 2, 8), "cam_fp32": (12, 2, 8), "e2e_fp32": (12, 2, 8), "steps": (2, 1, 256), "walk_r5": (10, 2, 256), "walk_plain": (4, 1, 192),
    "ins": (10, 2, 128), "ins_r10": (8, 2, 128), "coco": (10, 2, 2),
    "walk_voc": (10, 2, 192), "walk_voc_r5": (10, 2, 256), "steps_voc": (1, 1, 256),
}


def measure_traffic(a, workload, batch):
    """HBM bytes of ONE launch of the dominant kernel, measured now: two rocprofv3 passes of this very script (`--pmc FETCH_SIZE`,
    then `--pmc WRITE_SIZE` — separate passes, counters only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) with one
    step and no warm-up, i.e. exactly one launch of `resident_kernel`; FETCH_SIZE doubled (gfx950 reports half the bytes of wide
    reads).  -> (bytes, detail) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="irn_pmc_")
    kb = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "-d", out, "-o", "t", "-f", "csv", "--", sys.executable, os.path.abspath(__file__), "--workload", workload,
                   "--batch", str(batch), "--steps", "1", "--warmup", "0", "--no-legs", "--no-cpu-baseline", "--no-traffic", "--variant", str(a.variant),
                   "--accel", str(a.accel)]
            env = dict(os.environ, TMPDIR="/tmp")
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=90)                          # (a pass takes ~15 s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)            # exactly the process group started above (profiler + its python child)
                proc.wait()
                return None, "rocprofv3 --pmc %s did not finish within 90 s" % counter
            res = proc
            # (the exit status is not the criterion: on this image the profiled python process can die in an exit handler AFTER
            # rocprofv3 has written its tables — what counts is whether the kernel's counter rows are there)
            tot, launches = 0.0, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "resident_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        tot += float(row.get("Counter_Value", 0) or 0)
                        launches.add(row.get("Dispatch_Id"))
            if not launches:
                return None, "no resident_kernel dispatch in the %s pass (rocprofv3 rc %d)" % (counter, res.returncode)
            kb[counter] = tot / len(launches)
        hbm = (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0
        return hbm, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of one launch each; FETCH_SIZE %.0f KB raw (doubled), WRITE_SIZE %.0f KB" % (
            kb["FETCH_SIZE"], kb["WRITE_SIZE"])
    except Exception as e:          # noqa: BLE001 — never lose the bench line to the counters
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_object(a, workload, r, live_traffic=None):
    if workload not in WALK_WORKLOADS:
        return None
    sec = r["avg_launch_ms"] * 1e-3
    tflops = r["flops_per_launch"] / sec / 1e12
    gbs = r["bytes_per_launch"] / sec / 1e9
    # SURVEY.md 8(d)'s F and B are quoted for 2^exp_times applications of the operator; the schedule applies it n_applied times
    power_scale = float(r["n_sweeps"]) / max(r["n_applied"], 1) if a.variant == 2 else 1.0
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch") == r["batch"] and tj.get("variant", 1) == a.variant and tj.get("n_applied", r["n_applied"]) == r["n_applied"]:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = "profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s); not re-measured in this run" % (
                    workload, tj.get("session", "see profiles/README.md"))
        except Exception:
            traffic = None
    source = None if traffic is None else "static"
    if live_traffic is not None:
        if live_traffic[0] is not None:
            traffic, traffic_src, source = live_traffic[0], live_traffic[1], "measured"
        elif traffic_src:
            traffic_src += "; live measurement unavailable: %s" % live_traffic[1]
    resident = a.variant == 2
    kernel = ("resident_kernel<%d> (weights-stationary persistent walk: one launch = the whole walk of the batch)" % r["radius"]) if resident else \
             ("sweep_blocked_kernel<%d,CH> (one operator application over the batch = 1 launch per channel-chunk width)" % r["radius"])
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": r["bytes_per_launch"],
           "note": "SURVEY.md 8(d) bytes of a STREAMING kernel (weights re-read every application: 4*N*(|S|+1+2C') per image and application) / launch time"}
    fma = {"achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
           "flops_per_launch": r["flops_per_launch"]}
    if resident:
        # the weights live in registers for the whole walk: HBM is not what bounds this kernel (traffic << algorithmic
        # streaming bytes); its ceiling is the fp32 vector FMA rate — and, for 1-2 channel images, the tile-to-tile
        # exchange latency
        top = dict(fma)
        top["bound"] = "fp32_vector"
        top["hbm_equivalent"] = hbm
        hbm["note"] += ("; this kernel reads the weights ONCE per image, so the figure may exceed the HBM peak — it measures speed-up "
                        "over any weight-streaming kernel, not HBM utilisation (see `traffic`)")
        top["power_equivalent"] = {
            "achieved": tflops * power_scale, "unit": "TFLOP/s", "frac_of_peak": tflops * power_scale / FP32_VECTOR_PEAK_TFLOPS,
            "flops": r["flops_per_launch"] * power_scale,
            "note": "SURVEY.md 8(d) F = 2*(2|S|+1)*N*C'*2^exp_times (the reference operator applied 2^exp_times times) over the same "
                    "launch time; the kernel executes n_applied/n_sweeps of that (`schedule`), which is what `achieved`/`frac` count"}
    else:
        top = dict(hbm)
        top["bound"] = "hbm"
        top["fp32_vector"] = fma
    top.update({"traffic": traffic, "traffic_source": source, "traffic_detail": traffic_src, "kernel": kernel, "avg_launch_ms": r["avg_launch_ms"],
                "sweeps_per_launch": r["sweeps_per_launch"], "launches_timed": r["launches_timed"],
                "sweep_share_of_step": r["sweep_share_of_step"],
                "schedule": {"n_sweeps": r["n_sweeps"], "operator_applications": r["n_applied"],
                             "kind": "plain powers" if r["n_applied"] == r["n_sweeps"] else
                                     "truncated Chebyshev series of lambda^n (dropped coefficients sum to < 1e-7), three-term recurrence"}})
    return top


def self_launch(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly the way the
    driver does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same arguments>`), pass rank 0's JSON line through and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=a.launch_timeout_s if a.launch_timeout_s > 0 else None)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)              # exactly the process group started above
        proc.wait()
        print("bench.py: the %d-rank run did not finish within %.0f s and was killed" % (a.gpus, a.launch_timeout_s), file=sys.stderr)
        return 124
    except KeyboardInterrupt:
        import signal
        os.killpg(proc.pid, signal.SIGTERM)
        raise


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a, argv))            # N ranks, one per GPU; this process only waits for them
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # the line's n_gpus is the world that ran; a launcher and a --gpus that disagree would mislabel it
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    from irn_amd import parallel
    ordinal = parallel.rank_device_ordinal(local_rank, a.rank_devices)
    if ordinal >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d, %d visible (--rank-devices maps ranks to devices, e.g. 0,0)" %
                         (rank, ordinal, torch.cuda.device_count()))
    torch.cuda.set_device(ordinal)
    device = torch.device("cuda", ordinal)
    if world > 1:          # one line per rank: which device, which shipped data it found (an 8-GPU node met for the first time)
        from irn_amd.step import _common
        _common.say(_common.startup_line(rank, world, ordinal, os.environ.get("MIOPEN_USER_DB_PATH") or "(claimed at the first backbone pass)"))
    # nccl == RCCL on ROCm.  The data path has no collective — the group only serves the contract's barrier and
    # max-over-ranks — so an RCCL start-up that fails or hangs must not cost the line: `auto` probes it under a deadline,
    # the ranks agree over a gloo control group, and the line says which backend served it
    dist, backend_used = parallel.init_process_group_with_fallback(a.backend, device, rank_devices=a.rank_devices)

    r = run_workload(a, a.workload, rank, world, device, dist, parallel, a.steps, a.warmup, a.batch)
    if rank == 0:
        stage = {"walk": "random-walk label generation stage", "walk_r5": "random-walk label generation stage, radius 5",
                 "walk_plain": "random-walk label generation stage, plain 2^exp_times iteration",
                 "coco": "random-walk label generation stage, COCO shape", "ins": "instance label generation stage",
                 "ins_r10": "instance label generation stage, radius 10",
                 "cam": "multi-scale CAM inference stage", "e2e": "CAM + IRNet + walk + labels, end to end",
                 "cam_fp32": "multi-scale CAM inference stage, fp32 GEMMs only", "e2e_fp32": "CAM + IRNet + walk + labels, end to end, fp32 GEMMs only",
                 "walk_voc": "random-walk label generation stage, VOC12 image-size histogram (ragged grids)",
                 "walk_voc_r5": "random-walk label generation stage, VOC12 image-size histogram, radius 5",
                 "steps_voc": "run_sample.py step API on JPEGs with the VOC12 image-size histogram",
                 "steps": "run_sample.py step API, make_cam + make_ins_seg_labels + make_sem_seg_labels"}[a.workload]
        res = {
            "metric": "%s (%s)" % (METRIC, stage),
            "value": r["value"], "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": describe(a.workload, r), "images_per_gpu_per_step": r["batch"],
                       "sharding": "images strided over ranks, no collective",
                       "process_group": {"backend": backend_used, "ranks": world,
                                         "devices": (a.rank_devices or "one per rank (LOCAL_RANK)"),
                                         "note": getattr(dist, "note", None),
                                         "used_for": "barrier + max-over-ranks of the timed region only"}},
            "roofline": None,
        }
        live = None
        if world == 1 and a.workload == "walk" and a.variant == 2 and not a.no_traffic and not a.no_legs:
            torch.cuda.synchronize()
            live = measure_traffic(a, a.workload, r["batch"])      # (a child process on the same GPU; this one is idle meanwhile)
        res["roofline"] = roofline_object(a, a.workload, r, live)
        if "shapes" in r:
            res["config"].update({"variant": a.variant, "mean_channels": float(np.mean([s[2] for s in r["shapes"]])),
                                  "walk_self_checks": r.get("tuning")})
            res["label_checksum"] = r["label_checksum"]
        for k in ("detections_per_image", "fallback_runs", "rounds", "walk_fallback_runs", "trunk", "cam_trunk_passes", "cam_store_hits", "cam_store_misses", "edge_store_hits", "loader_workers", "pass_seconds", "instance_files", "through", "gflop_per_image", "matrix_fp32_frac"):
            if k in r:
                res["config"][k] = r[k]
        res["cpu_baseline"] = None
        if world == 1 and not a.no_cpu_baseline and a.workload in ("walk", "walk_r5", "walk_plain", "coco"):      # (square-grid workloads)
            res["cpu_baseline"] = cpu_baseline(a, a.workload, a.cpu_images, 1000, gpu_labels=r.get("head_labels"))
            if "label_parity" in res["cpu_baseline"]:
                res["label_parity"] = res["cpu_baseline"]["label_parity"]
        if world == 1 and not a.no_legs and a.workload == "walk":
            legs = {}
            legs_t0 = time.perf_counter()
            for name in [n for n in a.legs.split(",") if n]:
                st, wu, b = LEG_RUNS[name]
                t0 = time.perf_counter()
                if t0 - legs_t0 > a.legs_budget_s:
                    legs[name] = {"skipped": "legs budget of %.0f s used up" % a.legs_budget_s}
                    continue
                try:
                    lr = run_workload(a, name, rank, world, device, None, parallel, st, wu, b)
                    legs[name] = {"value": lr["value"], "unit": "images/s", "steps": st, "warmup": wu, "batch": lr["batch"],
                                  "ms_per_step": lr["ms_per_step"]}
                    if "shapes" in lr:
                        ro = roofline_object(a, name, lr)
                        legs[name]["fp32_vector_frac"] = ro["frac"] if ro.get("bound") == "fp32_vector" else ro["fp32_vector"]["frac"]
                    for k in ("detections_per_image", "fallback_runs", "rounds", "grid_pixels", "walk_fallback_runs", "trunk", "cam_trunk_passes", "cam_store_hits", "cam_store_misses", "edge_store_hits", "loader_workers", "pass_seconds", "instance_files", "setup_seconds", "through", "n_applied", "gflop_per_image", "matrix_fp32_frac"):
                        if k in lr:
                            legs[name][k] = lr[k]
                except Exception as e:                      # a leg must never cost the headline line
                    legs[name] = {"error": repr(e)[:300]}
                legs[name]["wall_s"] = time.perf_counter() - t0
                torch.cuda.empty_cache()
            res["legs"] = legs
        line = json.dumps(res)
        print(line, flush=True)
        if a.json_out:
            with open(a.json_out, "w") as f:
                f.write(line + "\n")
    if dist:
        stuck = getattr(dist, "stuck", False)
        dist.close()
        if stuck:                      # an RCCL probe thread that never returned would block interpreter shutdown
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
