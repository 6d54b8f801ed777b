#!/usr/bin/env python3
"""Where the time of the batched instance-label step goes (configs[3] shape: 512^2 images, 128x128 grids): every stage
of irn_amd.step.make_ins_seg_labels.instance_labels_batch timed with a device synchronisation behind it.
usage: ins_step_breakdown.py [radius=5] [batch=64] [reps=5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from irn_amd import ops, synth
from irn_amd.misc import indexing

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
h = w = 128
items = []
for i in range(N):
    k = synth.voc_num_classes(1000 + i)
    items.append({"edge": torch.from_numpy(synth.edge_field(h, w, 1000 + i))[None].to(dev),
                  "dp": torch.from_numpy(synth.displacement_field(h, w, seed=1000 + i, strength=0.3)).to(dev),
                  "cam": torch.from_numpy(synth.cam_blobs(k, h, w, 1000 + i)).to(dev),
                  "keys": torch.from_numpy(synth.voc_keys(k, 1000 + i)), "size": (512, 512)})
walker = indexing.RandomWalk(radius, dev)
acc = {}
det_t = {}


def tick(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


for rep in range(reps + 1):
    if rep == 1:
        acc.clear()
        det_t.clear()
    t = time.perf_counter()
    dps = [it["dp"] for it in items]
    cens = ops.find_centroids_batch(dps)
    t = tick("centroids", t)
    cmaps, ks = ops.cluster_centroids_batch(cens, dps)
    t = tick("cluster (+K to host)", t)
    rws = walker([it["edge"] for it in items], [it["cam"] for it in items], beta=10.0, exp_times=8, inst_maps=cmaps, k_inst=ks)
    walker.sync()
    t = tick("walk", t)
    ep = ops.label_epilogue(rws, [it["size"] for it in items], 0.25, want_labels=False, want_argmax=True, want_rw_up=True)
    t = tick("epilogue (argmax + rw_up)", t)
    n_ch = [it["cam"].shape[0] * k for it, k in zip(items, ks)]
    cids = [np.repeat(it["keys"].numpy(), k) for it, k in zip(items, ks)]
    dets = ops.detect_instance_batch(ep["rw_up"], ep["argmax"], cids, n_ch, [512 * 512 * 0.01] * N, timings=det_t)
    t = tick("detect (count, emit, D2H, unpack)", t)
tot = sum(acc.values())
nd = sum(0 if isinstance(d, Exception) else len(d["score"]) for d in dets)
print("radius %d, batch %d: %.1f images/s; %.1f channels and %.1f detections per image" %
      (radius, N, reps * N / tot, float(np.mean(n_ch)), nd / float(N)))
for k, v in acc.items():
    print("  %-36s %7.3f ms per image  (%4.1f %%)" % (k, 1e3 * v / (reps * N), 100 * v / tot))
print("  detect split: labelling + counts %.3f, emit + transfer %.3f (%.1f MB per image), host unpack %.3f ms per image" %
      (1e3 * det_t["count"] / (reps * N), 1e3 * det_t["emit_d2h"] / (reps * N), det_t["bytes"] / (reps * N) / 1e6,
       1e3 * det_t["unpack"] / (reps * N)))
