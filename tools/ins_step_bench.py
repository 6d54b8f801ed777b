#!/usr/bin/env python3
"""Instance-label step on resident tensors (configs[3] shape: 512^2 images, radius 5 as the reference's
call site hard-codes, beta 10, 2^8 sweeps): front-end + walk + epilogue + detection per image, walking
one image at a time (the reference's loop) vs `walk_batch` images per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from irn_amd import synth
from irn_amd.misc import indexing
from irn_amd.step import make_ins_seg_labels as mis
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h = w = 128
items = []
for i in range(N):
    k = synth.voc_num_classes(i)
    items.append({"edge": torch.from_numpy(synth.edge_field(h, w, i))[None].to(dev),
                  "dp": torch.from_numpy(synth.displacement_field(h, w, seed=i, strength=0.3)).to(dev),
                  "cam": torch.from_numpy(synth.cam_blobs(k, h, w, i)).to(dev),
                  "keys": torch.from_numpy(synth.voc_keys(k, i)), "size": (512, 512)})
walker = indexing.RandomWalk(5, dev)
def run(batch):
    out = []
    for s in range(0, N, batch):
        out += mis.instance_labels_batch(walker, items[s:s + batch], 10.0, 8, 0.25)
    return out
ref = run(1)
for batch in (1, 8, 32, 64):
    run(batch); torch.cuda.synchronize(); t0 = time.time(); res = run(batch); torch.cuda.synchronize(); dt = time.time() - t0
    same = all((isinstance(a, Exception) and isinstance(b, Exception)) or
               (not isinstance(a, Exception) and not isinstance(b, Exception) and np.array_equal(a["mask"], b["mask"]) and np.array_equal(a["score"], b["score"]) and np.array_equal(a["class"], b["class"]))
               for a, b in zip(ref, res))
    ndet = sum(0 if isinstance(r, Exception) else len(r["score"]) for r in res)
    print("walk_batch %2d: %.1f images/s (%.2f ms per image), %d detections, identical to batch 1: %s" % (batch, N / dt, 1e3 * dt / N, ndet, same), flush=True)
