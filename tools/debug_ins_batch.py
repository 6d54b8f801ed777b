#!/usr/bin/env python3
"""Does an image's instance result depend on what else is in its batch?  Stage-by-stage comparison of
irn_amd.step.make_ins_seg_labels.instance_labels_batch's pipeline for image X alone, as [X, Y] and as [Y, X]
(ragged sizes like tests/test_gpu_steps.py: 96x128 and 113x150 images, i.e. 24x32 and 29x38 grids)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from irn_amd import ops, synth
from irn_amd.misc import indexing

dev = torch.device("cuda", 0)


def item(seed, size):
    H, W = size
    h, w = (H - 1) // 4 + 1, (W - 1) // 4 + 1
    k = 1 + seed % 3
    return {"edge": torch.from_numpy(synth.edge_field(h, w, seed))[None].to(dev), "dp": torch.from_numpy(synth.displacement_field(h, w, seed=seed, strength=0.3)).to(dev),
            "cam": torch.from_numpy(synth.cam_blobs(k, h, w, seed)).to(dev), "keys": torch.arange(k), "size": size}


def stages(walker, items):
    dps = [it["dp"] for it in items]
    cens = ops.find_centroids_batch(dps)
    cmaps, ks = ops.cluster_centroids_batch(cens, dps)
    rws = walker([it["edge"] for it in items], [it["cam"] for it in items], beta=10.0, exp_times=8, inst_maps=cmaps, k_inst=ks)
    fell = walker.sync()
    ep = ops.label_epilogue(rws, [it["size"] for it in items], 0.25, want_labels=False, want_argmax=True, want_rw_up=True)
    n_ch = [it["cam"].shape[0] * k for it, k in zip(items, ks)]
    cids = [np.repeat(it["keys"].numpy(), k) for it, k in zip(items, ks)]
    dets = ops.detect_instance_batch(ep["rw_up"], ep["argmax"], cids, n_ch, [it["size"][0] * it["size"][1] * 0.01 for it in items])
    out = []
    for i in range(len(items)):
        d = dets[i]
        out.append({"cen": cens[i].cpu().numpy(), "cmap": cmaps[i].cpu().numpy(), "k": ks[i], "rw": rws[i].cpu().numpy(), "argmax": ep["argmax"][i].cpu().numpy(),
                    "rw_up": ep["rw_up"][i].cpu().numpy(), "det": None if isinstance(d, Exception) else d, "fell": fell})
    return out


def diff(a, b, tag):
    msgs = []
    for key in ("cen", "cmap", "k", "rw", "argmax", "rw_up"):
        same = np.array_equal(a[key], b[key])
        if not same:
            extra = ""
            if key in ("rw", "rw_up") and a[key].shape == b[key].shape:
                extra = " max |d| %.3g" % float(np.abs(a[key] - b[key]).max())
            elif key in ("cmap", "argmax", "cen") and np.shape(a[key]) == np.shape(b[key]):
                extra = " %d px" % int((np.asarray(a[key]) != np.asarray(b[key])).sum())
            msgs.append(key + extra)
    if (a["det"] is None) != (b["det"] is None):
        msgs.append("det presence")
    elif a["det"] is not None:
        if a["det"]["mask"].shape != b["det"]["mask"].shape or not np.array_equal(a["det"]["mask"], b["det"]["mask"]):
            msgs.append("det masks %s vs %s" % (a["det"]["mask"].shape, b["det"]["mask"].shape))
        elif not np.array_equal(a["det"]["score"], b["det"]["score"]):
            msgs.append("det scores")
    print("%-44s %s  (fell back: %s / %s)" % (tag, "IDENTICAL" if not msgs else "DIFFERS: " + ", ".join(msgs), a["fell"], b["fell"]))


for radius in (10, 5):
    walker = indexing.RandomWalk(radius, dev)
    X, Y, Z = item(3, (96, 128)), item(4, (113, 150)), item(5, (96, 128))
    alone = {n: stages(walker, [it])[0] for n, it in (("X", X), ("Y", Y), ("Z", Z))}
    for combo in ("XY", "YX", "XZ", "ZX", "XYZ"):
        its = [{"X": X, "Y": Y, "Z": Z}[c] for c in combo]
        res = stages(walker, its)
        for c, r in zip(combo, res):
            diff(alone[c], r, "radius %d: %s alone vs inside [%s]" % (radius, c, ",".join(combo)))
    again = stages(walker, [X])[0]
    diff(alone["X"], again, "radius %d: X alone, repeated" % radius)
    walker.close()
