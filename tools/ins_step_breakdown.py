import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from irn_amd import ops, synth
from irn_amd.misc import indexing
dev = torch.device("cuda", 0)
N = 32; h = w = 128
items = []
for i in range(N):
    k = synth.voc_num_classes(i)
    items.append({"edge": torch.from_numpy(synth.edge_field(h, w, i))[None].to(dev),
                  "dp": torch.from_numpy(synth.displacement_field(h, w, seed=i, strength=0.3)).to(dev),
                  "cam": torch.from_numpy(synth.cam_blobs(k, h, w, i)).to(dev), "keys": torch.from_numpy(synth.voc_keys(k, i)), "size": (512, 512)})
walker = indexing.RandomWalk(5, dev)
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3 / N, r
t, cens = T(lambda: [ops.find_centroids_with_refinement(it["dp"]) for it in items]); print("centroids   %.3f ms/img" % t)
t, cl = T(lambda: [ops.cluster_centroids(c, it["dp"]) for c, it in zip(cens, items)]); print("cluster     %.3f ms/img" % t)
cmaps = [c[0] for c in cl]; ks = [c[1] for c in cl]; print("instances per image:", ks)
t, rws = T(lambda: walker([it["edge"] for it in items], [it["cam"] for it in items], beta=10.0, exp_times=8, inst_maps=cmaps, k_inst=ks)); print("walk        %.3f ms/img (mean channels %.1f)" % (t, np.mean([it["cam"].shape[0] * k for it, k in zip(items, ks)])))
t, ep = T(lambda: ops.label_epilogue(rws, [it["size"] for it in items], 0.25, want_labels=False, want_argmax=True, want_rw_up=True)); print("epilogue    %.3f ms/img" % t)
def det():
    out = []
    for i, it in enumerate(items):
        try: out.append(ops.detect_instance(ep["rw_up"][i], ep["argmax"][i], np.repeat(np.asarray(it["keys"]), ks[i]), it["cam"].shape[0] * ks[i], 2621.44))
        except ValueError as e: out.append(e)
    return out
t, d = T(det); print("detect      %.3f ms/img (incl. D2H of the masks)" % t)
