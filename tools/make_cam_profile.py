"""Where does make_cam.run(args) spend its host time beyond the backbone?  cProfile of the main thread over a synthetic
VOC directory (the `steps` leg's make_cam pass is ~1 ms per image slower than the `cam` leg's resident loop).
usage: python tools/make_cam_profile.py <outdir> [n_images=128]"""
import argparse
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irn_amd import synth                                   # noqa: E402
from irn_amd.net import weights                             # noqa: E402
from irn_amd.step import _common, make_cam                  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tmp = tempfile.mkdtemp(prefix="irn_mcp_")
try:
    root = os.path.join(tmp, "voc")
    os.makedirs(os.path.join(root, "JPEGImages"))
    names, labels = [], {}
    for i in range(n):
        name = "2009_%06d" % (i + 1)
        Image.fromarray(synth.photo(512, 512, seed=7000 + i)).save(os.path.join(root, "JPEGImages", name + ".jpg"), quality=92)
        lab = np.zeros(20, np.float32)
        lab[synth.voc_keys(synth.voc_num_classes(i + 11), i + 11)] = 1
        names.append(name)
        labels[int(name.replace("_", ""))] = lab
    open(os.path.join(tmp, "train.txt"), "w").write("\n".join(names) + "\n")
    np.save(os.path.join(tmp, "cls_labels.npy"), labels)
    torch.save(weights.random_cam_state(1), os.path.join(tmp, "res50_cam.pth"))
    args = argparse.Namespace(num_workers=4, voc12_root=root, train_list=os.path.join(tmp, "train.txt"),
                              cam_network="net.resnet50_cam", cam_weights_name=os.path.join(tmp, "res50_cam"),
                              cam_scales=(1.0, 0.5, 1.5, 2.0), cam_out_dir=os.path.join(tmp, "cam"), worker_devices="0")
    make_cam.run(args)                                          # warm-up: MIOpen, caches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prof = cProfile.Profile()
    prof.enable()
    make_cam.run(args)
    torch.cuda.synchronize()
    prof.disable()
    dt = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(32)
    print("make_cam.run: %.3f s for %d images = %.2f ms per image" % (dt, n, 1e3 * dt / n))
    print(s.getvalue()[:6000])
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(18)
    print(s.getvalue()[:4000])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
