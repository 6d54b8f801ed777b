"""Where do make_ins_seg_labels.run(args) and make_sem_seg_labels.run(args) spend their host time?  cProfile of the main
thread over a synthetic VOC directory, in run_sample.py's order (cam -> ins at radius 5 -> sem at radius 10), the `steps`
leg's settings.  usage: python tools/label_steps_profile.py [n_images=256] [loader_workers=4]"""
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
from irn_amd.step import _common, make_cam, make_ins_seg_labels, make_sem_seg_labels     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = tempfile.mkdtemp(prefix="irn_lsp_")
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
    torch.save(weights.random_irn_state(2), os.path.join(tmp, "res50_irn.pth"))
    args = argparse.Namespace(
        num_workers=lw, voc12_root=root, train_list=os.path.join(tmp, "train.txt"), infer_list=os.path.join(tmp, "train.txt"),
        cam_network="net.resnet50_cam", cam_weights_name=os.path.join(tmp, "res50_cam"), cam_scales=(1.0, 0.5, 1.5, 2.0),
        irn_network="net.resnet50_irn", irn_weights_name=os.path.join(tmp, "res50_irn.pth"), beta=10, exp_times=8, sem_seg_bg_thres=0.25,
        ins_seg_bg_thres=0.25, cam_out_dir=os.path.join(tmp, "cam"), sem_seg_out_dir=os.path.join(tmp, "sem"),
        ins_seg_out_dir=os.path.join(tmp, "ins"), radius=10, walk_batch=64, worker_devices="0")

    class Quiet:
        def write(self, s):
            return len(s)

        def flush(self):
            pass

    def one_pass(profile):
        real = sys.stdout
        sys.stdout = Quiet()
        res = {}
        try:
            _common.EDGE_STORE.clear()
            make_cam.run(args)
            for name, mod, radius, wb in (("make_ins_seg_labels", make_ins_seg_labels, 5, 32), ("make_sem_seg_labels", make_sem_seg_labels, 10, 64)):
                args.radius, args.walk_batch = radius, wb
                torch.cuda.synchronize()
                prof = cProfile.Profile() if profile else None
                t0 = time.perf_counter()
                if prof:
                    prof.enable()
                mod.run(args)
                torch.cuda.synchronize()
                if prof:
                    prof.disable()
                res[name] = (time.perf_counter() - t0, prof)
        finally:
            sys.stdout = real
        return res

    one_pass(False)                                             # warm-up: MIOpen, caches, poll-delay probe
    for name, (dt, prof) in one_pass(True).items():
        print("== %s: %.3f s for %d images = %.2f ms per image (%d loader threads)" % (name, dt, n, 1e3 * dt / n, lw))
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(14)
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3500])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
