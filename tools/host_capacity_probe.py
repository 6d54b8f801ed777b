#!/usr/bin/env python3
"""What the HOST side of the three label-generation steps can sustain, without a GPU: P processes (one per GPU of a node),
each running the steps' own loader (`_common.make_loader` over `VOC12ClassificationDatasetMSF(raw=True)`: PIL JPEG decode in
threads) and writer (`_common.AsyncWriter`: np.save of real-size CAM / instance dictionaries, PNG label maps) at the same time,
with nothing in between — the rate at which a worker could be fed and drained if its GPU were infinitely fast.

    python tools/host_capacity_probe.py [--procs 1,8] [--images 192] [--size 512x512] [--loader-threads 8] [--writer-threads 4]

Per pass it prints images/s per process and in aggregate, next to what 8 GPUs ask for (8 x the `cam` / `steps` legs of the last
bench line, --demand-cam / --demand-steps).  Reference: the loader workers and per-image np.save / imageio.imsave of
step/make_cam.py:20-56, step/make_sem_seg_labels.py:24-51, step/make_ins_seg_labels.py:113-152."""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, a, q, go):
    import numpy as np
    import torch
    from PIL import Image
    from irn_amd import synth
    from irn_amd.step import _common
    from irn_amd.voc12 import dataloader
    torch.set_num_threads(1)
    h, w = (int(v) for v in a.size.split("x"))
    tmp = tempfile.mkdtemp(prefix="irn_hostprobe_%d_" % rank)
    try:
        os.makedirs(os.path.join(tmp, "voc", "JPEGImages"))
        names, labels = [], {}
        n_src = min(a.images, 32)                      # distinct JPEGs (decode cost does not depend on which)
        for i in range(a.images):
            name = "2010_%06d" % (rank * a.images + i + 1)
            src = os.path.join(tmp, "voc", "JPEGImages", "2010_%06d.jpg" % (rank * a.images + (i % n_src) + 1))
            dst = os.path.join(tmp, "voc", "JPEGImages", name + ".jpg")
            if i < n_src:
                Image.fromarray(synth.photo(h, w, seed=100 * rank + i)).save(dst, quality=92)
            else:
                os.link(src, dst)
            names.append(name)
            lab = np.zeros(20, np.float32)
            lab[synth.voc_keys(synth.voc_num_classes(i + 11), i + 11)] = 1
            labels[int(name.replace("_", ""))] = lab
        with open(os.path.join(tmp, "train.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        np.save(os.path.join(tmp, "cls_labels.npy"), labels)
        ds = dataloader.VOC12ClassificationDatasetMSF(os.path.join(tmp, "train.txt"), voc12_root=os.path.join(tmp, "voc"), raw=True)
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        gh, gw = (h - 1) // 4 + 1, (w - 1) // 4 + 1
        rng = np.random.RandomState(rank)
        # payloads of the real sizes (K classes per image from the VOC histogram; ~7.5 detections per image)
        cam_pay = {k: (torch.from_numpy(rng.rand(k, gh, gw).astype(np.float32)), rng.rand(k, h, w).astype(np.float32)) for k in (1, 2, 3, 4)}
        det_pay = {"score": rng.rand(8).astype(np.float32), "mask": rng.rand(8, h, w) > 0.7, "class": np.arange(8)}
        # a label map like the step writes: a few classes in blobs (noise would be incompressible: PNG encoding 3x slower)
        png_pay = np.kron(rng.choice([0, 0, 0, 5, 12, 15], size=((h + 63) // 64, (w + 63) // 64)), np.ones((64, 64), int))[:h, :w].astype(np.uint8)

        def save_png(path, lab):
            Image.fromarray(lab).save(path)

        def run_pass(kind):
            writer = _common.AsyncWriter(threads=a.writer_threads)
            t0 = time.perf_counter()
            n = 0
            for pack in _common.make_loader(ds, a.loader_threads):
                name = pack["name"][0]
                k = int(pack["label"][0].sum())
                staging = pack.pop("_staging", None)          # the loader staged the image in page-locked memory (as for a real step):
                if staging is not None:                       # there the upload recycles the buffer, here nothing uploads
                    _common.PINNED.give(staging)
                if kind == "make_cam":
                    cam, hi = cam_pay[min(max(k, 1), 4)]
                    writer.submit(np.save, os.path.join(out, name + ".npy"), {"keys": torch.arange(k), "cam": cam.clone(), "high_res": hi})
                elif kind == "make_sem_seg_labels":
                    writer.submit(save_png, os.path.join(out, name + ".png"), png_pay.copy())
                else:
                    writer.submit(np.save, os.path.join(out, name + "_ins.npy"), det_pay)
                n += 1
            writer.close()
            return n / (time.perf_counter() - t0)

        q.put((rank, "ready", None))
        go.wait()
        res = {}
        for kind in ("make_cam", "make_ins_seg_labels", "make_sem_seg_labels"):
            res[kind] = run_pass(kind)
            for f in os.listdir(out):
                os.remove(os.path.join(out, f))
        q.put((rank, "done", res))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="1,8")
    ap.add_argument("--images", type=int, default=192)
    ap.add_argument("--size", default="512x512")
    ap.add_argument("--loader-threads", type=int, default=8)
    ap.add_argument("--writer-threads", type=int, default=4)
    ap.add_argument("--demand-cam", type=float, default=117.0, help="images/s ONE GPU's make_cam asks for (bench leg `cam`)")
    ap.add_argument("--demand-steps", type=float, default=75.0, help="images/s one GPU's three passes sustain together (bench leg `steps`)")
    a = ap.parse_args()
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    print("host: %d cores; JPEG %s; %d images per process and pass; %d loader + %d writer threads per process (the steps' own caps)" %
          (os.cpu_count(), a.size, a.images, a.loader_threads, a.writer_threads))
    for procs in [int(v) for v in a.procs.split(",")]:
        q, go = ctx.Queue(), ctx.Event()
        ps = [ctx.Process(target=_worker, args=(r, a, q, go)) for r in range(procs)]
        for p in ps:
            p.start()
        for _ in ps:
            q.get()
        go.set()
        results = [q.get()[2] for _ in ps]
        for p in ps:
            p.join()
        print("-- %d process(es)" % procs)
        for kind, demand in (("make_cam", a.demand_cam), ("make_ins_seg_labels", None), ("make_sem_seg_labels", None)):
            per = [r[kind] for r in results]
            line = "  %-22s %7.1f images/s per process (min %.1f), %8.1f aggregate" % (kind, sum(per) / len(per), min(per), sum(per))
            if demand:
                line += "   | %d GPU(s) ask for %.0f" % (procs, procs * demand)
            print(line)
        # the three passes run one after the other per image set: harmonic combination = images/s of the whole job on the host side
        whole = [1.0 / sum(1.0 / r[k] for k in r) for r in results]
        print("  %-22s %7.1f images/s per process, %8.1f aggregate   | %d GPU(s) sustain %.0f through the three passes" %
              ("all three passes", sum(whole) / len(whole), sum(whole), procs, procs * a.demand_steps))


if __name__ == "__main__":
    main()
