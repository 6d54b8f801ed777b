#!/usr/bin/env python3
"""Fill MIOpen's find database for the shape set of the label-generation steps and ship it with the package.

    python tools/miopen_warmup.py [--out irn_amd/data/miopen] [--batch 8] [--sizes 512x512,375x500] [--find-mode 1]
    python tools/miopen_warmup.py --channels-last 1 --single 0 --from-list voc12/train_aug.txt --voc12_root /data/VOC2012 --coverage 0.98
        (the image sizes of YOUR dataset, read from the JPEG headers, most frequent first until `--coverage` of the images is
        covered; sizes the shipped database already knows are skipped; ~40 s of GPU per size; then tools/miopen_det_filter.py)

Runs the CAM network at the four scales (image + flip pairs, `--batch` pairs per trunk pass like make_cam) and the IRNet
forward (`--batch` padded images per pass like the label steps) with MIOpen's FIND api enabled (PyTorch's
`cudnn.benchmark`, MIOPEN_FIND_MODE = `--find-mode`: 1 = normal find, every applicable solver is timed) into a fresh
user database, then copies that database to `<out>/<device name>-hip<version>/`.  `irn_amd/step/_common.py:miopen_setup`
seeds every worker's database from there, so a fresh process finds measured solvers instead of the immediate-mode
heuristic without paying the find itself (MIOPEN_FIND_MODE=2 in the steps: database first).  The backbones stay on
PyTorch-ROCm / MIOpen (reference net/resnet50_cam.py:55-70); this only persists what MIOpen measured."""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sizes_of_list(a):
    """The (H, W) histogram of the images named in `--from-list` (sizes from the JPEG headers, nothing is decoded), most frequent
    first, minus the sizes the shipped channels-last database already covers, until `--coverage` of the images is covered."""
    import collections
    import json
    from PIL import Image
    from irn_amd.voc12 import dataloader
    names = [l.strip().split()[0] for l in open(a.from_list) if l.strip()]
    names = [os.path.splitext(os.path.basename(n))[0] for n in names]
    hist = collections.Counter()
    for n in names:
        with Image.open(dataloader.get_img_path(n, a.voc12_root)) as im:
            w, h = im.size
        hist[(h, w)] += 1
    tuned = set()
    for d in sorted(os.listdir(a.out)) if os.path.isdir(a.out) else []:
        path = os.path.join(a.out, d, "nhwc_shapes.json")
        if os.path.exists(path) and not d.endswith("-det"):
            tuned |= {(int(v[1]), int(v[2])) for v in json.load(open(path))}
    total, covered, todo = sum(hist.values()), 0, []
    for (h, w), c in hist.most_common():
        if covered >= a.coverage * total or len(todo) >= a.max_sizes:
            break
        covered += c
        if (h, w) not in tuned:
            todo.append("%dx%d" % (h, w))
    print("%d images, %d distinct sizes; %d sizes cover %.1f %% of them, %d of those are new: %s" % (
        total, len(hist), len(todo) + sum(1 for s in hist if s in tuned), 100.0 * covered / max(total, 1), len(todo), ",".join(todo)), flush=True)
    if not todo:
        raise SystemExit("nothing to tune")
    return ",".join(todo)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "irn_amd", "data", "miopen"))
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sizes", default="512x512")
    ap.add_argument("--find-mode", default="1")
    ap.add_argument("--channels-last", type=int, default=0)
    ap.add_argument("--single", type=int, default=1, help="also the one-pair / one-image shapes (the tail of a shard)")
    ap.add_argument("--suffix", default="", help="appended to the output directory's key (e.g. -nhwc)")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="1: tune under MIOpen's deterministic attribute (round 5 measured: no fast NHWC fp32 solver survives it).  The "
                         "reproducible mode's database is DERIVED from the fast one instead: run tools/miopen_det_filter.py afterwards")
    ap.add_argument("--fused-gemm", type=int, default=1, help="the trunk's stride-1 1x1 convolutions are hipBLASLt GEMMs (not MIOpen problems)")
    ap.add_argument("--from-list", default=None, help="an image list of the reference's format (voc12/train_aug.txt): tune the sizes that occur in it")
    ap.add_argument("--voc12_root", default=None, help="dataset root of --from-list (JPEGImages/<name>.jpg)")
    ap.add_argument("--coverage", type=float, default=0.98, help="--from-list: stop when this fraction of the images has a tuned size")
    ap.add_argument("--max-sizes", type=int, default=40, help="--from-list: at most this many new sizes")
    a = ap.parse_args()
    if a.from_list:
        a.sizes = sizes_of_list(a)
    db = tempfile.mkdtemp(prefix="irn_miopen_warm_")
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ["MIOPEN_FIND_MODE"] = a.find_mode
    os.environ["IRN_MIOPEN_DB_SET"] = db                      # keep miopen_setup out of the way
    os.environ["IRN_CHANNELS_LAST"] = "1" if a.channels_last else "0"
    os.environ["IRN_FUSED_GEMM"] = "1" if a.fused_gemm else "0"
    os.environ["IRN_DETERMINISTIC"] = "1" if a.deterministic else "0"
    import torch
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    from irn_amd.step import _common
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.deterministic = bool(a.deterministic)
    dev = torch.device("cuda", 0)
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(1))
    cam = cam.to(dev).eval()
    irn = resnet50_irn.EdgeDisplacement()
    irn.load_state_dict(weights.random_irn_state(2), strict=False)
    irn = irn.to(dev).eval()
    t0 = time.time()
    shapes = set()
    with torch.no_grad():
        for size in a.sizes.split(","):
            h, w = (int(v) for v in size.split("x"))
            for s in (1.0, 0.5, 1.5, 2.0):
                hs, ws = int(round(h * s)), int(round(w * s))
                for b in sorted({a.batch} | ({1} if a.single else set())):
                    x = torch.randn(2 * b, 3, hs, ws, device=dev)
                    shapes.add((2 * b, hs, ws))
                    t1 = time.time()
                    cam.forward_batch(x)
                    torch.cuda.synchronize()
                    print("cam   %4dx%-4d pairs %d: %.1f s" % (hs, ws, b, time.time() - t1), flush=True)
            for b in sorted({a.batch} | ({1} if a.single else set())):
                imgs = [torch.randn(2, 3, h, w, device=dev) for _ in range(b)]
                shapes.add((2 * b, irn.crop_size, irn.crop_size))
                t1 = time.time()
                irn.forward_batch(imgs)
                torch.cuda.synchronize()
                print("irnet %4dx%-4d images %d: %.1f s" % (h, w, b, time.time() - t1), flush=True)
    key = _common.miopen_cache_key() + a.suffix
    dst = os.path.join(a.out, key)
    os.makedirs(dst, exist_ok=True)
    n = _common.merge_miopen_db(db, dst)                      # adds to what an earlier warm-up (the other layout) left there
    for f in sorted(os.listdir(dst)):
        print("  %s  %d bytes" % (f, os.path.getsize(os.path.join(dst, f))))
    if a.channels_last:
        import json
        path = os.path.join(dst, "nhwc_shapes.json")
        old = {tuple(v) for v in json.load(open(path))} if os.path.exists(path) else set()
        json.dump(sorted(old | shapes), open(path, "w"))
        print("  nhwc_shapes.json: %s" % sorted(old | shapes))
    print("find database of %s: %d file(s) -> %s (%.0f s)" % (key, n, dst, time.time() - t0))
    shutil.rmtree(db, ignore_errors=True)


if __name__ == "__main__":
    main()
