"""Semantic pseudo-label step — drop-in for reference step/make_sem_seg_labels.py (`run(args)`).

Reads  args.irn_network, args.irn_weights_name, args.infer_list, args.voc12_root, args.cam_out_dir,
       args.beta, args.exp_times, args.sem_seg_bg_thres, args.num_workers
Writes args.sem_seg_out_dir/<name>.png  uint8 [H,W] (0 = background, class+1 otherwise)

Per image (step/make_sem_seg_labels.py:28-51): EdgeDisplacement forward (PyTorch-ROCm) -> edge;
CAM dict from disk; random walk of the CAMs over the edge affinities and the label epilogue run in
libirn_hip.so.  Unlike the reference's batch-1 loop the walk is issued for `walk_batch` images at a
time (default 64) so one launch fills the GPU for several rounds; results per image are unchanged.
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader

from .. import ops
from ..misc import indexing, torchutils
from ..voc12 import dataloader as voc12_dataloader
from . import _common

RADIUS = 5   # hard-coded at the reference call site (step/make_sem_seg_labels.py:41)


def _save_png(path, label):
    Image.fromarray(label).save(path)


def _flush(walker, pend, args, writer):
    if not pend:
        return
    rws = walker([p["edge"] for p in pend], [p["cam"] for p in pend],
                 beta=float(args.beta), exp_times=int(args.exp_times))
    out = ops.label_epilogue(rws, [p["size"] for p in pend], float(args.sem_seg_bg_thres),
                             keys=[p["keys"] for p in pend])
    labels = [lab.cpu().numpy() for lab in out["labels"]]
    walker.check()                          # the persistent walk reports a stuck tile instead of hanging
    for p, lab in zip(pend, labels):
        writer.submit(_save_png, os.path.join(args.sem_seg_out_dir, p["name"] + ".png"), lab)
    pend.clear()


def _work(process_id, model, dataset, args):
    databin = dataset[process_id]
    n_gpus = len(dataset)
    loader = DataLoader(databin, shuffle=False, num_workers=int(args.num_workers) // n_gpus, pin_memory=False)
    batch = int(getattr(args, "walk_batch", 0) or 64)   # 64 VOC-size images = 3-4 rounds of the resident walk
    with torch.no_grad(), torch.cuda.device(process_id):
        model.cuda()
        walker = indexing.RandomWalk(RADIUS)
        writer = _common.AsyncWriter()
        pend = []
        for it, pack in enumerate(loader):
            name = pack["name"][0]
            if not isinstance(name, str):
                name = voc12_dataloader.decode_int_filename(name)
            size = (int(pack["size"][0]), int(pack["size"][1]))
            edge, _dp = model(_common.device_images(pack, (1.0,))[0])
            cam_dict = np.load(os.path.join(args.cam_out_dir, name + ".npy"), allow_pickle=True).item()
            pend.append({"name": name, "size": size, "edge": edge,
                         "cam": torch.as_tensor(cam_dict["cam"]).cuda(),
                         "keys": torch.as_tensor(cam_dict["keys"]).cuda()})
            if len(pend) == batch:
                _flush(walker, pend, args, writer)
            _common.progress(process_id, n_gpus, it, len(databin))
        _flush(walker, pend, args, writer)
        writer.close()
        walker.close()


def run(args):
    model = getattr(_common.import_network(args.irn_network), "EdgeDisplacement")()
    model.load_state_dict(torch.load(args.irn_weights_name, map_location="cpu"), strict=False)
    model.eval()
    n_gpus = _common.n_gpus_or_raise()
    dataset = voc12_dataloader.VOC12ClassificationDatasetMSF(args.infer_list, voc12_root=args.voc12_root,
                                                             scales=(1.0,), raw=_common.device_preprocess(args))
    dataset = torchutils.split_dataset(dataset, n_gpus)
    os.makedirs(args.sem_seg_out_dir, exist_ok=True)
    print("[", end="")
    _common.spawn_workers(_work, model, dataset, args)
    print("]")
    torch.cuda.empty_cache()
