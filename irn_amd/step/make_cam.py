"""Multi-scale CAM inference step — drop-in for reference step/make_cam.py (`run(args)`).

Reads  args.cam_network, args.cam_weights_name(+'.pth'), args.train_list, args.voc12_root,
       args.cam_scales, args.num_workers
Writes args.cam_out_dir/<name>.npy = {"keys": LongTensor[K], "cam": FloatTensor[K,h,w] (cpu),
       "high_res": float32 ndarray [K,H,W]}   (same pickle schema as step/make_cam.py:55-56)

The per-scale inputs are built on the GPU from the decoded uint8 image (irn_msf_pack: Pillow-exact bicubic,
normalise, flip pair; args.device_preprocess=False keeps the reference's PIL loop in the loader workers).
The ResNet-50 forward passes run on PyTorch-ROCm (MIOpen), `cam_batch` images of one size per pass (default 8;
the reference: one image + its flip); the merge (step/make_cam.py:38-52) is one HIP kernel pair (irn_cam_merge).
Every image's {keys, cam} also stays on the device (`_common.CAM_STORE`) for label steps that run later in the same
process.  One process per GPU over strided shards, no communication.
"""
import os
import warnings

import numpy as np
import torch

from ..misc import torchutils
from ..voc12 import dataloader as voc12_dataloader
from . import _common


def merge_scales(outputs, size, label):
    """outputs: per-scale GPU [20,hs,ws] activation maps -> (keys, cam [K,ceil(H/4),ceil(W/4)],
    high_res [K,H,W]) — step/make_cam.py:38-52 in two launches of libirn_hip.so (irn_cam_merge).
    GPU tensors only (the torch-op restatement the tests compare with lives in oracle/torch_mirrors.py)."""
    from .. import ops
    return ops.cam_merge(outputs, size, label)


def _save_cam(path, keys_cpu, event, staging, cam_view, hi_view):
    """Writer-thread half of a CAM hand-off: wait for the image's device-to-host copies, write the reference's
    dictionary (step/make_cam.py:55-56), recycle the staging buffer."""
    event.synchronize()
    try:
        np.save(path, {"keys": keys_cpu, "cam": cam_view.clone(), "high_res": hi_view.numpy()})
    finally:
        _common.PINNED.give(staging)


def _flush_group(model, group, scales, args, writer, store):
    """One trunk pass per scale for all images of a size group ([image, flip, image, flip, ...]), then the per-image
    merge (irn_cam_merge) and the asynchronous write of the reference's dictionary.  Nothing here waits for the device:
    the present classes come from the host-side label, the results leave through page-locked staging on the stream, and
    the writer threads wait for each image's copy event — so the next group's passes are queued while this one runs."""
    if not group:
        return
    from ..net import resnet50 as _r50
    outs = []
    for si in range(len(scales)):
        x = torch.cat([g["imgs"][si] for g in group])
        before = (_r50.PASS_STATS["channels_last"], _r50.PASS_STATS["nchw"])
        outs.append(model.forward_batch(x))
        _common.CAM_STATS["channels_last"] += _r50.PASS_STATS["channels_last"] - before[0]
        _common.CAM_STATS["nchw"] += _r50.PASS_STATS["nchw"] - before[1]
    _common.CAM_STATS["full_group_flushes" if len(group) >= int(getattr(args, "cam_batch", 0) or 8) else "partial_group_flushes"] += 1
    for i, g in enumerate(group):
        keys_cpu = torch.nonzero(g["label"])[:, 0]
        keys, cam, high_res = merge_scales([o[i] for o in outs], g["size"], g["label"])
        if store is not None:
            store.put(g["name"], keys_cpu, keys, cam, cam_out_dir=args.cam_out_dir, run_id=getattr(args, "cam_run_id", None))
        n_cam, n_hi = cam.numel() * 4, high_res.numel() * 4
        staging = _common.PINNED.take(n_cam + n_hi)
        cam_view = staging[:n_cam].view(torch.float32).view(cam.shape)
        hi_view = staging[n_cam:n_cam + n_hi].view(torch.float32).view(high_res.shape)
        cam_view.copy_(cam, non_blocking=True)
        hi_view.copy_(high_res, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        writer.submit(_save_cam, os.path.join(args.cam_out_dir, g["name"] + ".npy"), keys_cpu, event, staging, cam_view, hi_view)
    group.clear()


def _work(process_id, model, dataset, args):
    model = _common.materialise(model)      # a network, or the (class, checkpoint) a worker builds it from
    databin = dataset[process_id]
    n_gpus = len(dataset)
    loader = _common.make_loader(databin, int(args.num_workers) // n_gpus)
    writer = _common.AsyncWriter(threads=_common.writer_threads(args, n_gpus))
    scales = tuple(float(s) for s in args.cam_scales)
    # images per trunk pass (the reference runs batch 2 = one image + flip, step/make_cam.py:32-33); images of one size
    # are stacked per scale — VOC is mostly 500x375 / 375x500 — and every size group is flushed at the end
    batch = int(getattr(args, "cam_batch", 0) or 8)
    store = _common.CAM_STORE if _common.keep_cams(args) else None
    _common.CAM_STORE.drop_dir(args.cam_out_dir)        # this directory's files are about to be rewritten
    groups = {}
    pending_bytes, max_pending = 0, int(getattr(args, "cam_pending_bytes", 0) or (4 << 30))
    try:
        with torch.no_grad(), torch.cuda.device(_common.worker_device(process_id, args)):
            model.cuda()
            for it, pack in enumerate(loader):
                img_name = pack["name"][0]
                if not isinstance(img_name, str):
                    img_name = voc12_dataloader.decode_int_filename(img_name)
                label = pack["label"][0]
                size = (int(pack["size"][0]), int(pack["size"][1]))
                if float(label.sum()) == 0:
                    # the reference would write an empty dictionary here and crash later in the label steps
                    warnings.warn("%s: no positive class in the image-level label, skipped" % img_name)
                    staging = pack.pop("_staging", None) if isinstance(pack, dict) else None
                    if staging is not None:                  # the loader thread's page-locked copy of an image nobody uploads
                        _common.PINNED.give(staging)
                    continue
                group = groups.setdefault(size, [])
                imgs = _common.device_images(pack, scales)
                group.append({"name": img_name, "size": size, "label": label, "imgs": imgs})
                pending_bytes += sum(t.numel() * t.element_size() for t in imgs)
                if len(group) == batch:
                    pending_bytes -= sum(t.numel() * t.element_size() for g in group for t in g["imgs"])
                    _flush_group(model, group, scales, args, writer, store)
                elif pending_bytes > max_pending:
                    # VOC has hundreds of distinct image sizes: incomplete size groups may not pile up on the device
                    # without bound (34 MB per waiting 500x375 image) — the fullest one goes early
                    big = max(groups.values(), key=len)
                    pending_bytes -= sum(t.numel() * t.element_size() for g in big for t in g["imgs"])
                    _flush_group(model, big, scales, args, writer, store)
                _common.progress(process_id, n_gpus, it, len(databin))
            for group in groups.values():
                _flush_group(model, group, scales, args, writer, store)
    finally:
        writer.close()


def run(args):
    model = _common.ModelSpec(args.cam_network, "CAM", args.cam_weights_name + ".pth", strict=True)   # built by the worker(s)
    n_gpus = _common.n_gpus_or_raise(args)
    scales = tuple(float(s) for s in args.cam_scales)
    dataset = voc12_dataloader.VOC12ClassificationDatasetMSF(args.train_list, voc12_root=args.voc12_root,
                                                             scales=scales, raw=_common.device_preprocess(args))
    names = [voc12_dataloader.decode_int_filename(v) for v in dataset.img_name_list]
    dataset = torchutils.split_dataset(dataset, n_gpus)
    os.makedirs(args.cam_out_dir, exist_ok=True)
    args.cam_run_id = _common.new_cam_run(args.cam_out_dir)      # device-held CAMs of earlier runs of this directory go stale
    print("[ ", end="")
    _common.spawn_workers(_work, model, dataset, args)
    print("]")
    # which worker made (and, with keep_cams_on_device, still holds) every image's CAM: strided like the shards
    # (misc/torchutils.py:66-68) — the label steps of this run send an image to the worker that has its CAM
    owners = {name: i % n_gpus for i, name in enumerate(names)}
    owners["__n_workers__"] = n_gpus
    _common.CAM_OWNERS[os.path.abspath(args.cam_out_dir)] = owners
    torch.cuda.empty_cache()
