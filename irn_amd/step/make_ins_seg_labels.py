"""Instance pseudo-label step — drop-in for reference step/make_ins_seg_labels.py (`run(args)`).

Reads  args.irn_network, args.irn_weights_name, args.infer_list, args.voc12_root, args.cam_out_dir,
       args.beta, args.exp_times, args.ins_seg_bg_thres, args.num_workers
Writes args.ins_seg_out_dir/<name>.npy = {'score': float[N], 'mask': bool[N,H,W], 'class': int64[N]}
       (schema consumed by step/make_cocoann.py:34-38 and step/eval_ins_seg.py)

Per image (step/make_ins_seg_labels.py:119-152): EdgeDisplacement forward -> edge, dp; centroid
refinement, clustering, per-instance CAM split, random walk, label epilogue and the per-mask
connected components all run on the GPU through libirn_hip.so.  An image with no detected instance
is skipped with a warning (the reference crashes in np.stack([])).
"""
import os
import warnings

import numpy as np
import torch

from .. import ops
from ..voc12 import dataloader as voc12_dataloader
from . import _common, make_sem_seg_labels

RADIUS = 5   # step/make_ins_seg_labels.py:135

find_centroids_with_refinement = ops.find_centroids_with_refinement
cluster_centroids = ops.cluster_centroids
detect_instance = ops.detect_instance


def instance_front(items):
    """First half of step/make_ins_seg_labels.py:131-150 for a batch: centroid refinement and clustering, ENQUEUED only — the
    instance counts K stay on the device.  -> what `instance_back` needs."""
    dps = [it["dp"] for it in items]
    return ops.cluster_centroids_batch(ops.find_centroids_batch(dps), dps, k_on_device=True)


def instance_back(walker, items, front, beta, exp_times, bg_thres, deferred=False):
    """Second half: reads the K of `instance_front` (the first host round trip of the batch), then the per-instance CAM split +
    random walk, the label epilogue and the detections."""
    cmaps, k_dev = front
    ks = [int(k) for k in k_dev.cpu().tolist()]
    rws = walker([it["edge"] for it in items], [it["cam"] for it in items], beta=beta, exp_times=exp_times,
                 inst_maps=cmaps, k_inst=ks)
    ep = ops.label_epilogue(rws, [it["size"] for it in items], bg_thres, want_labels=False, want_argmax=True,
                            want_rw_up=True)
    if walker.sync():                       # the persistent walk gave up and the batch was re-run on the streaming sweeps
        ep = ops.label_epilogue(rws, [it["size"] for it in items], bg_thres, want_labels=False, want_argmax=True,
                                want_rw_up=True)
    n_ch = [it["cam"].shape[0] * k for it, k in zip(items, ks)]
    class_ids = [np.repeat(np.asarray(torch.as_tensor(it["keys"]).cpu()), k) for it, k in zip(items, ks)]
    return ops.detect_instance_batch(ep["rw_up"], ep["argmax"], class_ids, n_ch,
                                     [it["size"][0] * it["size"][1] * 0.01 for it in items], deferred=deferred)


def instance_labels_batch(walker, items, beta, exp_times, bg_thres, deferred=False):
    """step/make_ins_seg_labels.py:131-150 for a batch of images.  items: dicts with GPU tensors
    `edge` [1,h,w], `dp` [2,h,w], `cam` [C,h,w], CPU/GPU `keys` [C] and `size` (H, W).  Every stage runs ONCE for the
    whole batch — centroid refinement, clustering, random walk (a 128x128 grid is 16 tiles at radius 5: one image uses
    1/16 of the GPU), label epilogue, detection — with three host round trips per BATCH: the instance counts K (they
    size the walk's channels), the detection counts, and the packed detections.  Returns a list of detection dicts
    (or the ValueError of an image without detections, in its slot); with `deferred=True` an `ops.PendingDetections`
    whose `result()` is that list — the packed transfer then runs under whatever the caller enqueues next."""
    return instance_back(walker, items, instance_front(items), beta, exp_times, bg_thres, deferred=deferred)


def instance_labels(walker, edge, dp, cams, keys, size, beta, exp_times, bg_thres):
    """One image: returns the detection dict (numpy) — step/make_ins_seg_labels.py:131-150."""
    det = instance_labels_batch(walker, [{"edge": edge, "dp": dp, "cam": cams, "keys": keys, "size": size}],
                                beta, exp_times, bg_thres)[0]
    if isinstance(det, Exception):
        raise det
    return det


def _write(names, pending, args, writer):
    for name, det in zip(names, pending.result()):
        if isinstance(det, Exception):
            warnings.warn("%s: %s — no file written" % (name, det))
            continue
        writer.submit(np.save, os.path.join(args.ins_seg_out_dir, name + ".npy"), det)


def _flush(model, walker, pend, args, writer, state):
    """One turn of the step's three-stage pipeline; `state` = {"front": batch whose IRNet forward + clustering are enqueued,
    "emit": batch whose detections are crossing PCIe}.  In this order:
      1. the batch in `front` gets its back half (the K read-back, by now long computed; walk, epilogue, detections) —
         BEFORE anything of the new batch is enqueued, so that its two small read-backs never queue behind 45 ms of IRNet;
      2. the batch in `pend` gets its IRNet forward, centroid refinement and clustering enqueued — the GPU works on them while
         the caller's loop decodes and uploads the next batch (round 5: the K read-back used to stall that loop for the whole
         forward, 40 % of the step);
      3. the batch in `emit` is collected and handed to the writer threads."""
    done = None
    if state.get("front") is not None:
        items, front = state["front"]
        done = ([it["name"] for it in items],
                instance_back(walker, items, front, float(args.beta), int(args.exp_times), float(args.ins_seg_bg_thres), deferred=True))
        state["front"] = None
    if pend:
        make_sem_seg_labels.edges_for(model, pend, int(getattr(args, "irn_batch", 0) or 8),
                                      **make_sem_seg_labels._edge_store_kw(model, args))
        items = list(pend)
        pend.clear()
        state["front"] = (items, instance_front(items))
    if state.get("emit") is not None:
        _write(*state["emit"], args, writer)
    state["emit"] = done


def _work(process_id, model, dataset, args):
    spec_key = model.key() if isinstance(model, _common.ModelSpec) else None
    model = _common.materialise(model)      # a network, or the (class, checkpoint) a worker builds it from
    if spec_key is not None:
        make_sem_seg_labels.remember_model(model, spec_key)
    databin = dataset[process_id]
    n_gpus = len(dataset)
    _common.set_skip_image(databin, make_sem_seg_labels.skip_image_predicate(
        model, args, torch.device("cuda", _common.worker_device(process_id, args))))
    loader = _common.make_loader(databin, int(args.num_workers) // n_gpus)
    batch = int(getattr(args, "walk_batch", 0) or 32)   # images per walk launch (results per image unchanged)
    writer = _common.AsyncWriter(threads=_common.writer_threads(args, n_gpus))
    try:
        dev_id = _common.worker_device(process_id, args)
        with torch.no_grad(), torch.cuda.device(dev_id):
            model.cuda()
            dev = torch.device("cuda", dev_id)
            walker = _common.make_walker(args, RADIUS)
            pend, state = [], {}
            cam_run, use_store = _common.current_cam_run(args.cam_out_dir), _common.keep_cams(args)
            for it, pack in enumerate(loader):
                name = pack["name"][0]
                if not isinstance(name, str):
                    name = voc12_dataloader.decode_int_filename(name)
                size = (int(pack["size"][0]), int(pack["size"][1]))
                keys, _keys_dev, cam = _common.CAM_STORE.get(name, args.cam_out_dir, dev, cam_run, use_store)
                imgs = _common.device_images(pack, (1.0,))
                pend.append({"name": name, "size": size, "img": None if imgs is None else imgs[0], "dev": dev,
                             "cam": cam, "keys": keys, "stamp": _common.image_stamp(args.voc12_root, name)})
                if len(pend) == batch:
                    _flush(model, walker, pend, args, writer, state)
                _common.progress(process_id, n_gpus, it, len(databin))
            _flush(model, walker, pend, args, writer, state)      # the last batch's front half ...
            _flush(model, walker, pend, args, writer, state)      # ... its back half ...
            _flush(model, walker, pend, args, writer, state)      # ... and its collection
            _common.WALK_STATS["fallback_runs"] += walker.fallback_runs
            _common.step_summary(process_id, walker)
            walker.close()
    finally:
        writer.close()


def run(args):
    model = _common.ModelSpec(args.irn_network, "EdgeDisplacement", args.irn_weights_name, strict=False)   # built by the worker(s)
    n_gpus = _common.n_gpus_or_raise(args)
    dataset = voc12_dataloader.VOC12ClassificationDatasetMSF(args.infer_list, voc12_root=args.voc12_root,
                                                             scales=(1.0,), raw=_common.device_preprocess(args))
    dataset = _common.label_step_shards(dataset, n_gpus, args)     # strided (misc/torchutils.py:66-68) or CAM-owner aware
    os.makedirs(args.ins_seg_out_dir, exist_ok=True)
    print("[ ", end="")
    _common.spawn_workers(_work, model, dataset, args)
    print("]")
