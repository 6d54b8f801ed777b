"""Semantic pseudo-label step — drop-in for reference step/make_sem_seg_labels.py (`run(args)`).

Reads  args.irn_network, args.irn_weights_name, args.infer_list, args.voc12_root, args.cam_out_dir,
       args.beta, args.exp_times, args.sem_seg_bg_thres, args.num_workers
Writes args.sem_seg_out_dir/<name>.png  uint8 [H,W] (0 = background, class+1 otherwise)

Per image (step/make_sem_seg_labels.py:28-51): EdgeDisplacement forward (PyTorch-ROCm) -> edge; the image's CAM —
from the device store when make_cam ran in this process, else from its file; random walk of the CAMs over the edge
affinities and the label epilogue run in libirn_hip.so.  Unlike the reference's batch-1 loop the IRNet forward runs
`irn_batch` images per pass (default 8) and the walk is issued for `walk_batch` images at a time (default 64) so one
launch fills the GPU for several rounds; results per image are unchanged.  `args.radius` (default 5 = the
reference's hard-coded value) selects the walk radius.
"""
import os

import numpy as np
import torch
from PIL import Image

from .. import ops
from ..voc12 import dataloader as voc12_dataloader
from . import _common

RADIUS = 5   # hard-coded at the reference call site (step/make_sem_seg_labels.py:41)


def _save_png(path, label):
    Image.fromarray(label).save(path)


def edges_for(model, pend, irn_batch, store=None, model_key=None, dump_dir=None):
    """EdgeDisplacement forward for the pending images, `irn_batch` at a time: ragged images are padded to the
    512^2 crop like the reference pads each one (net/resnet50_irn.py:225), so a chunk is ONE trunk pass.  With a `store`
    (`_common.EDGE_STORE`) the maps the other label step already computed for an image (same network, same file) are taken
    from device memory instead, and what is computed here is left there for it."""
    todo = pend
    if store is not None:
        todo = []
        for p in pend:
            key = (model_key, getattr(model, "crop_size", None), getattr(model, "stride", None), p.get("stamp"))
            dev = p["dev"] if "dev" in p else p["img"].device
            hit = store.get(key, dev) if p.get("stamp") is not None else None
            if hit is not None:
                p["edge"], p["dp"] = hit
                p.pop("img", None)
            else:
                p["_edge_key"] = key if p.get("stamp") is not None else None
                todo.append(p)
    for p in todo:
        if p.get("img") is None:
            # the loader skipped this image's pixels because the store held its maps then, and they are gone now (evicted
            # between the loader's look and this one): decode it here
            img = torch.from_numpy(np.array(Image.open(p["stamp"][0]).convert("RGB")))
            p["img"] = _common.device_images({"img": img[None]}, (1.0,))[0]
    for i in range(0, len(todo), irn_batch):
        chunk = todo[i:i + irn_batch]
        for p, (edge, dp) in zip(chunk, model.forward_batch([p.pop("img") for p in chunk])):
            p["edge"], p["dp"] = edge, dp
            key = p.pop("_edge_key", None)
            if store is not None and key is not None:
                store.put(key, edge, dp.clone())         # dp is a view into the batch's output: keep 1 image, not 8
    if dump_dir:
        # verification aid (args.edge_out_dir, off by default): the maps this step walks on, per image, so that a run can be
        # checked against the oracle on ITS OWN inputs (the reference keeps them in memory only, step/make_sem_seg_labels.py:28-34).
        # Synchronous on purpose: not a production path
        os.makedirs(dump_dir, exist_ok=True)
        for p in pend:
            np.save(os.path.join(dump_dir, p["name"] + ".npy"), {"edge": p["edge"].cpu().numpy(), "dp": p["dp"].cpu().numpy()})


def skip_image_predicate(model, args, device):
    """`skip_image` for the step's dataset: True for images whose boundary / displacement maps the other label step left on
    this device (the EdgeStore entry `edges_for` will hit) — their JPEG need not be decoded or uploaded."""
    kw = _edge_store_kw(model, args)
    store = kw.get("store")
    if store is None:
        return None
    mk, crop, stride = kw["model_key"], getattr(model, "crop_size", None), getattr(model, "stride", None)
    return lambda name: store.peek((mk, crop, stride, _common.image_stamp(args.voc12_root, name)), device)


_MODEL_KEYS = {}          # id(network) -> (weak reference to it, key of the ModelSpec it was built from); set in _work


def remember_model(model, spec_key):
    """Note which checkpoint (class + path + mtime + size) a worker built `model` from: the identity the EdgeStore keys
    its entries by.  The weak reference guards against a later object reusing the id of a freed network."""
    import weakref
    for k in [k for k, (ref, _) in _MODEL_KEYS.items() if ref() is None]:
        del _MODEL_KEYS[k]
    _MODEL_KEYS[id(model)] = (weakref.ref(model), spec_key)


def model_key(model):
    hit = _MODEL_KEYS.get(id(model))
    return hit[1] if hit is not None and hit[0]() is model else None


def _edge_store_kw(model, args):
    """Keyword arguments of `edges_for` that switch the device hand-off of the edge maps on (see _common.EdgeStore)."""
    dump = {"dump_dir": args.edge_out_dir} if getattr(args, "edge_out_dir", None) else {}
    if not _common.keep_edges(args):
        return dump
    key = model_key(model)
    if key is None:
        # a network handed over as a module (not built from a ModelSpec) has no identity that survives an in-place weight
        # update or a recycled id(): no hand-off for it, every step computes its own maps
        return dump
    return {"store": _common.EDGE_STORE, "model_key": key, **dump}


def _start_copy(batch):
    """The batch's label maps (views of one device buffer) leave for the host as ONE copy into page-locked memory on the
    copy stream, behind everything enqueued so far; `_collect` waits for it when the next batch is already queued."""
    flat = batch["flat"]
    staging = _common.PINNED.take(flat.numel())
    ready = torch.cuda.Event()
    ready.record()
    cs = ops._copy_stream(flat.device)
    cs.wait_event(ready)
    with torch.cuda.stream(cs):
        staging[:flat.numel()].copy_(flat, non_blocking=True)
        done = torch.cuda.Event()
        done.record(cs)
    batch["staging"], batch["done"] = staging, done


def _enqueue(model, walker, pend, args):
    """IRNet forward, walk, label epilogue and the transfer of the label maps of the images in `pend`, enqueued only;
    returns what `_collect` needs."""
    edges_for(model, pend, int(getattr(args, "irn_batch", 0) or 8), **_edge_store_kw(model, args))
    rws = walker([p["edge"] for p in pend], [p["cam"] for p in pend],
                 beta=float(args.beta), exp_times=int(args.exp_times))
    sizes, keys = [p["size"] for p in pend], [p["keys_dev"] for p in pend]
    out = ops.label_epilogue(rws, sizes, float(args.sem_seg_bg_thres), keys=keys, packed=True)
    batch = {"names": [p["name"] for p in pend], "rws": rws, "sizes": sizes, "keys": keys, "flat": out["labels_flat"]}
    _start_copy(batch)
    pend.clear()
    return batch


def _collect(walker, batch, args, writer):
    """Wait for the batch enqueued last and hand its label maps (already on their way to the host) to the writer threads."""
    if batch is None:
        return
    if walker.sync():                       # the persistent walk gave up and the batch was re-run on the streaming sweeps
        _common.PINNED.give(batch["staging"])
        batch["flat"] = ops.label_epilogue(batch["rws"], batch["sizes"], float(args.sem_seg_bg_thres), keys=batch["keys"],
                                           packed=True)["labels_flat"]
        _start_copy(batch)
    batch["done"].synchronize()
    host = batch["staging"].numpy()
    off = 0
    for name, (hh, ww) in zip(batch["names"], batch["sizes"]):
        lab = host[off:off + hh * ww].reshape(hh, ww).copy()      # 256 KB: the staging buffer goes back at once
        off += hh * ww
        writer.submit(_save_png, os.path.join(args.sem_seg_out_dir, name + ".png"), lab)
    _common.PINNED.give(batch["staging"])


def _flush(model, walker, pend, args, writer):
    """One batch start to finish (kept for callers that want the blocking form)."""
    if pend:
        _collect(walker, _enqueue(model, walker, pend, args), args, writer)


def _work(process_id, model, dataset, args):
    spec_key = model.key() if isinstance(model, _common.ModelSpec) else None
    model = _common.materialise(model)      # a network, or the (class, checkpoint) a worker builds it from
    if spec_key is not None:
        remember_model(model, spec_key)
    databin = dataset[process_id]
    n_gpus = len(dataset)
    _common.set_skip_image(databin, skip_image_predicate(model, args, torch.device("cuda", _common.worker_device(process_id, args))))
    loader = _common.make_loader(databin, int(args.num_workers) // n_gpus)
    batch = int(getattr(args, "walk_batch", 0) or 64)   # 64 VOC-size images = 3-4 rounds of the resident walk
    writer = _common.AsyncWriter(threads=_common.writer_threads(args, n_gpus))
    try:
        dev_id = _common.worker_device(process_id, args)
        with torch.no_grad(), torch.cuda.device(dev_id):
            model.cuda()
            dev = torch.device("cuda", dev_id)
            walker = _common.make_walker(args, RADIUS)
            # a batch is enqueued and left running while the loop gathers the next one (decoded images from the loader
            # threads, their uploads, the CAMs); it is collected just before the next batch is enqueued
            pend, running = [], None
            cam_run, use_store = _common.current_cam_run(args.cam_out_dir), _common.keep_cams(args)
            for it, pack in enumerate(loader):
                name = pack["name"][0]
                if not isinstance(name, str):
                    name = voc12_dataloader.decode_int_filename(name)
                size = (int(pack["size"][0]), int(pack["size"][1]))
                # CAM of this image: still on the device when make_cam ran in this process, else from its file
                _keys, keys_dev, cam = _common.CAM_STORE.get(name, args.cam_out_dir, dev, cam_run, use_store)
                imgs = _common.device_images(pack, (1.0,))
                pend.append({"name": name, "size": size, "img": None if imgs is None else imgs[0], "dev": dev,
                             "cam": cam, "keys_dev": keys_dev, "stamp": _common.image_stamp(args.voc12_root, name)})
                if len(pend) == batch:
                    _collect(walker, running, args, writer)
                    running = _enqueue(model, walker, pend, args)
                _common.progress(process_id, n_gpus, it, len(databin))
            _collect(walker, running, args, writer)
            _flush(model, walker, pend, args, writer)
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
    os.makedirs(args.sem_seg_out_dir, exist_ok=True)
    print("[", end="")
    _common.spawn_workers(_work, model, dataset, args)
    print("]")
    torch.cuda.empty_cache()
