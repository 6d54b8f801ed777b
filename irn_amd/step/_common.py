"""Shared plumbing of the label-generation steps: one worker process per GPU over strided shards
(reference step/make_cam.py:67-74), fail-fast, no communication between workers."""
import importlib
from concurrent.futures import ThreadPoolExecutor

import torch
from torch import multiprocessing

_NET_ALIASES = {"net.resnet50_cam": "irn_amd.net.resnet50_cam", "net.resnet50_irn": "irn_amd.net.resnet50_irn"}


def import_network(dotted):
    """`--cam_network net.resnet50_cam` style names of the reference resolve to this package."""
    return importlib.import_module(_NET_ALIASES.get(dotted, dotted))


def n_gpus_or_raise():
    n = torch.cuda.device_count()
    if n < 1:
        # the reference silently does nothing with zero GPUs (spawn(nprocs=0)); fail loudly instead
        raise RuntimeError("irn_amd steps need at least one GPU (torch.cuda.device_count() == 0)")
    return n


def spawn_workers(work, model, shards, args):
    n = len(shards)
    if n == 1:
        work(0, model, shards, args)        # same code path, no fork needed for a single GPU
    else:
        multiprocessing.spawn(work, nprocs=n, args=(model, shards, args), join=True)


def device_preprocess(args):
    """Steps build the multi-scale inputs on the GPU unless args.device_preprocess is set to False."""
    return bool(getattr(args, "device_preprocess", True))


def device_images(pack, scales, normal=None):
    """Loader item -> list over scales of GPU fp32 [2,3,Hs,Ws] (image + horizontal flip).  Raw uint8 items
    (dataset raw=True) go through irn_msf_pack; items already in the reference's format are copied as is."""
    img = pack["img"]
    if torch.is_tensor(img) and img.dtype == torch.uint8:
        from .. import ops
        kw = {} if normal is None else {"mean": normal.mean, "std": normal.std}
        return ops.msf_pack(img[0].cuda(non_blocking=True), scales, **kw)
    imgs = img if isinstance(img, (list, tuple)) else [img]
    return [i[0].cuda(non_blocking=True) for i in imgs]


def progress(process_id, n_workers, it, n_items):
    """The reference prints 5 % ticks from the last rank and divides by len//20 (ZeroDivision for
    shards under 20 images, step/make_cam.py:58); guarded here."""
    step = max(n_items // 20, 1)
    if process_id == n_workers - 1 and it % step == 0:
        print("%d " % ((5 * it + 1) // step), end="", flush=True)


class AsyncWriter:
    """Output files (4 MB CAM dictionaries, PNG label maps, detection dictionaries) are encoded and
    written by a small thread pool while the GPU works on the next images; at
    most `max_pending` writes are in flight, errors surface at the next submit or at close()."""

    def __init__(self, threads=4, max_pending=32):
        self._pool = ThreadPoolExecutor(max_workers=threads)
        self._pending = []
        self._max = max_pending

    def _reap(self, keep):
        while len(self._pending) > keep:
            self._pending.pop(0).result()        # re-raises a failed write

    def submit(self, fn, *args, **kw):
        self._reap(self._max - 1)
        self._pending.append(self._pool.submit(fn, *args, **kw))

    def close(self):
        try:
            self._reap(0)
        finally:
            self._pool.shutdown(wait=True)
