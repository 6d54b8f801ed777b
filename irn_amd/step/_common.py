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


class CamStore:
    """CAMs of this process kept on the device between steps: `make_cam` puts every image's
    {keys, cam} here besides writing the reference's `.npy` (step/make_cam.py:55-56), and the label steps take them
    from here instead of reading the 1-6 MB pickle back and uploading it again (step/make_sem_seg_labels.py:34-39).
    A miss — another process or an earlier run made the CAM — falls back to the file, so results never depend on the
    store.  One store per process and device; a 128x128 CAM is 64 KB per class, so all of VOC12 train_aug (10 582
    images) is ~1 GB of the 288 GB: capped at `max_bytes` anyway."""

    def __init__(self, max_bytes=16 << 30):
        self._items = {}
        self._bytes = 0
        self._max = max_bytes
        self.hits = self.misses = 0

    def put(self, name, keys_cpu, keys_dev, cam):
        nbytes = cam.numel() * cam.element_size()
        if name in self._items or self._bytes + nbytes > self._max:
            return
        self._items[name] = (keys_cpu, keys_dev, cam)
        self._bytes += nbytes

    def get(self, name, cam_out_dir, device):
        """-> (keys int64 on the CPU, keys on the device, cam fp32 [K,h,w] on the device)."""
        hit = self._items.get(name)
        if hit is not None and hit[2].device == device:
            self.hits += 1
            return hit
        self.misses += 1
        import os
        import numpy as np
        d = np.load(os.path.join(cam_out_dir, name + ".npy"), allow_pickle=True).item()
        keys = torch.as_tensor(d["keys"])
        return keys, keys.to(device), torch.as_tensor(d["cam"]).to(device)

    def clear(self):
        self._items.clear()
        self._bytes = 0


CAM_STORE = CamStore()


def keep_cams(args):
    return bool(getattr(args, "keep_cams_on_device", True))


def walk_radius(args, default):
    """The reference hard-codes radius 5 at its call sites (step/make_sem_seg_labels.py:41,
    step/make_ins_seg_labels.py:135); `args.radius` (run_sample.py --radius) overrides it, e.g. 10 for BASELINE
    configs[2]."""
    return int(getattr(args, "radius", 0) or default)


def make_loader(databin, num_workers, prefetch=4):
    """Items of `databin` in order, each collated like `DataLoader(databin, batch_size=1, shuffle=False)` collates it
    (what the reference's steps iterate over, step/make_cam.py:22).  The reference's loader workers are processes; here
    they are THREADS: the work of an item is JPEG decoding (and, with device_preprocess off, PIL resizes), which runs
    inside Pillow with the GIL released, while forking worker processes from a process that holds a HIP context costs
    2-3 s per loader on the GPU box (the parent's address space is large) — more than a 1 000-image shard takes."""
    from torch.utils.data import default_collate
    n = len(databin)
    if num_workers <= 0:
        for i in range(n):
            yield default_collate([databin[i]])
        return
    from collections import deque
    pool = ThreadPoolExecutor(max_workers=num_workers)
    pending = deque()
    try:
        nxt = 0
        depth = num_workers * prefetch
        while nxt < n or pending:
            while nxt < n and len(pending) < depth:
                pending.append(pool.submit(lambda i=nxt: default_collate([databin[i]])))
                nxt += 1
            yield pending.popleft().result()
    finally:
        pool.shutdown(wait=False, cancel_futures=True)


def progress(process_id, n_workers, it, n_items):
    """The reference prints 5 % ticks from the last rank and divides by len//20 (ZeroDivision for
    shards under 20 images, step/make_cam.py:58); guarded here."""
    step = max(n_items // 20, 1)
    if process_id == n_workers - 1 and it % step == 0:
        print("%d " % ((5 * it + 1) // step), end="", flush=True)


class PinnedPool:
    """Page-locked staging buffers for device-to-host copies that must not stall the step: `take(nbytes)` hands out a
    uint8 buffer (recycled, rounded up to 1 MiB), `give(buf)` returns it.  Thread-safe (writer threads give back)."""

    def __init__(self):
        import threading
        self._free = {}
        self._lock = threading.Lock()

    def take(self, nbytes):
        size = max(1 << 20, (int(nbytes) + (1 << 20) - 1) >> 20 << 20)
        with self._lock:
            bucket = self._free.get(size)
            if bucket:
                return bucket.pop()
        return torch.empty(size, dtype=torch.uint8, pin_memory=True)

    def give(self, buf):
        with self._lock:
            self._free.setdefault(buf.numel(), []).append(buf)


PINNED = PinnedPool()


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
