"""Shared plumbing of the label-generation steps: one worker process per GPU over strided shards
(reference step/make_cam.py:67-74), fail-fast, no communication between workers.

Unlike the reference, which forks a fresh set of workers inside every `run(args)` (`multiprocessing.spawn`,
step/make_cam.py:74, step/make_sem_seg_labels.py:70, step/make_ins_seg_labels.py:171), the workers here are spawned
ONCE per process and device list and serve every later step (`WorkerPool`): a HIP context, MIOpen's solver search and
the CAMs `make_cam` left in device memory (`CamStore`) survive from one step of `run_sample.py` to the next.  The step
API is unchanged: `step.X.run(args)` works on its own and returns when every shard is done."""
import atexit
import importlib
import os
import pickle
import traceback
from multiprocessing.reduction import ForkingPickler
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from torch import multiprocessing
from torch.utils.data import Subset

_NET_ALIASES = {"net.resnet50_cam": "irn_amd.net.resnet50_cam", "net.resnet50_irn": "irn_amd.net.resnet50_irn"}


def import_network(dotted):
    """`--cam_network net.resnet50_cam` style names of the reference resolve to this package."""
    return importlib.import_module(_NET_ALIASES.get(dotted, dotted))


class ModelSpec:
    """What a step hands its workers instead of a constructed network: the class and the checkpoint to build it from.
    The reference pickles the whole model into every spawned process (`spawn(args=(model, ...))`, step/make_cam.py:74);
    with persistent workers that would be ~270 shared-memory tensors (one file descriptor each) per worker and step.  A
    worker builds the network itself (`net/weights.load_checkpoint`: memory-mapped file, no random initialisation) and
    keeps it on its device for later steps that name the same checkpoint — make_sem_seg_labels and make_ins_seg_labels
    share one EdgeDisplacement."""

    def __init__(self, network, cls_name, path, strict):
        self.network, self.cls_name, self.path, self.strict = network, cls_name, os.path.abspath(path), bool(strict)

    def key(self):
        st = os.stat(self.path)
        return (self.network, self.cls_name, self.path, self.strict, st.st_mtime_ns, st.st_size)

    def build(self):
        from ..net import weights
        return weights.load_checkpoint(getattr(import_network(self.network), self.cls_name), self.path, strict=self.strict)


_MODELS = {}          # per process: ModelSpec.key() -> network (on this worker's device once a step has used it)


def materialise(model):
    """The network a `_work` function computes with: `model` itself, or the one a ModelSpec describes (cached per process;
    a checkpoint file that changed on disk is a different key)."""
    if not isinstance(model, ModelSpec):
        return model
    key = model.key()
    net = _MODELS.get(key)
    if net is None:
        for k in [k for k in _MODELS if k[:3] == key[:3]]:      # an older version of the same checkpoint: let it go
            del _MODELS[k]
        net = _MODELS[key] = model.build()
    return net


def worker_devices(args=None):
    """Device ordinal of every worker process: one per visible GPU like the reference (`torch.cuda.device_count()`
    shards, step/make_cam.py:67), unless `args.worker_devices` (run_sample.py --worker_devices, or the environment
    variable IRN_WORKER_DEVICES) lists them, e.g. "0,0" = two workers sharing GPU 0 (tests of the N > 1 path on a
    one-GPU box), "0,2,4,6" = every other GPU."""
    spec = getattr(args, "worker_devices", None) if args is not None else None
    if spec is None or spec == "" or spec == []:
        spec = os.environ.get("IRN_WORKER_DEVICES")
    if spec:
        devs = [int(v) for v in (spec.split(",") if isinstance(spec, str) else spec)]
        n = torch.cuda.device_count()
        if n < 1 or any(d < 0 or d >= n for d in devs):
            raise RuntimeError("worker_devices %s: %d GPU(s) visible" % (devs, n))
        return devs
    n = torch.cuda.device_count()
    if n < 1:
        # the reference silently does nothing with zero GPUs (spawn(nprocs=0)); fail loudly instead
        raise RuntimeError("irn_amd steps need at least one GPU (torch.cuda.device_count() == 0)")
    return list(range(n))


def n_gpus_or_raise(args=None):
    """Number of shards = number of worker processes (reference: torch.cuda.device_count())."""
    return len(worker_devices(args))


def worker_device(process_id, args=None):
    """The device worker `process_id` computes on.  The reference uses the process id itself
    (`torch.cuda.device(process_id)`, step/make_cam.py:24); a pool worker was told its device at start-up."""
    if _WORKER_DEVICE[0] is not None:
        return _WORKER_DEVICE[0]
    devs = worker_devices(args)
    return devs[process_id] if process_id < len(devs) else process_id


def miopen_cache_key():
    """(architecture, compute units, HIP version) as a directory name: MIOpen's find results and the GEMM rank table are only
    valid for the chip and library build that measured them.  From the device PROPERTIES (`gcnArchName`, e.g. gfx950), not from
    the marketing name: `torch.cuda.get_device_name` is "AMD Radeon Graphics" on boxes without amdgpu.ids and EMPTY under
    rocprofv3 — a profiled run then found no shipped database and silently ran the NCHW trunk (round 5, session 6)."""
    try:
        props = torch.cuda.get_device_properties(0)
        arch = str(getattr(props, "gcnArchName", "") or "unknown").split(":")[0]
        name = "%s-cu%d" % (arch, int(props.multi_processor_count))
    except Exception:
        name = "unknown_device"
    return "%s-hip%s" % (name, (torch.version.hip or "none").replace("/", "_"))


def merge_miopen_db(src_dir, dst_dir):
    """Add to the MIOpen user database in `dst_dir` every entry of the one in `src_dir` that it does not have yet (the
    find database `*.ufdb.txt` and the tuning database `*.udb.txt` are text files of `problem=solvers` lines; entries
    already present — what this machine measured itself — win).  -> number of entries added.  The channels-last mode of
    the trunk relies on this: an existing user database without the shipped NHWC entries would leave those problems
    untuned, which is 2x slower than not using the layout at all."""
    added = 0
    for f in sorted(os.listdir(src_dir)):
        if not (f.endswith(".ufdb.txt") or f.endswith(".udb.txt")):
            continue
        have = set()
        dst = os.path.join(dst_dir, f)
        if os.path.exists(dst):
            with open(dst) as fh:
                have = {line.split("=", 1)[0] for line in fh if "=" in line}
        with open(os.path.join(src_dir, f)) as fh:
            new = [line if line.endswith("\n") else line + "\n" for line in fh if "=" in line and line.split("=", 1)[0] not in have]
        if new:
            with open(dst, "a") as fh:
                fh.writelines(new)
            added += len(new)
    return added


_MIOPEN_LOCKS = []           # lock files held for the life of the process


def miopen_seed_root():
    """Directory of the find databases shipped with the package (one sub-directory per device / HIP version);
    IRN_MIOPEN_SEED_DIR names another one (A/B runs of a freshly tuned database)."""
    return os.environ.get("IRN_MIOPEN_SEED_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "miopen")


def gemm_table_root():
    """Directory of the shipped GEMM rank tables (`<arch>-cu<N>-hip<v>.json`, written by tools/conv1x1_tune.py)."""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "gemm")


def shipped_data_report():
    """What this build ships for THIS device / HIP version: {"key", "mode_key", "miopen" (is there a find database for
    the mode key), "miopen_keys" (the ones shipped), "gemm", "gemm_keys"}.  The tuned database and the rank table are this
    build's additions (all the reference says about solvers is `cudnn.enabled = True`, step/make_cam.py:14), so their
    absence is this build's to report."""
    def listing(root, strip=""):
        try:
            return sorted(f[:len(f) - len(strip)] if strip and f.endswith(strip) else f for f in os.listdir(root)
                          if not f.startswith("."))
        except OSError:
            return []
    key, mode_key = miopen_cache_key(), miopen_mode_key()
    mi_keys, ge_keys = listing(miopen_seed_root()), listing(gemm_table_root(), ".json")
    return {"key": key, "mode_key": mode_key, "miopen": mode_key in mi_keys and os.environ.get("IRN_MIOPEN_SEED", "1") != "0",
            "miopen_keys": mi_keys, "gemm": key in ge_keys and os.environ.get("IRN_GEMM_TABLE", "1") != "0", "gemm_keys": ge_keys}


_WARNED = set()


def warn_missing_shipped_data():
    """One `warnings.warn` per process and kind when the find database or the GEMM rank table for this (architecture, CU
    count, HIP version) is not shipped: the steps still run and still give the reference's results, ~10-20 % slower (NCHW
    trunk without the fused 1x1 GEMMs; hipBLASLt's first heuristic pick instead of the measured one) — a cliff that used to
    be silent.  -> the report."""
    import warnings
    rep = shipped_data_report()
    if rep["key"].startswith("unknown_device"):          # no GPU in this process (CPU tests, tooling): nothing runs slower
        return rep
    if not rep["miopen"] and "miopen" not in _WARNED:
        _WARNED.add("miopen")
        warnings.warn("irn_amd: no tuned MIOpen find database for '%s' under %s (shipped: %s): every trunk pass runs NCHW without "
                      "the fused 1x1 GEMMs, ~0.8x the throughput of the tuned channels-last trunk.  Remedy, on a box with this "
                      "GPU: python tools/miopen_warmup.py --channels-last 1, then python tools/miopen_det_filter.py (the "
                      "reproducible mode's '-det' database)." % (rep["mode_key"], miopen_seed_root(), ", ".join(rep["miopen_keys"]) or "none"),
                      RuntimeWarning, stacklevel=2)
    if not rep["gemm"] and "gemm" not in _WARNED:
        _WARNED.add("gemm")
        warnings.warn("irn_amd: no GEMM rank table for '%s' under %s (shipped: %s): the 1x1 convolutions use hipBLASLt's first "
                      "heuristic pick for every problem.  Remedy, on a box with this GPU: python tools/conv1x1_tune.py."
                      % (rep["key"], gemm_table_root(), ", ".join(rep["gemm_keys"]) or "none"), RuntimeWarning, stacklevel=2)
    return rep


def untuned_report(reset=True):
    """One line about the trunk passes of this process that took the NCHW branch (network-input sizes the shipped database
    is not tuned for: net/resnet50.channels_last_for), counts per size, or "" when there were none; printed at the end of
    every step by whoever ran it.  `reset` starts the next step's count."""
    from ..net import resnet50 as _r50
    sizes = dict(_r50.PASS_STATS["nchw_sizes"])
    n_cl, n_nchw, pad = _r50.PASS_STATS["channels_last"], _r50.PASS_STATS["nchw"], _r50.PASS_STATS["pad_rows"]
    if reset:
        _r50.PASS_STATS.update({"channels_last": 0, "nchw": 0, "pad_rows": 0, "nchw_sizes": {}})
    if not sizes:
        return ""
    top = sorted(sizes.items(), key=lambda kv: -kv[1])
    return ("irn_amd: %d of %d trunk passes ran NCHW (~0.8x) because their network-input size is not in the tuned database: %s%s"
            "%s; tools/miopen_warmup.py --channels-last 1 --sizes ... (or --from-list <image list>) adds sizes" % (
                n_nchw, n_nchw + n_cl, ", ".join("%s x%d" % kv for kv in top[:8]), ", ... (%d sizes)" % len(top) if len(top) > 8 else "",
                "; %d zero rows filled partial passes" % pad if pad else ""))


def say(line):
    """One whole line on stderr in ONE write: eight ranks starting at once must not interleave each other's lines."""
    import sys
    try:
        sys.stderr.flush()
        os.write(sys.stderr.fileno(), (line.rstrip("\n") + "\n").encode("utf-8", "replace"))
    except Exception:            # no real file behind sys.stderr (a capturing harness)
        print(line, file=sys.stderr, flush=True)


def startup_line(rank, n_workers, device, db_dir):
    """One line per worker / rank at start-up (stderr): which device it owns and whether the data this build's speed rests
    on was found for it — the things that differ between the box the defaults were measured on and an eight-GPU node met for
    the first time (reference step/make_cam.py:67-74 spawns its workers without a word)."""
    rep = shipped_data_report()
    try:
        props = torch.cuda.get_device_properties(int(device))
        chip = "%s, %d CUs, %.0f GB" % (str(getattr(props, "gcnArchName", "?")).split(":")[0], props.multi_processor_count, props.total_memory / 2 ** 30)
    except Exception:
        chip = "no GPU"
    try:
        from ..misc import indexing
        saved = indexing._saved_poll_delay(torch.device("cuda", int(device)))
        poll = "IRN_POLL_DELAY=%s" % os.environ["IRN_POLL_DELAY"] if os.environ.get("IRN_POLL_DELAY") else (
            "poll delay %d from %s" % (saved, indexing._poll_delay_file(torch.device("cuda", int(device)))) if saved
            else "poll delay: start-up probe at the first radius-10 batch")
    except Exception:
        poll = "poll delay: n/a"
    return ("irn_amd worker %d/%d: cuda:%d (%s) | %s backbones | MIOpen database '%s': %s, user database %s | GEMM rank table: %s | %s"
            % (int(rank), int(n_workers), int(device), chip, "reproducible" if deterministic_backbones() else "fast",
               rep["mode_key"], "shipped" if rep["miopen"] else "NOT SHIPPED (NCHW trunk)", db_dir, "shipped" if rep["gemm"] else "not shipped", poll))


def step_summary(rank, walker=None):
    """One line at the end of a step (stderr) when there is something to say: trunk passes that fell to the NCHW branch by
    size, and the walk's self-checks (block -> XCD placement, poll delay and where it came from, fall-backs)."""
    import sys
    parts = []
    msg = untuned_report()
    if msg:
        parts.append(msg)
    if walker is not None:
        try:
            t = walker.tuning()
            parts.append("irn_amd: walk radius %d: XCD placement %s, poll delay %d (%s), %d fall-back run(s)" % (
                walker.radius, {0: "not checked", 1: "round robin holds", 2: "does NOT hold (no XCD packing)"}.get(t["placement"], "?"),
                t["poll_delay"], getattr(walker, "poll_delay_source", "library default"), walker.fallback_runs))
        except Exception:
            pass
    for m in parts:
        say("[worker %d] %s" % (int(rank), m))


def check_split_overflow(what):
    """End of a step: did an activation leave fp16's range inside the split-precision 1x1 convolutions (net/resnet50.py
    SPLIT_GEMM)?  Then this step's outputs are invalid and the step raises — loudly, like a worker's exception."""
    from .. import ops
    if ops.split_overflowed():
        raise RuntimeError("%s: an activation of the backbone was beyond fp16's range (|x| > 65504) or NaN inside the split-precision "
                           "1x1 convolutions; the outputs of this step are INVALID.  Run with IRN_SPLIT_GEMM=0 (run_sample.py "
                           "--split_gemm 0): the fp32 GEMMs have no such limit." % what)


def deterministic_backbones():
    """IRN_DETERMINISTIC (default 1; run_sample.py --deterministic 0/1): the backbones' outputs are a function of their inputs
    only, whichever process, worker layout or run computes them — an N-GPU run writes bit for bit the files of a 1-GPU run
    (reference step/make_cam.py:67-74: any `n_gpus` must give the same files).  How (net/resnet50.channels_last_for):
      * input shapes the shipped database is tuned for run the channels-last trunk (fused GEMMs) on `<key>-det`, the tuned
        database WITHOUT the one order-dependent kind of kernel in it — MIOpen's NHWC implicit GEMM with a configuration that
        splits K across workgroups and adds with atomics (tools/miopen_det_filter.py: 86 of 264 records fall to the
        composable-kernel convolution); measured bit-identical across repeats and processes for every layer of CAM and IRNet
        at 8 pairs, `cam` 114.5 against 117.4 images/s (profiles/r05_s11_deterministic_database.txt);
      * every other convolution (untuned sizes, partial batches, the heads) runs under MIOpen's deterministic attribute
        (`torch.backends.cudnn.deterministic`) in NCHW — the attribute alone leaves no fast NHWC fp32 solver
        (profiles/r05_s3_deterministic_ab.txt), which is why the tuned shapes do not use it.
    IRN_DETERMINISTIC=0: the last 2.5 %, with split-K accumulations that move the CAMs by ~1e-5 from run to run (labels then
    differ only at exact ties; tests prove each).  The mode belongs to the PROCESS: it is fixed before the first convolution
    (MIOpen keeps the solver it resolved for a problem, whatever the attribute says later)."""
    return os.environ.get("IRN_DETERMINISTIC", "1") != "0"


def miopen_mode_key():
    """Directory name of the databases of this process's mode: `<arch>-cu<N>-hip<v>` or, reproducible mode, `…-det`."""
    return miopen_cache_key() + ("-det" if deterministic_backbones() else "")


def apply_deterministic_setting():
    """Hand the mode to the trunk (net/resnet50.DETERMINISTIC) and put PyTorch's process-wide flag in the mode's resting
    state: on in the reproducible mode (the trunk switches it off for the duration of a channels-last pass on the filtered
    database), off otherwise."""
    from ..net import resnet50 as _r50
    det = deterministic_backbones()
    _r50.DETERMINISTIC = det
    torch.backends.cudnn.deterministic = det


def miopen_setup(device_ordinal):
    """MIOpen's settings for the process that is about to run the backbones on `device_ordinal` — the SAME for a pool
    worker and for the in-process single-GPU path, so that one-GPU and N-GPU runs pick their convolution solvers the
    same way: find mode 2 ("fast": the find database first, the immediate-mode heuristic on a miss)
    unless the user chose one, and a user database directory that is
      * stable across runs — `$IRN_MIOPEN_CACHE` or `~/.cache/irn_amd/miopen`, / (device name, HIP version) / dev<ordinal>
        — so that what one run's workers found the next run's workers reuse (each worker process used to start with
        a cold database named after its rank);
      * completed from the database shipped with the package for this (device, HIP) pair (`irn_amd/data/miopen/<key>/`,
        written on a GPU box by `tools/miopen_warmup.py`): every shipped entry the user database lacks is added
        (`merge_miopen_db`), what this machine measured itself is kept — the trunk's channels-last layout, chosen per
        input shape (net/resnet50.py), depends on the shipped NHWC entries being there;
      * never shared by two live processes: the directory is claimed with an advisory lock, and a process that finds
        it taken (a second job on the same host and GPU) works on a private copy `dev<ordinal>-pid<pid>` instead
        (eight workers appending to one database collided in round 2).
    A `MIOPEN_USER_DB_PATH` the user exported is respected as the base.  Must run before the process's first
    convolution; returns the directory."""
    import fcntl
    import shutil
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    apply_deterministic_setting()
    warn_missing_shipped_data()
    marker = os.environ.get("IRN_MIOPEN_DB_SET")
    if marker and os.environ.get("MIOPEN_USER_DB_PATH") == marker and os.environ.get("IRN_MIOPEN_DB_DEV", str(int(device_ordinal))) == str(int(device_ordinal)):
        return marker                                    # this process (or the parent it inherited from, if it holds no lock itself) did it, for this device
    # (IRN_MIOPEN_BASE: the base the first caller of this job resolved — a worker must not take its parent's claimed
    # directory, which it inherits in MIOPEN_USER_DB_PATH, for the user's base)
    base = os.environ.get("IRN_MIOPEN_BASE") or os.environ.get("IRN_MIOPEN_CACHE") or os.environ.get("MIOPEN_USER_DB_PATH") or \
        os.path.join(os.path.expanduser("~"), ".cache", "irn_amd", "miopen")
    os.environ["IRN_MIOPEN_BASE"] = base
    key = miopen_mode_key()
    stable = os.path.join(base, key, "dev%d" % int(device_ordinal))
    os.makedirs(stable, exist_ok=True)
    use = stable
    try:
        fh = open(os.path.join(stable, ".lock"), "w")
        fcntl.flock(fh, fcntl.LOCK_EX | fcntl.LOCK_NB)
        _MIOPEN_LOCKS.append(fh)
    except OSError:                                      # another live process owns it: private copy
        use = os.path.join(base, key, "dev%d-pid%d" % (int(device_ordinal), os.getpid()))
        os.makedirs(use, exist_ok=True)
        for f in os.listdir(stable):
            if f != ".lock" and os.path.isfile(os.path.join(stable, f)):
                shutil.copy2(os.path.join(stable, f), os.path.join(use, f))
        atexit.register(shutil.rmtree, use, True)           # (a pool worker is terminated, not exited: WorkerPool.close() cleans up then)
    seed = os.path.join(miopen_seed_root(), key)
    if os.environ.get("IRN_MIOPEN_SEED", "1") != "0" and os.path.isdir(seed):
        merge_miopen_db(seed, use)         # (after the claim: nobody else appends to `use` now)
    os.environ["MIOPEN_USER_DB_PATH"] = use
    os.environ["IRN_MIOPEN_DB_SET"] = use
    os.environ["IRN_MIOPEN_DB_DEV"] = str(int(device_ordinal))
    return use


def reap_private_miopen_dbs(pids):
    """Private database copies (`dev<N>-pid<pid>`) of worker processes that are gone: what they found is merged back into the
    stable per-device directory (entries it lacks), the copy is removed.  Called by WorkerPool.close() — a terminated daemon
    process never runs its atexit handlers."""
    import re
    import shutil
    base = os.environ.get("IRN_MIOPEN_BASE")
    if not base:
        return 0
    root = os.path.join(base, miopen_mode_key())
    n = 0
    try:
        entries = os.listdir(root)
    except OSError:
        return 0
    for d in entries:
        m = re.match(r"^(dev\d+)-pid(\d+)$", d)
        if not m or int(m.group(2)) not in pids:
            continue
        src, stable = os.path.join(root, d), os.path.join(root, m.group(1))
        try:
            if os.path.isdir(stable):
                merge_miopen_db(src, stable)
            shutil.rmtree(src, ignore_errors=True)
            n += 1
        except OSError:
            pass
    return n


_WORKER_DEVICE = [None]      # set inside a pool worker process
_POOL = [None]               # the parent's pool


def _pool_worker(rank, device, n_workers, cmd_q, res_q):
    """Main loop of a pool worker: bound to one device for its whole life, runs the `_work` functions the parent names."""
    try:
        _WORKER_DEVICE[0] = int(device)
        if int(device) >= 0:                  # negative ordinals: workers without a GPU (the pool's own CPU tests)
            os.environ.pop("IRN_MIOPEN_DB_SET", None)    # the parent's claim is the parent's: this process makes its own
            os.environ.pop("IRN_MIOPEN_DB_DEV", None)
            db_dir = miopen_setup(int(device))           # stable per-device database, never shared by two live processes
            torch.cuda.set_device(int(device))
            say(startup_line(rank, n_workers, device, db_dir))
        res_q.put((rank, "ready", None))
        while True:
            cmd = cmd_q.get()
            if cmd is None:
                break
            res_q.put((rank, "ack", None))          # the command arrived: the parent tells "never received" from "never finished"
            try:
                mod, fn, model, shards, args = pickle.loads(cmd)
                work = getattr(importlib.import_module(mod), fn)
                work(rank, model, shards, args)
                if int(device) >= 0:
                    torch.cuda.synchronize()
                    step_summary(rank)
                    check_split_overflow("%s.%s" % (mod, fn))
                res_q.put((rank, "ok", {"cam_store_hits": CAM_STORE.hits, "cam_store_misses": CAM_STORE.misses,
                                        "edge_store_hits": EDGE_STORE.hits, "edge_store_misses": EDGE_STORE.misses,
                                        "walk_fallback_runs": WALK_STATS["fallback_runs"], "cam_trunk_passes": dict(CAM_STATS)}))
            except BaseException:
                res_q.put((rank, "error", traceback.format_exc()))
    except BaseException:
        res_q.put((rank, "error", traceback.format_exc()))


class WorkerPool:
    """One process per entry of `devices`, spawned once and alive until `close()` (or interpreter exit)."""

    def __init__(self, devices):
        self.devices = list(devices)
        ctx = multiprocessing.get_context("spawn")       # a forked child cannot use the parent's HIP runtime
        self._res_q = ctx.Queue()
        self._cmd_qs = [ctx.Queue() for _ in self.devices]
        self._procs = []
        self.stats = [{} for _ in self.devices]
        for rank, dev in enumerate(self.devices):
            p = ctx.Process(target=_pool_worker, args=(rank, dev, len(self.devices), self._cmd_qs[rank], self._res_q), daemon=True)
            p.start()
            self._procs.append(p)
        self._collect("ready", float(os.environ.get("IRN_WORKER_START_TIMEOUT_S", "900")), "worker start-up")

    def _collect(self, want, timeout_s=0.0, what="step"):
        """Wait until every worker has answered `want`.  Fail fast on a worker's exception or death (like
        spawn(join=True)); with `timeout_s` > 0 also on a worker that is alive but silent — wedged in a GPU or RCCL call,
        stuck behind another tenant of its device: the pool is stopped and the error names the ranks that never answered
        (and whether the command had reached them)."""
        import time
        answered, acked = set(), set()
        t0 = time.monotonic()
        while len(answered) < len(self._procs):
            try:
                rank, status, payload = self._res_q.get(timeout=1.0 if timeout_s > 0 else 5.0)
            except Exception:            # queue.Empty: is everybody still alive, and inside the deadline?
                dead = [i for i, p in enumerate(self._procs) if not p.is_alive()]
                if dead:
                    codes = [self._procs[i].exitcode for i in dead]
                    self.close(force=True)
                    raise RuntimeError("step worker(s) %s died without reporting (exit codes %s)" % (dead, codes))
                if timeout_s > 0 and time.monotonic() - t0 > timeout_s:
                    silent = sorted(set(range(len(self._procs))) - answered)
                    self.close(force=True)
                    raise RuntimeError("%s: worker(s) %s (device(s) %s) did not answer within %.0f s; the command had reached %s; "
                                       "all workers were stopped" % (what, silent, [self.devices[i] for i in silent], timeout_s,
                                                                     sorted(acked & set(silent)) or "none of them"))
                continue
            if status == "error":        # fail fast like spawn(join=True): the other workers are stopped, the error surfaces
                self.close(force=True)
                raise RuntimeError("step worker %d failed:\n%s" % (rank, payload))
            if status == "ack":
                acked.add(rank)
            elif status == want:
                if payload:
                    self.stats[rank] = payload
                answered.add(rank)

    def run(self, work, model, shards, args, timeout_s=0.0):
        # pickled HERE, not by the queue's feeder thread: an unpicklable model / shard / argument (a lambda, an open handle)
        # raises in the caller like the reference's spawn() does, instead of being printed by the feeder while the parent
        # waits for workers that never got a command
        # (once per worker: a pickled tensor carries a one-shot handle of its shared-memory file)
        blobs = [bytes(ForkingPickler.dumps((work.__module__, work.__name__, model, shards, args))) for _ in self._cmd_qs]
        for q, blob in zip(self._cmd_qs, blobs):
            q.put(blob)
        self._collect("ok", timeout_s, "%s.%s" % (work.__module__, work.__name__))

    def alive(self):
        return bool(self._procs) and all(p.is_alive() for p in self._procs)

    def close(self, force=False):
        for q, p in zip(self._cmd_qs, self._procs):
            if p.is_alive() and not force:
                try:
                    q.put(None)
                except Exception:
                    pass
        pids = set()
        for p in self._procs:
            p.join(timeout=0.1 if force else 20.0)
            if p.is_alive():
                p.terminate()
                p.join(timeout=5.0)
            if p.pid is not None and not p.is_alive():
                pids.add(p.pid)
        self._procs = []
        if pids:
            try:
                reap_private_miopen_dbs(pids)
            except Exception:
                pass


def get_pool(devices):
    pool = _POOL[0]
    if pool is not None and (pool.devices != list(devices) or not pool.alive()):
        pool.close()
        pool = _POOL[0] = None
    if pool is None:
        pool = _POOL[0] = WorkerPool(devices)
    return pool


def shutdown_workers():
    """Stop the worker processes (they otherwise live until the interpreter exits)."""
    if _POOL[0] is not None:
        _POOL[0].close()
        _POOL[0] = None


atexit.register(shutdown_workers)


def pool_stats():
    """Per-worker counters after the last step ({} entries before any): CAM hand-offs served from device memory, and
    batches the persistent walk handed to the streaming sweeps."""
    return [] if _POOL[0] is None else [dict(s) for s in _POOL[0].stats]


def spawn_workers(work, model, shards, args):
    """Run `work(process_id, model, shards, args)` for every shard, one worker process per shard, and wait for all of
    them (reference: `multiprocessing.spawn(_work, nprocs=n_gpus, args=(model, dataset, args), join=True)`).  One shard
    on the process's current device runs in-process; otherwise the shards go to the persistent pool.  `work` functions
    that never touch a GPU (CPU tests) may pass args without device information."""
    n = len(shards)
    try:
        devs = worker_devices(args if not isinstance(args, dict) else None)[:n]
    except RuntimeError:
        devs = None                      # no GPU: the callers that need one have raised already (n_gpus_or_raise)
    if devs is None or len(devs) != n:
        if n == 1:
            work(0, model, shards, args)
        else:
            multiprocessing.spawn(work, nprocs=n, args=(model, shards, args), join=True)
        return
    if n == 1 and not getattr(args, "always_use_workers", False):
        # same code path, no second process needed for a single GPU — but this IS the caller's process: its own
        # torch.backends.cudnn.deterministic and the trunk's mode are put back when the step returns
        from ..net import resnet50 as _r50
        saved = (torch.backends.cudnn.deterministic, _r50.DETERMINISTIC)
        try:
            db_dir = miopen_setup(devs[0])      # the same MIOpen settings a pool worker would run with
            if "startup" not in _WARNED:
                _WARNED.add("startup")
                say(startup_line(0, 1, devs[0], db_dir))
            work(0, model, shards, args)
            torch.cuda.synchronize()
            step_summary(0)
            check_split_overflow("%s.%s" % (work.__module__, work.__name__))
        finally:
            torch.backends.cudnn.deterministic, _r50.DETERMINISTIC = saved
        return
    get_pool(devs).run(work, model, shards, args, timeout_s=step_timeout(args))


def step_timeout(args=None):
    """Seconds a step may spend in its workers before the pool is stopped (0 = no limit, the reference's behaviour):
    `args.step_timeout` (run_sample.py --step_timeout) or the environment variable IRN_STEP_TIMEOUT_S."""
    v = getattr(args, "step_timeout", None) if args is not None and not isinstance(args, dict) else None
    if v in (None, "", 0, 0.0):
        v = os.environ.get("IRN_STEP_TIMEOUT_S", 0)
    return max(0.0, float(v or 0))


def split_by_owner(dataset, n_splits, names, owners, slack=0.25):
    """Shards for a label step that follows `make_cam` in the same run: image `names[i]` goes to the worker that holds
    its CAM in device memory (`owners[name]`, what `make_cam.run` recorded) as long as that worker's shard stays within
    (1 + slack) of the even share; everything else goes to the least loaded worker, like the reference's strided split
    (misc/torchutils.py:66-68) would balance it.  Which worker processes an image never changes its outputs."""
    n_items = len(names)
    cap = int(np.ceil(n_items / float(n_splits) * (1.0 + slack)))
    shards = [[] for _ in range(n_splits)]
    later = []
    for i, name in enumerate(names):
        o = owners.get(name)
        if o is not None and 0 <= o < n_splits and len(shards[o]) < cap:
            shards[o].append(i)
        else:
            later.append(i)
    for i in later:
        k = min(range(n_splits), key=lambda j: len(shards[j]))
        shards[k].append(i)
    return [Subset(dataset, np.asarray(sorted(s), dtype=np.int64)) for s in shards]


# make_cam's trunk passes of this process, by layout (net/resnet50.channels_last_for: channels-last only for input shapes the
# shipped find database is tuned for), and its size-group flushes (full groups of `cam_batch` images vs partial ones)
CAM_STATS = {"channels_last": 0, "nchw": 0, "full_group_flushes": 0, "partial_group_flushes": 0}      # (passes, not flushes: a partial group is one padded pass, an untuned size one pass per image — net/resnet50.run_rows)
CAM_OWNERS = {}       # abspath(cam_out_dir) -> {image name: worker that made (and still holds) its CAM}
WALK_STATS = {"fallback_runs": 0}


def label_step_shards(dataset, n_workers, args):
    """Shards of make_sem_seg_labels / make_ins_seg_labels: CAM-owner aware when make_cam ran before in this process
    with the same worker layout, else the reference's strided split."""
    from ..misc import torchutils
    owners = CAM_OWNERS.get(os.path.abspath(args.cam_out_dir)) if keep_cams(args) else None
    if owners and owners.get("__n_workers__") == n_workers and n_workers > 1:
        from ..voc12.dataloader import decode_int_filename
        return split_by_owner(dataset, n_workers, [decode_int_filename(v) for v in dataset.img_name_list], owners)
    return torchutils.split_dataset(dataset, n_workers)


def device_preprocess(args):
    """Steps build the multi-scale inputs on the GPU unless args.device_preprocess is set to False."""
    return bool(getattr(args, "device_preprocess", True))


_UPLOADS = []            # (event, page-locked staging buffer) of uploads still in flight


def upload(t, staging=None):
    """Host tensor -> device tensor WITHOUT stalling the host.  `t.cuda()` of pageable memory returns only when the stream has
    drained — in a step's loop that is "when the previous batch's trunk passes have finished", so the host could never
    enqueue the next batch under them and the GPU idled 10 % of make_cam (7.5 ms of host time per image inside `.cuda()`,
    `tools/make_cam_profile.py`, round 4).  Here the bytes go through a recycled page-locked buffer and cross on the copy
    stream; the current stream waits for the copy on the DEVICE (an event), the host moves on."""
    from .. import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    src = t.contiguous()
    nbytes = src.numel() * src.element_size()
    while _UPLOADS and _UPLOADS[0][0].query():          # buffers whose copy has landed go back to the pool
        PINNED.give(_UPLOADS.pop(0)[1])
    if nbytes == 0:
        return src.cuda()
    if staging is None:
        staging = PINNED.take(nbytes)
        staging[:nbytes].copy_(src.view(-1).view(torch.uint8) if src.dtype != torch.uint8 else src.view(-1))
    # (else: `t` already lives in the page-locked buffer `staging` — a loader thread put it there, make_loader)
    cs = ops._copy_stream(dev)
    with torch.cuda.stream(cs):
        out = staging[:nbytes].to(dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record(cs)
    cur = torch.cuda.current_stream(dev)
    cur.wait_event(done)
    out.record_stream(cur)                               # allocated on the copy stream, consumed on this one
    _UPLOADS.append((done, staging))
    return out.view(src.dtype).view(src.shape)


def device_images(pack, scales, normal=None):
    """Loader item -> list over scales of GPU fp32 [2,3,Hs,Ws] (image + horizontal flip).  Raw uint8 items
    (dataset raw=True) go through irn_msf_pack; items already in the reference's format are copied as is."""
    img = pack["img"]
    if torch.is_tensor(img) and img.dtype == torch.uint8 and img.numel() == 0:
        return None                                  # the dataset skipped the pixels (VOC12ClassificationDatasetMSF skip_image)
    if torch.is_tensor(img) and img.dtype == torch.uint8:
        from .. import ops
        kw = {} if normal is None else {"mean": normal.mean, "std": normal.std}
        return ops.msf_pack(upload(img[0], pack.pop("_staging", None)), scales, **kw)
    imgs = img if isinstance(img, (list, tuple)) else [img]
    return [upload(i[0]) for i in imgs]


class CamStore:
    """CAMs of this process kept on the device between steps: `make_cam` puts every image's
    {keys, cam} here besides writing the reference's `.npy` (step/make_cam.py:55-56), and the label steps take them
    from here instead of reading the 1-6 MB pickle back and uploading it again (step/make_sem_seg_labels.py:34-39).
    Entries are keyed by (cam_out_dir, image name) and a later `put` replaces an earlier one, so a second make_cam run
    (other weights, scales or output directory) can never be answered with the first run's CAMs.  A miss — another
    process or an earlier run made the CAM — falls back to the file, so results never depend on the store.  One store
    per process (= per worker and device); a 128x128 CAM is 64 KB per class, so all of VOC12 train_aug (10 582 images)
    is ~1 GB of the 288 GB: capped at `max_bytes` anyway (oldest entries leave first)."""

    def __init__(self, max_bytes=16 << 30):
        self._items = {}             # insertion-ordered
        self._bytes = 0
        self._max = max_bytes
        self.hits = self.misses = 0

    @staticmethod
    def _key(name, cam_out_dir):
        return (os.path.abspath(cam_out_dir) if cam_out_dir else "", name)

    def put(self, name, keys_cpu, keys_dev, cam, cam_out_dir=None, run_id=None):
        key = self._key(name, cam_out_dir)
        nbytes = cam.numel() * cam.element_size()
        old = self._items.pop(key, None)
        if old is not None:
            self._bytes -= old[2].numel() * old[2].element_size()
        while self._items and self._bytes + nbytes > self._max:
            dropped = self._items.pop(next(iter(self._items)))
            self._bytes -= dropped[2].numel() * dropped[2].element_size()
        if nbytes > self._max:
            return
        self._items[key] = (keys_cpu, keys_dev, cam, run_id)
        self._bytes += nbytes

    def get(self, name, cam_out_dir, device, run_id=None, use_store=True):
        """-> (keys int64 on the CPU, keys on the device, cam fp32 [K,h,w] on the device).  An entry is served only when
        the caller allows it (`use_store`: args.keep_cams_on_device) and it was made by the make_cam run whose stamp the
        output directory carries now (`run_id` = `current_cam_run(cam_out_dir)`): a later make_cam of the same directory —
        in this process, in another pool, with another worker layout — rewrote the files, and this worker's older CAMs
        must not answer for them."""
        hit = self._items.get(self._key(name, cam_out_dir)) if use_store else None
        if hit is not None and hit[2].device == device and hit[3] is not None and hit[3] == run_id:
            self.hits += 1
            return hit[:3]
        self.misses += 1
        d = np.load(os.path.join(cam_out_dir, name + ".npy"), allow_pickle=True).item()
        keys = torch.as_tensor(d["keys"])
        return keys, keys.to(device), torch.as_tensor(d["cam"]).to(device)

    def drop_dir(self, cam_out_dir):
        """Forget the entries of one output directory (make_cam.run starts with this: its files are about to change)."""
        d = os.path.abspath(cam_out_dir)
        for key in [k for k in self._items if k[0] == d]:
            old = self._items.pop(key)
            self._bytes -= old[2].numel() * old[2].element_size()

    def clear(self):
        self._items.clear()
        self._bytes = 0

    def __len__(self):
        return len(self._items)


CAM_STORE = CamStore()
_RUN_STAMP = ".irn_cam_run"


def new_cam_run(cam_out_dir):
    """make_cam.run: stamp the output directory with a fresh run id (and return it); the CAMs kept on the device carry
    it, the label steps compare it with the directory's."""
    import uuid
    run_id = uuid.uuid4().hex
    os.makedirs(cam_out_dir, exist_ok=True)
    tmp = os.path.join(cam_out_dir, _RUN_STAMP + ".tmp%d" % os.getpid())
    with open(tmp, "w") as f:
        f.write(run_id)
    os.replace(tmp, os.path.join(cam_out_dir, _RUN_STAMP))
    return run_id


def current_cam_run(cam_out_dir):
    """The run id make_cam last stamped `cam_out_dir` with, or None (CAM files written by something else)."""
    try:
        with open(os.path.join(cam_out_dir, _RUN_STAMP)) as f:
            return f.read().strip() or None
    except OSError:
        return None


def keep_cams(args):
    return bool(getattr(args, "keep_cams_on_device", True))


class EdgeStore:
    """Boundary and displacement maps of this process kept on the device between the two label steps.  The reference runs
    `EdgeDisplacement` once per image in make_ins_seg_labels (step/make_ins_seg_labels.py:119-129) and again, on the same
    image with the same weights, in make_sem_seg_labels (step/make_sem_seg_labels.py:28-34): 2 ms of a 2.7 ms image in the
    second step.  Whichever label step runs first puts {edge [1,h,w], dp [2,h,w]} here (196 KB at 512^2; all of VOC12
    train_aug is 1.5 GB), the other one takes them and skips its IRNet forward.  An entry is keyed by the network
    (class + checkpoint path + the file's mtime and size; a network not built from a ModelSpec gets no hand-off), the crop / stride of the forward, and the
    image FILE (path + mtime + size): other weights or another image under the same name is another key.  A miss is simply
    computed.  `args.keep_edges_on_device = False` (run_sample.py --keep_edges_on_device 0) switches it off.  Results: the
    maps are what the first step computed — identical to a recomputation up to MIOpen's choice of solver for another batch
    composition (the tail batch of a shard), like every batched forward here."""

    def __init__(self, max_bytes=8 << 30):
        self._items = {}
        self._bytes = 0
        self._max = max_bytes
        self.hits = self.misses = 0

    def get(self, key, device):
        hit = self._items.get(key)
        if hit is not None and hit[0].device == device:
            self.hits += 1
            return hit
        self.misses += 1
        return None

    def peek(self, key, device):
        """Is the entry there (no hit / miss accounting)?  For the loader threads' `skip_image`."""
        hit = self._items.get(key)
        return hit is not None and hit[0].device == device

    def put(self, key, edge, dp):
        nbytes = (edge.numel() + dp.numel()) * 4
        old = self._items.pop(key, None)
        if old is not None:
            self._bytes -= (old[0].numel() + old[1].numel()) * 4
        while self._items and self._bytes + nbytes > self._max:
            dropped = self._items.pop(next(iter(self._items)))
            self._bytes -= (dropped[0].numel() + dropped[1].numel()) * 4
        if nbytes <= self._max:
            self._items[key] = (edge, dp)
            self._bytes += nbytes

    def clear(self):
        self._items.clear()
        self._bytes = 0

    def __len__(self):
        return len(self._items)


EDGE_STORE = EdgeStore()


def keep_edges(args):
    return bool(getattr(args, "keep_edges_on_device", True))


def image_stamp(voc12_root, name):
    """(path, mtime, size) of an image file: part of the EdgeStore key."""
    from ..voc12.dataloader import get_img_path
    path = os.path.abspath(get_img_path(name, voc12_root))
    try:
        st = os.stat(path)
        return (path, st.st_mtime_ns, st.st_size)
    except OSError:
        return (path, -1, -1)


def walk_radius(args, default):
    """The reference hard-codes radius 5 at its call sites (step/make_sem_seg_labels.py:41,
    step/make_ins_seg_labels.py:135); `args.radius` (run_sample.py --radius) overrides it, e.g. 10 for BASELINE
    configs[2]."""
    return int(getattr(args, "radius", 0) or default)


def make_walker(args, default_radius):
    """The steps' random walk: radius as `walk_radius`, and the schedule switch — `args.walk_accel` (run_sample.py
    --walk_accel {0,1}; None = the environment variable IRN_WALK_ACCEL, else the library default 1) chooses between the
    truncated Chebyshev series of the operator and the reference's own count of 2^exp_times applications
    (misc/indexing.py:136-137); `args.walk_accel_tol_exp` = e moves the series' truncation bound to 10^-e."""
    from ..misc import indexing
    walker = indexing.RandomWalk(walk_radius(args, default_radius))
    accel = getattr(args, "walk_accel", None)
    if accel not in (None, ""):
        walker.set_option("accel", 1 if int(accel) else 0)
    tol = getattr(args, "walk_accel_tol_exp", None)
    if tol not in (None, "", 0):
        walker.set_option("accel_tol_exp", int(tol))
    return walker


def set_skip_image(databin, predicate):
    """Hand the step's `skip_image` predicate to the dataset behind a shard (a `Subset` of VOC12ClassificationDatasetMSF)."""
    ds = getattr(databin, "dataset", databin)
    if hasattr(ds, "skip_image"):
        ds.skip_image = predicate


def make_loader(databin, num_workers, prefetch=4):
    """Items of `databin` in order, each collated like `DataLoader(databin, batch_size=1, shuffle=False)` collates it
    (what the reference's steps iterate over, step/make_cam.py:22).  The reference's loader workers are processes; here
    they are THREADS: the work of an item is JPEG decoding (and, with device_preprocess off, PIL resizes), which runs
    inside Pillow with the GIL released, while forking worker processes from a process that holds a HIP context costs
    2-3 s per loader on the GPU box (the parent's address space is large) — more than a 1 000-image shard takes."""
    from torch.utils.data import default_collate
    n = len(databin)
    num_workers = min(int(num_workers), MAX_LOADER_THREADS)     # threads, not processes: more of them only fight over the GIL
    if num_workers <= 0:
        for i in range(n):
            yield default_collate([databin[i]])
        return
    from collections import deque
    pool = ThreadPoolExecutor(max_workers=num_workers)
    pending = deque()
    pin = torch.cuda.is_available()

    def load(i):
        pack = default_collate([databin[i]])
        img = pack.get("img") if isinstance(pack, dict) else None
        if pin and torch.is_tensor(img) and img.dtype == torch.uint8 and img.numel():
            # the decoded image goes into page-locked memory HERE, in the loader thread: `upload` then only issues the
            # asynchronous copy (the 0.8 MB memcpy cost the step's main thread 0.4 ms per image, round 5)
            nbytes = img.numel()
            buf = PINNED.take(nbytes)
            buf[:nbytes].copy_(img.reshape(-1))
            pack["img"] = buf[:nbytes].view(img.shape)
            pack["_staging"] = buf
        return pack

    try:
        nxt = 0
        depth = num_workers * prefetch
        while nxt < n or pending:
            while nxt < n and len(pending) < depth:
                pending.append(pool.submit(load, nxt))
                nxt += 1
            yield pending.popleft().result()
    finally:
        # an early close (an exception in the consumer, a generator dropped half-way) must not strand the page-locked buffers
        # of items nobody will consume: cancel what has not started, wait for what has, hand every buffer back
        pool.shutdown(wait=True, cancel_futures=True)
        for fut in pending:
            try:
                if not fut.cancelled():
                    buf = fut.result().pop("_staging", None)
                    if buf is not None:
                        PINNED.give(buf)
            except Exception:
                pass


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


def writer_threads(args, n_workers):
    """Encoder / writer threads of a worker.  Measured on the `steps` leg (128 images, `tools/steps_threads_probe.py`, round 3
    session 13): 4 loader + 4 writer threads 69.4 images/s, 8 + 8 69.0, 16 + 16 64.4, 32 + 16 56.5 — beyond a handful the
    threads only contend for the interpreter lock with the thread that drives the GPU."""
    return 4


MAX_LOADER_THREADS = 8      # same measurement: `--num_workers` defaults to half the host's cores (reference run_sample.py:11)


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
