"""Multi-GPU plumbing of the hot path: one process per GPU, images strided over ranks, no
collective on the data path (reference misc/torchutils.py:66-68 + multiprocessing.spawn at
step/make_cam.py:74).  The only communication is the benchmark's barrier / max-over-ranks timing and
an optional fan-in of finished label maps to rank 0 for in-memory consumers.

Two groups per job:
  * the CONTROL group is always gloo over the launcher's own rendezvous (MASTER_ADDR / MASTER_PORT, or the agent
    store `torch.distributed.run` hands its ranks) — it cannot fail for GPU reasons and it is what every rank uses to
    agree on what happened to the other one;
  * the RCCL group (`backend "nccl"` IS RCCL on ROCm) is a sub-group created beside it and probed with one
    all-reduce from a helper thread under a deadline (`IRN_RCCL_PROBE_TIMEOUT_S`, default 60 s).  The ranks then
    agree over the control group (MIN of the outcome): RCCL is used only if EVERY rank completed the probe.  A start-up
    that throws on one rank, or hangs on any, ends in the gloo group on all ranks — inside a minute, with the same
    rendezvous, whoever launched the job (round 3's fall-back re-rendezvoused on MASTER_PORT + 1, where nothing
    listens under `torch.distributed.run`'s agent store, and relied on every rank seeing the failure).
"""
import os
import sys
import threading

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def rank_device_ordinal(local_rank, rank_devices=None):
    """Device ordinal of a rank: its LOCAL_RANK (one process per GPU), unless `rank_devices` ("0,0", a list, or the
    environment variable IRN_RANK_DEVICES) maps local ranks to ordinals — the bench-side twin of the steps'
    `worker_devices`, e.g. "0,0" = two ranks sharing GPU 0 to exercise the N > 1 path on a one-GPU box."""
    spec = rank_devices if rank_devices not in (None, "", []) else os.environ.get("IRN_RANK_DEVICES")
    if not spec:
        return int(local_rank)
    devs = [int(v) for v in (spec.split(",") if isinstance(spec, str) else spec)]
    if local_rank >= len(devs):
        raise RuntimeError("rank_devices %s has no entry for local rank %d" % (devs, local_rank))
    return devs[local_rank]


def devices_shared(rank_devices=None):
    """True when two local ranks are mapped onto one device: RCCL cannot form a communicator then (one rank per
    device is its rule), so `auto` goes straight to gloo and an explicit `nccl` is an error."""
    spec = rank_devices if rank_devices not in (None, "", []) else os.environ.get("IRN_RANK_DEVICES")
    if not spec:
        return False
    devs = [int(v) for v in (spec.split(",") if isinstance(spec, str) else spec)]
    return len(set(devs)) < len(devs)


def init_process_group(backend=None, device=None, timeout_s=None):
    """Joins the job described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run) with ONE group of the given
    backend.  backend 'nccl' is RCCL on ROCm; 'gloo' serves the control group and the CPU tests."""
    import torch.distributed as dist
    rank, _, world = rank_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    if timeout_s:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


# The RCCL sub-group's OWN timeout.  It must never be the deadline that fires: when ProcessGroupNCCL's watchdog sees a
# collective older than the group timeout it tears the communicator down and, with the default
# TORCH_NCCL_ASYNC_ERROR_HANDLING, takes the whole process with it — before the ranks could agree on gloo, or minutes later in
# the middle of the benchmark.  So the group gets a timeout far beyond any run of this job, the watchdog is
# told not to kill the process, and the only deadline that decides anything is the Python one around the probe thread.
RCCL_GROUP_TIMEOUT_S = float(os.environ.get("IRN_RCCL_GROUP_TIMEOUT_S", "7200"))
# gloo control group: ranks may be skewed by minutes (a cold MIOpen find on one of them) without anything being wrong
CONTROL_TIMEOUT_S = float(os.environ.get("IRN_CONTROL_TIMEOUT_S", "1800"))


def _rccl_probe(device, timeout_s):
    """Create the RCCL sub-group and run one all-reduce on `device`.  -> group.  `timeout_s` is the caller's (Python)
    deadline, NOT the group's: see RCCL_GROUP_TIMEOUT_S."""
    import datetime
    import torch.distributed as dist
    # read by ProcessGroupNCCL's constructor: a timed-out or failed collective is reported to the caller, the process lives
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")      # no heartbeat monitor killing a process whose probe thread is stuck
    group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(RCCL_GROUP_TIMEOUT_S, 4.0 * float(timeout_s))))
    t = torch.ones(1, device=device)
    dist.all_reduce(t, group=group)               # the first collective is where a broken fabric shows
    torch.cuda.synchronize(device)
    if int(t.item()) != dist.get_world_size():
        raise RuntimeError("RCCL all-reduce returned %r for %d ranks" % (t.item(), dist.get_world_size()))
    return group


class JobGroup:
    """What bench.py (or an in-memory consumer) holds for a world > 1 job: `barrier()`, `max(value)`, `backend`
    ("nccl" = RCCL or "gloo"), `close()`."""

    def __init__(self, dist, rccl_group, device, note=None, stuck=False):
        self.dist, self.rccl, self.device, self.note, self.stuck = dist, rccl_group, device, note, stuck
        self.backend = "nccl" if rccl_group is not None else "gloo"
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    # the surface bench.py used before (a torch.distributed module): barrier / get_backend / get_rank / get_world_size
    def get_backend(self):
        return self.backend

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def barrier(self):
        if self.rccl is not None:
            t = torch.zeros(1, device=self.device)
            self.dist.all_reduce(t, group=self.rccl)       # over xGMI; the caller synchronises the device
            torch.cuda.synchronize(self.device)
        else:
            self.dist.barrier()

    def max(self, value):
        if self.rccl is not None:
            t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.rccl)
        else:
            t = torch.tensor([float(value)], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        """Leave the job.  With a probe thread still stuck inside RCCL the group cannot be torn down (the destructor
        would wait for it): the caller should flush its output and leave with os._exit — `stuck` says so."""
        if self.stuck:
            return
        try:
            self.dist.barrier()
            self.dist.destroy_process_group()
        except Exception:
            pass

    def destroy_process_group(self):
        self.close()


def init_process_group_with_fallback(backend="auto", device=None, rank_devices=None, probe_timeout_s=None):
    """-> (JobGroup or None, name of the backend in use).  `gloo`: the control group only.  `nccl`: RCCL or raise.
    `auto`: RCCL when every rank's probe completes inside the deadline, else gloo — decided jointly (module docstring)."""
    rank, _, world = rank_world()
    if world == 1:
        return None, None
    if probe_timeout_s is None:
        probe_timeout_s = float(os.environ.get("IRN_RCCL_PROBE_TIMEOUT_S", "60"))
    # IRN_RCCL_ALLOW_SHARED=1: attempt RCCL even with two ranks on one device (diagnosis only: tools/rccl_shared_probe.sh)
    shared = devices_shared(rank_devices) and os.environ.get("IRN_RCCL_ALLOW_SHARED", "0") != "1"
    if backend == "nccl" and shared:
        raise RuntimeError("backend nccl with two ranks on one device (rank_devices): RCCL needs one device per rank")
    dist = init_process_group("gloo", None, timeout_s=max(CONTROL_TIMEOUT_S, 4 * probe_timeout_s))
    want_rccl = backend in ("auto", "nccl") and torch.cuda.is_available() and device is not None and not shared
    if not want_rccl:
        note = "two ranks share a device: RCCL not attempted" if (shared and backend == "auto") else None
        return JobGroup(dist, None, device, note), "gloo"

    box = {}

    def probe():
        try:
            box["group"] = _rccl_probe(device, probe_timeout_s)
        except BaseException as e:          # noqa: BLE001 — whatever RCCL raises, the job goes on over gloo
            box["error"] = repr(e)[:300]

    th = threading.Thread(target=probe, name="irn-rccl-probe", daemon=True)
    th.start()
    th.join(probe_timeout_s + 5.0)
    stuck = th.is_alive()
    ok = (not stuck) and box.get("group") is not None
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)               # control group: every rank learns about every other
    all_ok = bool(flag.item())
    if all_ok:
        return JobGroup(dist, box["group"], device), "nccl"
    why = "probe still running after %.0f s" % probe_timeout_s if stuck else box.get("error", "another rank's RCCL start-up failed")
    if backend == "nccl":
        raise RuntimeError("RCCL start-up failed on rank %d: %s" % (rank, why))
    print("irn_amd.parallel[rank %d]: RCCL not usable (%s); barrier and timing go over gloo" % (rank, why), file=sys.stderr)
    return JobGroup(dist, None, device, note=why, stuck=stuck), "gloo"


def max_over_ranks(value, group, device="cpu"):
    """Wall time of the slowest rank (the bench contract: barrier, time, MAX over ranks)."""
    if group is None:
        return float(value)
    if isinstance(group, JobGroup):
        return group.max(value)
    if group.get_backend() == "gloo":      # a bare torch.distributed module (single-backend jobs, CPU tests)
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    group.all_reduce(t, op=group.ReduceOp.MAX)
    return float(t.item())


def gather_label_maps(labels, group, dst=0, chunk=32):
    """Optional: collect finished uint8 label maps on rank `dst` for in-memory consumers.
    `labels`: this rank's list of [H,W] uint8 tensors, shapes free (VOC images are ragged).  Direct fan-in, not a
    ring or tree collective: every rank sends straight to `dst` (one xGMI link per peer; 7 links into rank 0 of an
    8-GPU node work at once), `dst` posts all its receives before waiting for any.  Per rank: one int64 header
    (count + shapes) and the maps packed into messages of at least `chunk` maps each — a 512^2 map is 256 KB, so a
    message is >= 8 MB and the transfer is bandwidth-, not latency-bound.  -> on `dst` a list over ranks of lists of
    tensors (on the device the maps came from), elsewhere None."""
    if group is None:
        return [list(labels)]
    dist = group.dist if isinstance(group, JobGroup) else group
    sub = group.rccl if isinstance(group, JobGroup) else None
    on_device = sub is not None or (dist.get_backend() == "nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    # over RCCL every tensor of the exchange lives on THIS rank's device — also on a rank whose shard is empty or whose maps
    # are on the host (a CPU tensor through the RCCL group errors on that rank and hangs the others)
    if on_device:
        dev = group.device if isinstance(group, JobGroup) and group.device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(dev)
    else:
        dev = torch.device("cpu")
    kw = {"group": sub} if sub is not None else {}

    def pack(ts):
        return torch.cat([t.reshape(-1).to(dev) for t in ts]) if ts else torch.empty(0, dtype=torch.uint8, device=dev)

    # phase 1: every rank tells dst how many maps of which shapes follow
    n_max = torch.tensor([len(labels)], dtype=torch.int64, device=dev)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX, **kw)          # tiny; sizes the headers
    n_max = int(n_max.item())
    head = torch.zeros(1 + 2 * n_max, dtype=torch.int64, device=dev)
    head[0] = len(labels)
    for i, t in enumerate(labels):
        head[1 + 2 * i], head[2 + 2 * i] = t.shape[0], t.shape[1]
    if rank != dst:
        dist.send(head, dst, **kw)
        for c0 in range(0, len(labels), chunk):
            dist.send(pack([t.to(torch.uint8) for t in labels[c0:c0 + chunk]]), dst, **kw)
        return None
    peers = [r for r in range(world) if r != dst]
    heads = {r: torch.empty_like(head) for r in peers}
    for w in [dist.irecv(heads[r], r, **kw) for r in peers]:
        w.wait()
    shapes = {r: [(int(heads[r][1 + 2 * i]), int(heads[r][2 + 2 * i])) for i in range(int(heads[r][0]))] for r in peers}
    # phase 2: all receives of all peers posted, then waited for — the peers' links are busy at the same time
    bufs, works = {}, []
    for r in peers:
        bufs[r] = []
        for c0 in range(0, len(shapes[r]), chunk):
            n = sum(h * w for h, w in shapes[r][c0:c0 + chunk])
            b = torch.empty(n, dtype=torch.uint8, device=dev)
            bufs[r].append(b)
            works.append(dist.irecv(b, r, **kw))
    for w in works:
        w.wait()
    out = []
    for r in range(world):
        if r == dst:
            out.append(list(labels))
            continue
        maps, ci = [], 0
        for c0 in range(0, len(shapes[r]), chunk):
            off = 0
            for h, w in shapes[r][c0:c0 + chunk]:
                maps.append(bufs[r][ci][off:off + h * w].view(h, w))
                off += h * w
            ci += 1
        out.append(maps)
    return out
