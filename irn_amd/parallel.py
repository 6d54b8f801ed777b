"""Multi-GPU plumbing of the hot path: one process per GPU, images strided over ranks, no
collective on the data path (reference misc/torchutils.py:66-68 + multiprocessing.spawn at
step/make_cam.py:74).  The only collectives are the benchmark's barrier / max-over-ranks timing and
an optional gather of finished label maps to rank 0 for in-memory consumers."""
import os

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None, device=None, timeout_s=None):
    """Joins the job described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).  backend 'nccl'
    is RCCL on ROCm; 'gloo' serves the CPU tests."""
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


def init_process_group_with_fallback(backend="auto", device=None):
    """-> (dist or None, name of the backend in use).  `auto` tries RCCL ("nccl") and, when its start-up or first
    collective fails (missing IPC support, a bad fabric), joins the same rendezvous over gloo instead: the hot path
    has no collective, the group only serves a benchmark's barrier and max-over-ranks."""
    rank, _, world = rank_world()
    if world == 1:
        return None, None
    import torch.distributed as dist
    want = ("nccl" if torch.cuda.is_available() else "gloo") if backend == "auto" else backend
    if want == "nccl":
        try:
            d = init_process_group("nccl", device, timeout_s=600)     # a hanging start-up must end in the fall-back too
            t = torch.zeros(1, device=device)
            d.all_reduce(t)                       # the first collective is where a broken fabric shows
            torch.cuda.synchronize()
            return d, "nccl"
        except Exception as e:                    # every rank sees the failure of the collective start-up
            if backend != "auto":
                raise
            import sys
            print("irn_amd.parallel: RCCL start-up failed (%s); using gloo for the barrier" % (repr(e)[:200],), file=sys.stderr)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            # a fresh rendezvous on the next port: the store of the failed group may be half torn down
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            want = "gloo"
    return init_process_group(want, device), want


def max_over_ranks(value, dist, device="cpu"):
    """Wall time of the slowest rank (the bench contract: barrier, time, MAX over ranks)."""
    if dist is None:
        return float(value)
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_label_maps(labels, dist, dst=0):
    """Optional: collect finished uint8 label maps on rank `dst` (list of [H,W] tensors per rank,
    equal shapes).  One batched message per rank — the maps are small (256 KB at 512^2), so this is
    latency-bound and sits off the critical path."""
    if dist is None:
        return [labels]
    stacked = torch.stack(labels)
    world = dist.get_world_size()
    bufs = [torch.empty_like(stacked) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(stacked, bufs, dst=dst)
    return None if bufs is None else [list(b) for b in bufs]
