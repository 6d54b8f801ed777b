"""Multi-GPU plumbing of the hot path: one process per GPU, images strided over ranks, no
collective on the data path (reference misc/torchutils.py:66-68 + multiprocessing.spawn at
step/make_cam.py:74).  The only collectives are the benchmark's barrier / max-over-ranks timing and
an optional gather of finished label maps to rank 0 for in-memory consumers."""
import os

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None, device=None):
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
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(value, dist, device="cpu"):
    """Wall time of the slowest rank (the bench contract: barrier, time, MAX over ranks)."""
    if dist is None:
        return float(value)
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
