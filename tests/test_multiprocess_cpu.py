"""N > 1 paths on CPU: (1) the steps' one-process-per-shard spawn writes every output exactly once;
(2) the bench's barrier / max-over-ranks timing and the optional label gather over a world_size-2
gloo group."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from irn_amd.misc import torchutils


def _fake_work(process_id, model, shards, args):
    # stands in for step.*._work: same signature, same per-item file output, no GPU
    for item in shards[process_id]:
        np.save(os.path.join(args["out"], "%05d.npy" % item), {"rank": process_id, "item": item})


def test_spawn_workers_strided_shards(tmp_path):
    from irn_amd.step import _common
    items = list(range(23))
    shards = torchutils.split_dataset(items, 3)
    _common.spawn_workers(_fake_work, None, shards, {"out": str(tmp_path)})
    files = sorted(os.listdir(tmp_path))
    assert files == ["%05d.npy" % i for i in items]
    for i in items:
        d = np.load(os.path.join(tmp_path, "%05d.npy" % i), allow_pickle=True).item()
        assert d["rank"] == i % 3


_CALLS = []          # lives in a pool worker: grows from one step to the next if (and only if) the process persists


def _stateful_work(process_id, model, shards, args):
    _CALLS.append(args["tag"])
    for item in shards[process_id]:
        np.save(os.path.join(args["out"], "%s_%05d.npy" % (args["tag"], item)),
                {"rank": process_id, "pid": os.getpid(), "calls": list(_CALLS), "model": None if model is None else float(model.sum())})


def _failing_work(process_id, model, shards, args):
    if process_id == 1:
        raise ValueError("shard %d cannot be processed" % process_id)


def test_worker_pool_persists_across_steps_and_fails_fast(tmp_path):
    """The per-GPU workers are spawned once and serve every step (the reference re-spawns per step,
    step/make_cam.py:74): same PIDs and surviving module state in the second step, the model travels by pickling, a
    failing shard surfaces in the parent like spawn(join=True) would raise, and a new pool works afterwards."""
    from irn_amd.step import _common
    pool = _common.WorkerPool([-1, -1, -1])                      # three workers without a GPU
    try:
        shards = torchutils.split_dataset(list(range(10)), 3)
        pool.run(_stateful_work, torch.ones(4), shards, {"out": str(tmp_path), "tag": "a"})
        pool.run(_stateful_work, None, shards, {"out": str(tmp_path), "tag": "b"})
        for i in range(10):
            a = np.load(tmp_path / ("a_%05d.npy" % i), allow_pickle=True).item()
            b = np.load(tmp_path / ("b_%05d.npy" % i), allow_pickle=True).item()
            assert a["rank"] == b["rank"] == i % 3 and a["pid"] == b["pid"] != os.getpid()
            assert a["calls"] == ["a"] and b["calls"] == ["a", "b"] and a["model"] == 4.0 and b["model"] is None
        assert len({np.load(tmp_path / ("a_%05d.npy" % i), allow_pickle=True).item()["pid"] for i in range(3)}) == 3
        try:
            pool.run(_failing_work, None, shards, {})
            raise AssertionError("a failing shard must raise in the parent")
        except RuntimeError as e:
            assert "shard 1 cannot be processed" in str(e) and "ValueError" in str(e)
        assert not pool.alive()
    finally:
        pool.close(force=True)
    pool = _common.WorkerPool([-1])
    try:
        pool.run(_stateful_work, None, torchutils.split_dataset([7], 1), {"out": str(tmp_path), "tag": "c"})
        assert os.path.exists(tmp_path / "c_00007.npy")
    finally:
        pool.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_rank(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from irn_amd import parallel
    dist = parallel.init_process_group(backend="gloo")
    assert dist is not None and dist.get_world_size() == world
    mine = torchutils.shard_indices(11, rank, world)
    dist.barrier()
    slowest = parallel.max_over_ranks(1.0 + rank, dist)
    labels = [torch.full((4, 5), 10 * rank + j, dtype=torch.uint8) for j in range(2)]
    gathered = parallel.gather_label_maps(labels, dist, dst=0)
    np.save(os.path.join(out_dir, "r%d.npy" % rank),
            {"mine": mine, "slowest": slowest, "gathered": None if gathered is None else
             [[int(t[0, 0]) for t in per_rank] for per_rank in gathered]})
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_timing_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_gloo_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy", allow_pickle=True).item()
    r1 = np.load(tmp_path / "r1.npy", allow_pickle=True).item()
    assert sorted(np.concatenate([r0["mine"], r1["mine"]]).tolist()) == list(range(11))
    assert not set(r0["mine"]) & set(r1["mine"])
    assert r0["slowest"] == r1["slowest"] == 2.0
    assert r0["gathered"] == [[0, 1], [10, 11]] and r1["gathered"] is None


def _fallback_rank(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from irn_amd import parallel
    real = parallel.init_process_group

    def broken_rccl(backend=None, device=None, timeout_s=None):
        if backend == "nccl":
            raise RuntimeError("hipIpcGetMemHandle: invalid argument (simulated RCCL start-up failure)")
        return real(backend, device, timeout_s)

    parallel.init_process_group = broken_rccl
    torch.cuda.is_available = lambda: True             # `auto` then asks for RCCL first, as on a GPU box
    dist, used = parallel.init_process_group_with_fallback("auto", None)
    assert used == "gloo" and dist.get_world_size() == world and os.environ["MASTER_PORT"] == str(port + 1)
    dist.barrier()
    slowest = parallel.max_over_ranks(3.0 - rank, dist, device="cuda:0")     # a GPU device name must not reach gloo
    np.save(os.path.join(out_dir, "f%d.npy" % rank), {"slowest": slowest, "backend": used})
    dist.barrier()
    dist.destroy_process_group()


def test_bench_process_group_falls_back_to_gloo(tmp_path):
    """bench.py --gpus N needs the group only for its barrier and max-over-ranks: when RCCL cannot start, every rank
    joins a gloo group on the next port instead and the line is still produced (with the backend recorded)."""
    port = _free_port()
    mp.spawn(_fallback_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        d = np.load(tmp_path / ("f%d.npy" % r), allow_pickle=True).item()
        assert d == {"slowest": 3.0, "backend": "gloo"}


def _dying_work(process_id, model, shards, args):
    if process_id == 0:
        os._exit(7)                      # a worker that disappears (segfault, OOM kill) instead of raising


def test_worker_pool_reports_a_worker_that_dies(tmp_path):
    from irn_amd.step import _common
    pool = _common.WorkerPool([-1, -1])
    try:
        try:
            pool.run(_dying_work, None, torchutils.split_dataset(list(range(4)), 2), {})
            raise AssertionError("a dead worker must raise in the parent")
        except RuntimeError as e:
            assert "died without reporting" in str(e) and "7" in str(e)
    finally:
        pool.close(force=True)
