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
