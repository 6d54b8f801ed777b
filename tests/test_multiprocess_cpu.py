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


def _ragged_gather_rank(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from irn_amd import parallel
    grp, used = parallel.init_process_group_with_fallback("gloo", None)
    assert used == "gloo" and grp.get_world_size() == world
    # VOC label maps are ragged: rank r holds r + 2 maps of its own shapes (rank 2: 70 maps = three messages of <= 32)
    n = 70 if rank == 2 else rank + 2
    maps = [torch.full((3 + rank + (j % 4), 5 + j % 3), (7 * rank + j) % 251, dtype=torch.uint8) for j in range(n)]
    got = parallel.gather_label_maps(maps, grp, dst=0)
    if rank == 0:
        assert len(got) == world
        for r in range(world):
            nr = 70 if r == 2 else r + 2
            assert len(got[r]) == nr
            for j, t in enumerate(got[r]):
                assert tuple(t.shape) == (3 + r + (j % 4), 5 + j % 3) and int(t.min()) == int(t.max()) == (7 * r + j) % 251
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert got is None
    grp.close()


def test_gather_label_maps_direct_fan_in_ragged(tmp_path):
    """Finished label maps go straight to rank 0 (send / recv per peer, all receives posted before any is waited for),
    ragged shapes, batched >= 32 maps per message."""
    port = _free_port()
    mp.spawn(_ragged_gather_rank, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert os.path.exists(tmp_path / "ok")


def _fallback_rank(rank, world, port, out_dir, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from irn_amd import parallel

    def probe(device, timeout_s):
        # what RCCL can do to a job: throw on ONE rank only, or never come back on one rank
        if mode == "raise" and rank == 1:
            raise RuntimeError("hipIpcGetMemHandle: invalid argument (simulated RCCL start-up failure)")
        if mode == "hang" and rank == 0:
            time.sleep(3600)
        return "pretend-rccl-group"

    parallel._rccl_probe = probe
    torch.cuda.is_available = lambda: True             # `auto` then asks for RCCL first, as on a GPU box
    t0 = time.perf_counter()
    grp, used = parallel.init_process_group_with_fallback("auto", "cuda:0", probe_timeout_s=2.0)
    took = time.perf_counter() - t0
    assert used == "gloo" and grp.backend == "gloo" and grp.get_world_size() == world
    assert os.environ["MASTER_PORT"] == str(port)      # the same rendezvous: nothing listens on port + 1 under an agent store
    assert took < 30.0
    grp.barrier()
    slowest = parallel.max_over_ranks(3.0 - rank, grp, device="cuda:0")     # a GPU device name must not reach gloo
    np.save(os.path.join(out_dir, "f%d.npy" % rank), {"slowest": slowest, "backend": used, "stuck": grp.stuck})
    grp.close()
    if grp.stuck:
        os._exit(0)                                    # what bench.py does with a probe thread that never returned


def test_bench_process_group_falls_back_when_one_rank_fails(tmp_path):
    """bench.py --gpus N needs the group only for its barrier and max-over-ranks: when RCCL fails on ONE rank, every rank
    ends up on the gloo control group (the outcome is agreed over it), on the launcher's own rendezvous."""
    port = _free_port()
    mp.spawn(_fallback_rank, args=(2, port, str(tmp_path), "raise"), nprocs=2, join=True)
    for r in range(2):
        d = np.load(tmp_path / ("f%d.npy" % r), allow_pickle=True).item()
        assert d == {"slowest": 3.0, "backend": "gloo", "stuck": False}


def test_bench_process_group_falls_back_when_one_rank_hangs(tmp_path):
    """An RCCL start-up that never returns on one rank (round 3's 500 s two-rank run) costs the probe's deadline, not the
    line: the stuck rank reports it, the other learns of it over the control group, both time the run over gloo."""
    port = _free_port()
    mp.spawn(_fallback_rank, args=(2, port, str(tmp_path), "hang"), nprocs=2, join=True)
    d0 = np.load(tmp_path / "f0.npy", allow_pickle=True).item()
    d1 = np.load(tmp_path / "f1.npy", allow_pickle=True).item()
    assert d0 == {"slowest": 3.0, "backend": "gloo", "stuck": True} and d1 == {"slowest": 3.0, "backend": "gloo", "stuck": False}


_TORCHRUN_SCRIPT = """
import os, sys, json
sys.path.insert(0, %r)
import torch
from irn_amd import parallel
def probe(device, timeout_s):
    raise RuntimeError('simulated RCCL failure')
parallel._rccl_probe = probe
torch.cuda.is_available = lambda: True
grp, used = parallel.init_process_group_with_fallback('auto', 'cuda:0', probe_timeout_s=2.0)
grp.barrier()
m = parallel.max_over_ranks(float(grp.get_rank()), grp)
if grp.get_rank() == 0:
    print(json.dumps({'backend': used, 'world': grp.get_world_size(), 'max': m,
                      'agent_store': os.environ.get('TORCHELASTIC_USE_AGENT_STORE')}), flush=True)
grp.close()
"""


def test_fallback_under_torch_distributed_run(tmp_path):
    """The documented launcher: under `python -m torch.distributed.run` every rank is a CLIENT of the agent's store
    (TORCHELASTIC_USE_AGENT_STORE), so a fall-back must stay on that rendezvous."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fb.py"
    script.write_text(_TORCHRUN_SCRIPT % root)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["backend"] == "gloo" and d["world"] == 2 and d["max"] == 1.0 and d["agent_store"] == "True"


def test_bench_gpus_flag_is_checked_and_self_launches(tmp_path):
    """`--gpus N` is what the line's n_gpus must mean: a launcher that started another world is an error, and without a
    launcher bench.py starts its N ranks itself (here they stop at once: no GPU in the build container)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-legs"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
    if torch.cuda.is_available():
        return                                   # the real two-rank run is a -m gpu test (tests/test_gpu_bench_ranks.py)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-legs", "--launch-timeout-s", "240"],
                         env=env, capture_output=True, text=True, timeout=300)
    # both ranks were started by torch.distributed.run (its failure report names them) and refused to run without a GPU
    assert out.returncode != 0 and out.returncode != 124
    assert "needs a GPU" in out.stderr and "local_rank: 1" in out.stderr.replace("local_rank  : 1", "local_rank: 1")


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


def _sleepy_work(process_id, model, shards, args):
    import time
    if process_id == 1:
        time.sleep(3600)                 # alive but silent: a wedged GPU / RCCL call, a launch stuck behind another tenant


def test_worker_pool_deadline_and_unpicklable_commands(tmp_path):
    """A worker that hangs without dying is bounded by the step's time-out (the pool is stopped, the error names the silent
    rank and says the command had reached it); a command that cannot be pickled raises in the CALLER, like the reference's
    spawn() does, instead of leaving the parent waiting for workers that never got it."""
    import time
    from irn_amd.step import _common
    pool = _common.WorkerPool([-1, -1])
    try:
        t0 = time.monotonic()
        try:
            pool.run(_sleepy_work, None, torchutils.split_dataset(list(range(4)), 2), {}, timeout_s=3.0)
            raise AssertionError("a silent worker must raise once the deadline has passed")
        except RuntimeError as e:
            assert "worker(s) [1]" in str(e) and "did not answer within 3 s" in str(e) and "had reached [1]" in str(e), str(e)
        assert time.monotonic() - t0 < 30.0 and not pool.alive()
    finally:
        pool.close(force=True)
    pool = _common.WorkerPool([-1])
    try:
        try:
            pool.run(_fake_work, None, torchutils.split_dataset([1], 1), {"out": str(tmp_path), "bad": (lambda: 0)})
            raise AssertionError("an unpicklable argument must raise in the caller")
        except Exception as e:
            assert "pickle" in repr(e).lower() or "lambda" in repr(e).lower(), repr(e)
        pool.run(_fake_work, None, torchutils.split_dataset([1], 1), {"out": str(tmp_path)})       # the pool is still usable
        assert os.path.exists(tmp_path / "00001.npy")
    finally:
        pool.close()
    assert _common.step_timeout(None) == 0.0
    os.environ["IRN_STEP_TIMEOUT_S"] = "7.5"
    try:
        import argparse
        assert _common.step_timeout(argparse.Namespace()) == 7.5 and _common.step_timeout(argparse.Namespace(step_timeout=2)) == 2.0
    finally:
        del os.environ["IRN_STEP_TIMEOUT_S"]
