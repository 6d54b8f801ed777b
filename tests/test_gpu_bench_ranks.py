"""bench.py as a multi-rank job on the one GPU of the box: `--gpus 2` starts its own two ranks (torch.distributed.run,
one process each), `--rank-devices 0,0` puts both on device 0, the process group serves the barrier and max-over-ranks.
What an 8-GPU node adds to this is only RCCL's own start-up, which `--backend auto` probes under a deadline
(irn_amd/parallel.py; its failure modes are CPU-tested in tests/test_multiprocess_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    return env


def _run(extra, timeout=400):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", "--batch", "16", "--steps", "2",
           "--warmup", "1", "--launch-timeout-s", "300"] + extra
    out = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


@pytest.mark.parametrize("backend", ["gloo", "auto"])
def test_two_ranks_on_one_device(backend):
    out, lines = _run(["--gpus", "2", "--rank-devices", "0,0", "--backend", backend])
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    pg = d["config"]["process_group"]
    print("two ranks on device 0, --backend %s: %.0f images/s, %.2f ms per step, process group %s" %
          (backend, d["value"], d["ms_per_step"], pg))
    assert d["n_gpus"] == 2 and pg["ranks"] == 2 and pg["backend"] == "gloo"      # RCCL wants one device per rank
    assert d["value"] > 0 and d["scaling"] == "weak" and d["config"]["images_per_gpu_per_step"] == 16
    assert d["roofline"] is not None and d["roofline"]["achieved"] > 0


def test_one_rank_under_the_drivers_launcher():
    """The driver's N > 1 command line at N = 1: python -m torch.distributed.run ... bench.py --gpus 1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-legs", "--no-cpu-baseline", "--batch", "16",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_gpus_must_match_the_launched_world():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-legs", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


@pytest.mark.parametrize("workload,extra", [("e2e", ["--batch", "2", "--steps", "2", "--warmup", "1"]),
                                            ("steps", ["--batch", "8", "--steps", "1", "--warmup", "1"])])
def test_backbone_workloads_as_two_ranks_on_one_device(workload, extra):
    """The stage that bounds the metric under `--gpus N`: `--workload e2e` (CAM + IRNet + walk +
    labels) and `--workload steps` (the run_sample.py step API on JPEG files) as two ranks, each with its own images,
    temporary directory and MIOpen user database, barrier + max-over-ranks around the timed region.  Both ranks share the one
    GPU of this box, so their persistent walks may lose the bounded wait to each other: `--allow-walk-fallback` records the
    re-runs instead of failing the line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", "--launch-timeout-s", "500", "--workload", workload,
           "--gpus", "2", "--rank-devices", "0,0", "--backend", "gloo", "--allow-walk-fallback"] + extra
    out = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    print("%s as two ranks on device 0: %.1f images/s whole job, %.1f ms per step, walk re-runs %s" %
          (workload, d["value"], d["ms_per_step"], d["config"].get("walk_fallback_runs")))
    assert d["n_gpus"] == 2 and d["config"]["process_group"]["ranks"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["images_per_gpu_per_step"] == int(extra[1])


def _rank_with_a_hung_rccl_probe(rank, port, q):
    """One rank of a two-rank job whose RCCL start-up never completes: rank 1's probe sleeps for ever, so rank 0's REAL
    `new_group(backend='nccl')` + all-reduce blocks inside RCCL waiting for it."""
    import time
    import torch
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "IRN_RCCL_ALLOW_SHARED": "1", "IRN_RCCL_PROBE_TIMEOUT_S": "20"})
    from irn_amd import parallel
    torch.cuda.set_device(0)
    if rank == 1:
        parallel._rccl_probe = lambda device, timeout_s: time.sleep(3600)
    t0 = time.time()
    group, backend = parallel.init_process_group_with_fallback("auto", torch.device("cuda", 0), rank_devices="0,0")
    dt = time.time() - t0
    group.barrier()
    mx = group.max(float(rank + 1))
    q.put((rank, backend, dt, mx, bool(group.stuck)))
    q.close()
    q.join_thread()      # the result is on its way before the process leaves without interpreter shutdown ...
    os._exit(0)          # ... which is what bench.py does with a stuck probe thread


def test_rccl_start_up_that_hangs_ends_on_gloo_inside_the_deadline():
    """the Python deadline around the probe must be the only one that decides — the RCCL group's own timeout
    (watchdog) is set far beyond it and told not to kill the process.  Here one rank never enters RCCL; the other one's real
    start-up hangs; both must agree on gloo shortly after the 20 s deadline and the control group must still work."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_with_a_hung_rccl_probe, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    print("hung RCCL start-up: %s" % got)
    for rank, backend, dt, mx, stuck in got:
        assert backend == "gloo" and dt < 60.0 and mx == 2.0
    assert all(p.exitcode == 0 for p in ps)
