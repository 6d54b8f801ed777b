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
