"""bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU), with both ranks
forced onto GPU 0 of a one-GPU box: RCCL cannot form a group of two ranks on one device, so this exercises the gloo
fall-back of the barrier / max-over-ranks on real hardware, and two ranks' resident walks sharing one GPU."""
import os
import subprocess
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, IRN_BENCH_FORCE_DEVICE="0")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
       "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-legs",
       "--no-cpu-baseline", "--batch", "64"]
r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
print("rc", r.returncode)
print(r.stdout[-3000:])
print(r.stderr[-3000:])
