"""`steps` leg of bench.py (run_sample.py step API through run(args)) for several loader / writer thread counts."""
import json
import os
import subprocess
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "."
for n in (4, 8, 16, 32):
    js = os.path.join(out, "steps_workers_%d.json" % n)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "steps", "--batch", "128", "--steps", "2", "--warmup", "1",
                        "--loader-workers", str(n), "--no-legs", "--no-cpu-baseline", "--json-out", js], capture_output=True, text=True, timeout=400)
    try:
        d = json.load(open(js))
        print("num_workers %2d: %.1f images/s, passes %s" % (n, d["value"], d["config"].get("last_pass_seconds")))
    except Exception as e:
        print("num_workers %d failed: %r %s" % (n, e, r.stderr[-300:]))
