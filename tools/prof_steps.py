"""Per-step and per-job time stamps of the resident walk for 1, 2, 3 channel jobs (PROF instantiation)."""
import os
import subprocess
import sys
out = sys.argv[1] if len(sys.argv) > 1 else "."
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for args in (["10", "8", "1"], ["10", "8", "2"], ["10", "8", "3"], ["10", "8", "1", "accel=0"], ["5", "32", "1"], ["5", "32", "2"]):
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "resident_profile.py")] + args, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
