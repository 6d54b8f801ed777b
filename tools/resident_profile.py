"""Per-sweep latency breakdown of the resident walk (option profile=1)."""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd import synth
from irn_amd.misc import indexing

dev = torch.device("cuda", 0)
r, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 10, 128, 128
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
edges = [torch.from_numpy(synth.edge_field(h, w, seed=i)).to(dev) for i in range(nimg)]
cams = [torch.from_numpy(synth.cam_blobs(cch, h, w, seed=i)).to(dev) for i in range(nimg)]
wk = indexing.RandomWalk(r, dev)
wk.set_option("variant", 2)
wk.set_option("profile", 1)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    wk.set_option(k, int(v))
for _ in range(2):
    wk(edges, cams, beta=10, n_sweeps=256)
wk.check()
n_steps = wk.steps(256) * cch
p = wk.read_profile().astype(np.float64) * 0.01      # us
for g in range(2):
    q = p[g, 8:min(n_steps - 2, 246)]
    tot = np.diff(q[:, 0])
    print(" ".join(sys.argv[1:]), "wg %d: step period %.2f us (min %.2f max %.2f); poll+stage %.2f, first partials %.2f, reduce+store %.2f" %
          (g, tot.mean(), tot.min(), tot.max(), (q[:, 1] - q[:, 0]).mean(), (q[:, 2] - q[:, 1]).mean(), (q[:, 3] - q[:, 2]).mean()))

# per-job prologue of rounds 0..: rows 248-255 = {job begin, weights + degree part in registers, 1/deg ready, first step}
for g in range(2):
    j = p[g, 248:248 + max(1, min(8, (nimg * (4 if r == 10 else 1) + 3) // 4 if r == 10 else 8))]
    j = j[j[:, 0] > 0]
    if len(j):
        print(" ".join(sys.argv[1:]), "wg %d prologue per job (us): weights+degree %.1f, 1/deg %.1f, tables+fill %.1f; job period %s" %
              (g, (j[:, 1] - j[:, 0]).mean(), (j[:, 2] - j[:, 1]).mean(), (j[:, 3] - j[:, 2]).mean(),
               np.round(np.diff(j[:, 0]), 1).tolist()))

if os.environ.get("RAW"):
    q = p[0, 100:124]
    base = q[0, 0]
    for row in q:
        print("  start %7.2f  staged +%5.2f  partials +%5.2f  stored +%5.2f" % (row[0] - base, row[1] - row[0], row[2] - row[1], row[3] - row[2]))
