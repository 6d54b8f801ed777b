"""Diagnostic (library built with -DIRN_PROF_COMBINE=1 or 2, IRN_HIP_LIB=irn_amd/lib/libirn_hip_diag<mode>.so): where the
combine phase of a resident-walk step goes.
  mode 1 stamps: [0] step start, [1] combine entry (behind the barrier that follows the partial sums), [2] every LDS read of
                 the combine has landed, [3] its arithmetic is done; the next step's [0] closes the stores.
  mode 2 stamps: [0] step start, [1] arithmetic done, [2] LDS writes + global stores issued, [3] end of the step (behind the
                 barrier of single-channel jobs); the next step's [0] closes the loop top (scalar set-up, coefficient load).
usage: python tools/combine_profile.py <radius> <images> <channels> [mode=1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd import synth
from irn_amd.misc import indexing

dev = torch.device("cuda", 0)
r, nimg, cch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
h = w = 128
edges = [torch.from_numpy(synth.edge_field(h, w, seed=i)).to(dev) for i in range(nimg)]
cams = [torch.from_numpy(synth.cam_blobs(cch, h, w, seed=i)).to(dev) for i in range(nimg)]
wk = indexing.RandomWalk(r, dev)
wk.set_option("profile", 1)
for _ in range(2):
    wk(edges, cams, beta=10, n_sweeps=256)
wk.check()
n_steps = wk.steps(256) * cch
p = wk.read_profile().astype(np.float64) * 0.01      # us
for g in range(2):
    q = p[g, 8:min(n_steps - 2, 246)]
    period = np.diff(q[:, 0])
    if mode == 2:
        print("radius %d C=%d wg %d: step %.2f us = up to the end of the combine's arithmetic %.2f + LDS writes and store issue %.2f + "
              "end-of-step barrier %.2f + loop top (to next step start) %.2f" % (
                  r, cch, g, period.mean(), (q[:-1, 1] - q[:-1, 0]).mean(), (q[:-1, 2] - q[:-1, 1]).mean(),
                  (q[:-1, 3] - q[:-1, 2]).mean(), (q[1:, 0] - q[:-1, 3]).mean()))
        continue
    print("radius %d C=%d wg %d: step %.2f us = up to the combine %.2f + LDS reads %.2f + arithmetic %.2f + writes/stores (to next step start) %.2f" % (
        r, cch, g, period.mean(), (q[:-1, 1] - q[:-1, 0]).mean(), (q[:-1, 2] - q[:-1, 1]).mean(), (q[:-1, 3] - q[:-1, 2]).mean(),
        (q[1:, 0] - q[:-1, 3]).mean()))
