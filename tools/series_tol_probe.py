"""Error of the walk against the fp64 oracle (oracle/walk_oracle.c) for the plain 2^8-fold iteration and for the truncated
Chebyshev series at several truncation bounds, on 128x128 grids (radius 10 and 5) — the data behind the default bound."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd import synth
from irn_amd.misc import indexing
from oracle import build_oracle, irn_oracle as O

dev = torch.device("cuda", 0)
lib = build_oracle.load()
for r in (10, 5):
    shapes = [(128, 128, 1), (128, 128, 2), (128, 128, 3), (94, 125, 4), (128, 128, 1), (128, 128, 2)]
    cams = [synth.cam_blobs(c, h, w, seed=2000 + i) for i, (h, w, c) in enumerate(shapes)]
    edges = [synth.edge_field(h, w, seed=2000 + i) for i, (h, w, c) in enumerate(shapes)]
    truth = build_oracle.walk_batch(lib, cams, edges, r, 10, 256)
    ups = [O.sem_seg_epilogue(t, (4 * t.shape[-2], 4 * t.shape[-1]), np.arange(t.shape[0]), 0.25) for t in truth]
    ce = [torch.from_numpy(e).to(dev) for e in edges]
    cc = [torch.from_numpy(c).to(dev) for c in cams]
    for name, opts in (("plain powers (256 applications)", {"accel": 0}), ("series tol 1e-8", {"accel_tol_exp": 8}),
                       ("series tol 1e-7", {"accel_tol_exp": 7}), ("series tol 1e-6", {"accel_tol_exp": 6}),
                       ("series tol 1e-5", {"accel_tol_exp": 5}), ("series tol 1e-4", {"accel_tol_exp": 4})):
        wk = indexing.RandomWalk(r, dev)
        for k, v in opts.items():
            wk.set_option(k, v)
        out = wk(ce, cc, beta=10, exp_times=8)
        wk.check()
        errs = [float(np.abs(o.cpu().numpy() - t).max()) for o, t in zip(out, truth)]
        # label maps through the oracle's epilogue on both walks
        flips = 0
        for o, (up, lab, _) in zip(out, ups):
            _, lab_g, _ = O.sem_seg_epilogue(o.cpu().numpy(), lab.shape, np.arange(o.shape[0]), 0.25)
            flips += int((lab_g != lab).sum())
        print("radius %2d  %-32s applications %3d  max |gpu - fp64| %.2e (mean over images %.2e)  label pixels differing %d of %d" %
              (r, name, wk.steps(256), max(errs), float(np.mean(errs)), flips, sum(u[1].size for u in ups)))
        wk.close()
