#!/usr/bin/env python3
"""How fast is an image size the shipped database does NOT know (the long tail of a real dataset), in the default reproducible mode?
CAM passes (four scales) of 8 pairs of one untuned size:
  product        rows travel one pair per pass, NCHW under MIOpen's deterministic attribute (net/resnet50.run_rows)
  nchw16         the same trunk with all 16 rows in one pass (round 5's behaviour: not layout-independent)
  cl_split       channels-last forced, split-precision GEMMs, MIOpen's untuned picks for what is left, attribute off (fast mode)
  cl_split_pair  the same one pair per pass
Reference: step/make_cam.py:26-56."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from irn_amd.net import resnet50 as r50, resnet50_cam, weights  # noqa: E402
from irn_amd.step import _common  # noqa: E402

dev = torch.device("cuda", 0)
_common.miopen_setup(0)
cam = resnet50_cam.CAM()
cam.load_state_dict(weights.random_cam_state(1))
cam = cam.to(dev).eval()
real_rows = r50.pass_rows
for size in (sys.argv[1] if len(sys.argv) > 1 else "366x500,300x400").split(","):
    h, w = (int(v) for v in size.split("x"))
    xs = [torch.randn(16, 3, int(round(h * s)), int(round(w * s)), device=dev) for s in (1.0, 0.5, 1.5, 2.0)]
    for tag, layout, det, rows in (("product", "auto", True, None), ("nchw16", "auto", True, 16), ("cl_split", "1", False, 16), ("cl_split_pair", "1", False, 2)):
        r50.CHANNELS_LAST_MODE, r50.DETERMINISTIC = layout, det
        torch.backends.cudnn.deterministic = det
        r50.pass_rows = real_rows if rows is None else (lambda x, rows=rows: rows)
        with torch.no_grad():
            outs = [cam.forward_batch(x) for x in xs]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                outs = [cam.forward_batch(x) for x in xs]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        print("%-9s %-14s %8.2f ms per 8 images = %6.1f images/s" % (size, tag, 1e3 * dt, 8 / dt), flush=True)
