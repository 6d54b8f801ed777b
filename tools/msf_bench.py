#!/usr/bin/env python3
"""Input pipeline of one 4-scale CAM item (voc12/dataloader.py:191-201): PIL/numpy loop on one host core
against irn_msf_pack on the GPU, plus the host-to-device bytes either way.   python tools/msf_bench.py [H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd import ops, synth                                  # noqa: E402
from irn_amd.misc import imutils                                # noqa: E402
from irn_amd.voc12.dataloader import TorchvisionNormalize      # noqa: E402


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
    scales = (1.0, 0.5, 1.5, 2.0)
    img = synth.photo(h, w, seed=1)
    norm = TorchvisionNormalize()

    def host_item():
        ms = []
        for s in scales:
            s_img = img if s == 1 else imutils.pil_rescale(img, s, order=3)
            s_img = imutils.HWC_to_CHW(norm(s_img))
            ms.append(np.stack([s_img, np.flip(s_img, -1)], axis=0))
        return ms

    host_item()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        item = host_item()
    t_host = (time.perf_counter() - t0) / n
    item_bytes = sum(a.nbytes for a in item)
    print("host (PIL bicubic + numpy normalise/flip, 1 core): %.1f ms per image; item = %.1f MB of fp32 to copy to the GPU "
          "(raw image: %.2f MB)" % (1e3 * t_host, item_bytes / 1e6, img.nbytes / 1e6))

    dev = torch.device("cuda", 0)
    u8 = torch.from_numpy(img).to(dev)
    for _ in range(3):
        outs = ops.msf_pack(u8, scales)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 50
    e0.record()
    for _ in range(reps):
        outs = ops.msf_pack(u8, scales)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("irn_msf_pack: %.3f ms per image (%.0f GB/s of fp32 output written); bit-identical: %s"
          % (ms, item_bytes / ms / 1e6, all(np.array_equal(o.cpu().numpy(), a) for o, a in zip(outs, item))))
    pinned = [torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in item]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for p in pinned:
            p.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    print("H2D of the host-built item from pinned memory: %.2f ms per image" % (1e3 * (time.perf_counter() - t0) / 10))


if __name__ == "__main__":
    main()
