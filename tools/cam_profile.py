#!/usr/bin/env python3
"""Where does the CAM stage spend its time?  backbone per batch size vs the merge of step/make_cam.py:38-52."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIOPEN_FIND_MODE"] = "2"
import torch, torch.nn.functional as F
from irn_amd.net import resnet50_cam, weights
from irn_amd.step import make_cam
dev = torch.device("cuda", 0)
net = resnet50_cam.CAM(); net.load_state_dict(weights.random_cam_state(1)); net = net.to(dev).eval()
scales = (1.0, 0.5, 1.5, 2.0)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for B in (1, 2, 4, 8):
    imgs = {s: torch.randn(B, 3, int(512 * s), int(512 * s), device=dev) for s in scales}
    def backbone():
        with torch.no_grad():
            outs = []
            for s in scales:
                x = imgs[s]; f = F.relu(F.conv2d(net.features(torch.cat([x, x.flip(-1)], 0)), net.classifier.weight))
                outs.append(f[:B] + f[B:].flip(-1))
            return outs
    for s in scales:
        def one(s=s):
            with torch.no_grad():
                x = imgs[s]; return net.features(torch.cat([x, x.flip(-1)], 0))
        print("B=%d scale %.1f backbone %.2f ms/img" % (B, s, 1e3 * t(one) / B), flush=True)
    dt = t(backbone)
    outs = backbone()
    lab = torch.zeros(20, device=dev); lab[[3, 7]] = 1
    def merge():
        with torch.no_grad():
            return [make_cam.merge_scales([o[i] for o in outs], (512, 512), lab) for i in range(B)]
    dm = t(merge)
    print("B=%d backbone %.2f ms/img, merge %.2f ms/img, mem %.1f GB" % (B, 1e3 * dt / B, 1e3 * dm / B, torch.cuda.max_memory_allocated() / 1e9), flush=True)
