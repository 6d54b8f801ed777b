#!/usr/bin/env python3
"""CAM backbone on MIOpen: which cheap, parity-preserving changes pay?  (baseline / channels_last /
BN folded into the convolutions / both), ms per image at B=8 per scale + max-abs deviation of the CAM."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch, torch.nn as nn, torch.nn.functional as F
from irn_amd.net import resnet50_cam, weights
dev = torch.device("cuda", 0)
base = resnet50_cam.CAM(); base.load_state_dict(weights.random_cam_state(1)); base = base.to(dev).eval()

def fold(net):
    net = copy.deepcopy(net)
    def fold_pair(conv, bn):
        g = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        new = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, bias=True).to(dev)
        new.weight.data = conv.weight.data * g[:, None, None, None]
        new.bias.data = bn.bias.data - bn.running_mean * g
        return new
    t = net.resnet50
    t.conv1 = fold_pair(t.conv1, t.bn1); t.bn1 = nn.Identity()
    for li in range(1, 5):
        for blk in getattr(t, "layer%d" % li):
            blk.conv1 = fold_pair(blk.conv1, blk.bn1); blk.bn1 = nn.Identity()
            blk.conv2 = fold_pair(blk.conv2, blk.bn2); blk.bn2 = nn.Identity()
            blk.conv3 = fold_pair(blk.conv3, blk.bn3); blk.bn3 = nn.Identity()
            if blk.downsample is not None:
                blk.downsample = nn.Sequential(fold_pair(blk.downsample[0], blk.downsample[1]))
    net.stage1 = nn.Sequential(t.conv1, t.bn1, t.relu, t.maxpool, t.layer1)
    net.stage2 = nn.Sequential(t.layer2); net.stage3 = nn.Sequential(t.layer3); net.stage4 = nn.Sequential(t.layer4)
    return net

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n

B = 8
scales = (0.5, 1.0, 1.5, 2.0)
imgs = {s: torch.randn(2 * B, 3, int(512 * s), int(512 * s), device=dev) for s in scales}
variants = {"baseline": (base, False), "channels_last": (copy.deepcopy(base).to(memory_format=torch.channels_last), True),
            "bn_folded": (fold(base), False), "bn_folded+channels_last": (fold(base).to(memory_format=torch.channels_last), True)}
ref = {}
for name, (net, cl) in variants.items():
    tot = 0.0
    line = []
    for s in scales:
        x = imgs[s].contiguous(memory_format=torch.channels_last) if cl else imgs[s]
        def one():
            with torch.no_grad():
                return F.relu(F.conv2d(net.features(x), net.classifier.weight))
        dt = timeit(one) / B * 1e3
        out = one().float()
        if name == "baseline": ref[s] = out
        dev_abs = (out - ref[s]).abs().max().item(); scale_abs = ref[s].abs().max().item()
        tot += dt
        line.append("s%.1f %.2f ms (dev %.1e of %.1e)" % (s, dt, dev_abs, scale_abs))
    print("%-26s total %.2f ms/img | %s" % (name, tot, " | ".join(line)), flush=True)
