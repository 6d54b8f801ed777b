"""End-to-end error of the CAM network when its 1x1 convolutions are computed as bf16x3 split products (emulated on CPU:
operands rounded to bf16 hi/lo, products exact in fp32 accumulate), vs fp64 and vs plain fp32."""
import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from irn_amd.net import resnet50_cam, weights
from irn_amd import synth
torch.manual_seed(0)
torch.set_num_threads(8)

def split(t):
    hi = t.to(torch.bfloat16).float(); lo = (t - hi).to(torch.bfloat16).float(); return hi, lo

class Split1x1(nn.Module):
    def __init__(self, conv, terms): super().__init__(); self.conv, self.terms = conv, terms
    @property
    def weight(self): return self.conv.weight
    def forward(self, x):
        c = self.conv
        xh, xl = split(x); wh, wl = split(c.weight)
        kw = dict(stride=c.stride, padding=c.padding)
        y = F.conv2d(xh, wh, **kw) + F.conv2d(xl, wh, **kw) + F.conv2d(xh, wl, **kw)
        if self.terms == 4: y = y + F.conv2d(xl, wl, **kw)
        return y

def patch(m, terms):
    for name, ch in list(m.named_children()):
        if isinstance(ch, nn.Conv2d) and ch.kernel_size == (1, 1): setattr(m, name, Split1x1(ch, terms))
        elif not isinstance(ch, Split1x1): patch(ch, terms)

def cam_norm(a):   # what make_cam keeps: per-channel max-normalised maps
    return a / (a.amax(dim=(1, 2), keepdim=True) + 1e-5)

for (h, w) in ((128, 160), (256, 256)):
    img, flip = synth.image_pair(h, w, seed=3) if hasattr(synth, "image_pair") else (None, None)
    x = torch.from_numpy(__import__("numpy").stack([img, flip])) if img is not None else torch.randn(2, 3, h, w)
    res = {}
    for mode in ("fp64", "fp32", "bf16x3", "bf16x4"):
        net = resnet50_cam.CAM(); net.load_state_dict(weights.random_cam_state(1)); net.eval()
        with torch.no_grad():
            if mode == "fp64": y = net.double()(x.double())
            else:
                if mode.startswith("bf16"): patch(net, int(mode[-1]))
                y = net(x.float())
        res[mode] = cam_norm(y.double())
    for mode in ("fp32", "bf16x3", "bf16x4"):
        print("%dx%d %-7s: max |normalised CAM - fp64| = %.2e" % (h, w, mode, (res[mode] - res["fp64"]).abs().max().item()))
