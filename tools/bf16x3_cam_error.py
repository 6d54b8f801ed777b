"""End-to-end error of the CAM network when its 1x1 convolutions are computed as split-precision products (emulated on
CPU: operands rounded to the narrow type, products exact in fp32, fp32 accumulation), vs fp64 and vs plain fp32.
  bf16x3 / bf16x4   x = hi + lo in bf16 (16 mantissa bits together): round 3, profiles/r03_split_gemm_note.txt
  fp16s             round 6: x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16((x - hi) 2^11) (22 mantissa bits together; the
                    scale keeps lo' out of fp16's subnormals); product = hi.hi + 2^-11 (hi.lo' + lo'.hi): two fp16 GEMMs,
                    the second with alpha = 2^-11, beta = 1.  Operands outside fp16's range are reported (they would need a
                    per-tensor power-of-two scale on top)."""
import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from irn_amd.net import resnet50_cam, weights
from irn_amd import synth
torch.manual_seed(0)
torch.set_num_threads(8)

def split(t):
    hi = t.to(torch.bfloat16).float(); lo = (t - hi).to(torch.bfloat16).float(); return hi, lo

S11 = 2.0 ** 11
OVER = {"n": 0, "max": 0.0}
def split16(t):
    OVER["max"] = max(OVER["max"], t.abs().max().item())
    OVER["n"] += int((t.abs() > 65504).sum())
    hi = t.to(torch.float16).float(); lo = ((t - hi) * S11).to(torch.float16).float(); return hi, lo

class Split1x1(nn.Module):
    def __init__(self, conv, terms): super().__init__(); self.conv, self.terms = conv, terms
    @property
    def weight(self): return self.conv.weight
    def forward(self, x):
        c = self.conv
        kw = dict(stride=c.stride, padding=c.padding)
        if self.terms == 16:
            xh, xl = split16(x); wh, wl = split16(c.weight)
            return F.conv2d(xh, wh, **kw) + (F.conv2d(xl, wh, **kw) + F.conv2d(xh, wl, **kw)) / S11
        xh, xl = split(x); wh, wl = split(c.weight)
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
    for mode in ("fp64", "fp32", "bf16x3", "bf16x4", "fp16s"):
        net = resnet50_cam.CAM(); net.load_state_dict(weights.random_cam_state(1)); net.eval()
        with torch.no_grad():
            if mode == "fp64": y = net.double()(x.double())
            else:
                if mode.startswith("bf16"): patch(net, int(mode[-1]))
                if mode == "fp16s": patch(net, 16)
                y = net(x.float())
        res[mode] = cam_norm(y.double())
    for mode in ("fp32", "bf16x3", "bf16x4", "fp16s"):
        print("%dx%d %-7s: max |normalised CAM - fp64| = %.2e" % (h, w, mode, (res[mode] - res["fp64"]).abs().max().item()))
    print("%dx%d fp16s operands: max |value| %.3g, %d beyond fp16's range" % (h, w, OVER["max"], OVER["n"]))
