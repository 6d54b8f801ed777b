"""IRNet (boundary + displacement heads on a ResNet-50 trunk), inference form, host side on
PyTorch-ROCm.  API mirror of reference net/resnet50_irn.py:7-133 (``Net``) and :216-234
(``EdgeDisplacement``); attribute names reproduce the reference's state-dict keys.

The training wrapper ``AffinityDisplacementLoss`` (net/resnet50_irn.py:144-213) is out of scope
 — its path-max gather is the same operator as
``irn_amd.misc.indexing.edge_to_affinity``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import resnet50 as _r50


def _head(c_in, c_out, groups, up=0, tail=()):
    """1x1 conv -> GroupNorm -> [bilinear x`up`] -> ReLU (-> tail modules)."""
    mods = [nn.Conv2d(c_in, c_out, 1, bias=False), nn.GroupNorm(groups, c_out)]
    if up:
        mods.append(nn.Upsample(scale_factor=up, mode="bilinear", align_corners=False))
    mods.append(nn.ReLU(inplace=True))
    mods.extend(tail)
    return nn.Sequential(*mods)


class MeanShift(nn.Module):
    """Subtracts the displacement running mean at inference (net/resnet50_irn.py:99-108)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(n))

    def forward(self, x):
        return x if self.training else x - self.running_mean.view(1, -1, 1, 1)


class Net(nn.Module):
    MeanShift = MeanShift

    def __init__(self):
        super().__init__()
        t = _r50.resnet50(strides=(2, 2, 2, 1))
        self.resnet50 = t
        self.stage1 = nn.Sequential(t.conv1, t.bn1, t.relu, t.maxpool)
        self.stage2 = nn.Sequential(t.layer1)
        self.stage3 = nn.Sequential(t.layer2)
        self.stage4 = nn.Sequential(t.layer3)
        self.stage5 = nn.Sequential(t.layer4)
        self.mean_shift = MeanShift(2)

        # boundary branch: five 32-channel taps at stride 4 -> 1x1 over their concat
        self.fc_edge1 = _head(64, 32, 4)
        self.fc_edge2 = _head(256, 32, 4)
        self.fc_edge3 = _head(512, 32, 4, up=2)
        self.fc_edge4 = _head(1024, 32, 4, up=4)
        self.fc_edge5 = _head(2048, 32, 4, up=4)
        self.fc_edge6 = nn.Conv2d(160, 1, 1, bias=True)

        # displacement branch
        self.fc_dp1 = _head(64, 64, 8)
        self.fc_dp2 = _head(256, 128, 16)
        self.fc_dp3 = _head(512, 256, 16)
        self.fc_dp4 = _head(1024, 256, 16, up=2)
        self.fc_dp5 = _head(2048, 256, 16, up=2)
        self.fc_dp6 = _head(768, 256, 16, up=2)
        self.fc_dp7 = _head(448, 256, 16, tail=(nn.Conv2d(256, 2, 1, bias=False), self.mean_shift))

        self.backbone = nn.ModuleList([self.stage1, self.stage2, self.stage3, self.stage4, self.stage5])
        self.edge_layers = nn.ModuleList([self.fc_edge1, self.fc_edge2, self.fc_edge3,
                                          self.fc_edge4, self.fc_edge5, self.fc_edge6])
        self.dp_layers = nn.ModuleList([self.fc_dp1, self.fc_dp2, self.fc_dp3, self.fc_dp4,
                                        self.fc_dp5, self.fc_dp6, self.fc_dp7])

    def forward(self, x):
        f1 = self.stage1(x)
        f2 = self.stage2(f1)
        f3 = self.stage3(f2)
        f4 = self.stage4(f3)
        f5 = self.stage5(f4)

        e2 = self.fc_edge2(f2)
        eh, ew = e2.shape[2:]
        taps = [self.fc_edge1(f1), e2] + [m(f)[..., :eh, :ew] for m, f in
                                          ((self.fc_edge3, f3), (self.fc_edge4, f4), (self.fc_edge5, f5))]
        edge = self.fc_edge6(torch.cat(taps, dim=1))

        d2 = self.fc_dp2(f2)
        d3 = self.fc_dp3(f3)
        dh, dw = d3.shape[2:]
        mid = torch.cat([d3, self.fc_dp4(f4)[..., :dh, :dw], self.fc_dp5(f5)[..., :dh, :dw]], dim=1)
        up3 = self.fc_dp6(mid)[..., :d2.shape[2], :d2.shape[3]]
        dp = self.fc_dp7(torch.cat([self.fc_dp1(f1), d2, up3], dim=1))
        return edge, dp


class EdgeDisplacement(Net):
    """[2,3,H,W] (image, flip) -> edge [1,h,w] in (0,1) and dp [2,h,w], h=ceil(H/4)
    (net/resnet50_irn.py:216-234): zero-pad to crop_size, run, crop to the strided size,
    average the two boundary logits (flip undone) through a sigmoid, take dp of the un-flipped."""

    def __init__(self, crop_size=512, stride=4):
        super().__init__()
        self.crop_size = crop_size
        self.stride = stride

    def forward(self, x):
        H, W = x.shape[2:]
        h, w = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
        x = F.pad(x, [0, self.crop_size - W, 0, self.crop_size - H])
        e, d = super().forward(x)
        e = e[..., :h, :w]
        d = d[..., :h, :w]
        return torch.sigmoid(e[0] / 2 + e[1].flip(-1) / 2), d[0]
