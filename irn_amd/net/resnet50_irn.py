"""IRNet (boundary + displacement heads on a ResNet-50 trunk), inference form, host side on
PyTorch-ROCm.  API mirror of reference net/resnet50_irn.py:7-133 (``Net``) and :216-234
(``EdgeDisplacement``); attribute names reproduce the reference's state-dict keys.

``AffinityDisplacementLoss`` (net/resnet50_irn.py:144-213) is mirrored as the training SEAM only (SURVEY.md §8f
rank 4): its two gather operators run as differentiable HIP ops (``irn_amd.misc.indexing.edge_to_affinity``,
``pair_displacement``); the training loop, optimiser and datasets around it are out of scope.
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
    return _Head(*mods)


class _Head(nn.Sequential):
    """A head's module list with the reference's state-dict keys (net/resnet50_irn.py:21-97).  On the inference path
    (GPU, autograd off) `Upsample -> ReLU` is one pass of a hand-written kernel (`ops.upsample_bilinear`): ATen's generic
    bilinear kernel took 4 % of the end-to-end time for writing 0.5 GB per batch."""

    def forward(self, x):
        mods = list(self)
        i = 0
        m0 = mods[0]
        if (isinstance(m0, nn.Conv2d) and m0.kernel_size == (1, 1) and m0.stride == (1, 1) and m0.bias is None and m0.groups == 1
                and _r50._gemm_path(x)):
            # a channels-last trunk feature: the head's 1x1 convolution is a GEMM over the activation as it lies (round 5: the
            # four trunk outputs used to be transposed to NCHW first, 1.2 GB of copies per IRNet pass); its 32-256-channel
            # result is what goes to NCHW for the GroupNorm
            from .. import ops
            x = ops.conv1x1_nhwc(x, m0.weight.detach().flatten(1)).contiguous()
            i = 1
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.Upsample) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and _r50._fused(x)
                    and m.mode == "bilinear" and not m.align_corners and float(m.scale_factor) == int(m.scale_factor)):
                from .. import ops
                x = ops.upsample_bilinear(x, int(m.scale_factor), relu=True)
                i += 2
                continue
            x = m(x)
            i += 1
        return x


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
        self.stage1 = _r50.Stem(t.conv1, t.bn1, t.relu, t.maxpool)
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
        cl = _r50.channels_last_for(x)      # the trunk's stages on MIOpen's NHWC solvers where they are tuned (resnet50.py)
        f1 = self.stage1(x).detach()        # the trunk is frozen (net/resnet50_irn.py:111-115)
        f2 = self.stage2(_r50.to_stage_format(f1, cl)).detach()
        f3 = self.stage3(f2).detach()
        f4 = self.stage4(f3).detach()
        f5 = self.stage5(f4).detach()
        _r50.end_trunk_pass()               # the heads run NCHW (reproducible mode: under MIOpen's deterministic attribute)
        if cl and not (_r50.FUSED_GEMM and _r50.FUSED_EPILOGUE):
            f2, f3, f4, f5 = (_r50.to_nchw(f) for f in (f2, f3, f4, f5))      # (else each head's first GEMM reads them channels-last)

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

    def trainable_parameters(self):
        return tuple(self.edge_layers.parameters()), tuple(self.dp_layers.parameters())

    def train(self, mode=True):
        super().train(mode)
        self.backbone.eval()                # net/resnet50_irn.py:139-141
        return self


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

    def forward_batch(self, items):
        """list of [2,3,H_i,W_i] (image, flip) pairs of ANY sizes <= crop_size -> list of (edge [1,h_i,w_i],
        dp [2,h_i,w_i]).  Every item is zero-padded to the crop like `forward` pads it (:225), so the ragged batch is ONE
        [2B,3,crop,crop] pass of the trunk and heads; each result is cropped and merged exactly as in `forward`."""
        cs = self.crop_size
        x = torch.stack([F.pad(it, [0, cs - it.shape[3], 0, cs - it.shape[2]]) for it in items]).flatten(0, 1)
        e, d = _r50.run_rows(lambda c: Net.forward(self, c), x)      # reproducible mode: passes of a fixed size
        out = []
        for i, it in enumerate(items):
            H, W = it.shape[2:]
            h, w = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
            ei, di = e[2 * i:2 * i + 2, :, :h, :w], d[2 * i:2 * i + 2, :, :h, :w]
            out.append((torch.sigmoid(ei[0] / 2 + ei[1].flip(-1) / 2), di[0]))
        return out


class AffinityDisplacementLoss(Net):
    """Training-time losses of IRNet (net/resnet50_irn.py:144-213): same constructor (a ``PathIndex``), same four
    outputs of ``forward(x, True)``.  ``to_affinity`` and ``to_pair_displacement`` are the HIP operators (forward and
    backward) instead of |S| index_select / slice / stack launches; GPU tensors only."""

    def __init__(self, path_index):
        super().__init__()
        self.path_index = path_index
        dst = torch.as_tensor(path_index.search_dst, dtype=torch.float32)            # [|S|, 2] (dy, dx)
        self.register_buffer("disp_target", dst.t().unsqueeze(0).unsqueeze(-1))       # [1, 2, |S|, 1]

    def to_affinity(self, edge):
        from ..misc import indexing
        return indexing.edge_to_affinity(edge, radius=self.path_index.radius, size=self.path_index.default_size)

    def to_pair_displacement(self, disp):
        from ..misc import indexing
        return indexing.pair_displacement(disp, self.path_index.radius)

    def to_displacement_loss(self, pair_disp):
        return torch.abs(pair_disp - self.disp_target)

    def forward(self, x, return_loss=True):
        edge_out, dp_out = super().forward(x)
        if return_loss is False:
            return edge_out, dp_out
        aff = self.to_affinity(torch.sigmoid(edge_out))
        pos_aff_loss = (-1) * torch.log(aff + 1e-5)
        neg_aff_loss = (-1) * torch.log(1. + 1e-5 - aff)
        pair_disp = self.to_pair_displacement(dp_out)
        return pos_aff_loss, neg_aff_loss, self.to_displacement_loss(pair_disp), torch.abs(pair_disp)
