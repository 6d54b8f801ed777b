"""ResNet-50 trunk for the CAM / IRNet backbones, host side on PyTorch-ROCm (MIOpen convolutions).

Mirrors the parameter naming of the reference trunk (reference net/resnet50.py:11-108) so that
checkpoints written by the reference (`res50_cam.pth`, `res50_irn.pth`) load unchanged:
``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``.
Differences by design:
  * construction never touches the network (the reference downloads ImageNet weights in the
    constructor, net/resnet50.py:111-118); weights are an *input* to the hot path;
  * batch-norm layers always use their running statistics (reference `FixedBatchNorm`,
    net/resnet50.py:11-14) — here that is simply the module's forward.
"""
import torch.nn as nn
import torch.nn.functional as F

STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_PLANES = (64, 128, 256, 512)


class FrozenBatchNorm(nn.BatchNorm2d):
    """Inference-statistics batch norm regardless of train()/eval() (net/resnet50.py:11-14)."""

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                            False, 0.0, self.eps)


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride, dilation) -> 1x1(x4) residual unit (net/resnet50.py:17-57)."""
    expansion = 4

    def __init__(self, c_in, planes, stride=1, project=False, dilation=1):
        super().__init__()
        c_out = planes * self.expansion
        self.conv1 = nn.Conv2d(c_in, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, c_out, 1, bias=False)
        self.bn3 = FrozenBatchNorm(c_out)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride=stride, bias=False),
                                            FrozenBatchNorm(c_out))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


def _stage(c_in, planes, n_blocks, stride, dilation):
    c_out = planes * Bottleneck.expansion
    units = [Bottleneck(c_in, planes, stride=stride, project=(stride != 1 or c_in != c_out), dilation=1)]
    units += [Bottleneck(c_out, planes, dilation=dilation) for _ in range(n_blocks - 1)]
    return nn.Sequential(*units)


class ResNet50Trunk(nn.Module):
    """conv1/bn1/maxpool + four bottleneck stages; no pooling head (the reference never uses it)."""

    def __init__(self, strides=(2, 2, 2, 2), dilations=(1, 1, 1, 1)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=strides[0], padding=3, bias=False)
        self.bn1 = FrozenBatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        stage_strides = (1,) + tuple(strides[1:])
        c_in = 64
        for i, (planes, n, s, d) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS, stage_strides, dilations)):
            setattr(self, "layer%d" % (i + 1), _stage(c_in, planes, n, s, d))
            c_in = planes * Bottleneck.expansion

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnet50(pretrained=False, state_dict=None, **kw):
    """Build the trunk.  ``pretrained`` is accepted for signature compatibility with the reference
    (net/resnet50.py:111) but never downloads; pass ``state_dict`` to initialise explicitly."""
    net = ResNet50Trunk(**kw)
    if state_dict is not None:
        sd = {k: v for k, v in state_dict.items() if not k.startswith("fc.")}
        net.load_state_dict(sd)
    return net
