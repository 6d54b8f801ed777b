"""CAM classifier backbone (inference form), host side on PyTorch-ROCm.

API mirror of reference net/resnet50_cam.py: module attribute names (``resnet50, stage1..4,
classifier, backbone, newly_added``) reproduce the reference's state-dict keys, aliases included,
so ``load_state_dict(torch.load('res50_cam.pth'), strict=True)`` (step/make_cam.py:64) works.
"""
import torch.nn as nn
import torch.nn.functional as F

from . import resnet50 as _r50

N_CLASSES = 20


class Net(nn.Module):
    """Classifier form (net/resnet50_cam.py:7-47); only its parameters matter to the hot path."""

    def __init__(self):
        super().__init__()
        t = _r50.resnet50(strides=(2, 2, 2, 1))
        self.resnet50 = t
        self.stage1 = _r50.Stem(t.conv1, t.bn1, t.relu, t.maxpool, t.layer1)
        self.stage2 = nn.Sequential(t.layer2)
        self.stage3 = nn.Sequential(t.layer3)
        self.stage4 = nn.Sequential(t.layer4)
        self.classifier = nn.Conv2d(2048, N_CLASSES, 1, bias=False)
        self.backbone = nn.ModuleList([self.stage1, self.stage2, self.stage3, self.stage4])
        self.newly_added = nn.ModuleList([self.classifier])

    def features(self, x):
        return self.stage4(self.stage3(self.stage2(self.stage1(x))))

    def forward(self, x):
        f = self.features(x)
        _r50.end_trunk_pass()
        return self.classifier(f.mean(dim=(2, 3), keepdim=True)).flatten(1)


class CAM(Net):
    """[2,3,H,W] (image, h-flipped image) -> [20, ceil(H/16), ceil(W/16)] activation maps:
    relu(1x1 conv with the classifier weights), original + flipped-back (net/resnet50_cam.py:55-70)."""

    def forward(self, x):
        a = _r50.to_nchw(F.relu(F.conv2d(self.features(x), self.classifier.weight)))
        return a[0] + a[1].flip(-1)

    def forward_batch(self, x):
        """[2B,3,H,W] = B (image, h-flipped image) pairs of ONE size back to back -> [B,20,h,w]: the forward above for
        every pair in one pass of the trunk (the steps stack the images of a size group per scale).  In the reproducible
        mode the rows travel in passes of a fixed size per image size (`resnet50.run_rows`): a pair's maps do not depend on
        what else is in `x`."""
        a = _r50.run_rows(self._maps, x)
        return a[0::2] + a[1::2].flip(-1)

    def _maps(self, x):
        return _r50.to_nchw(F.relu(F.conv2d(self.features(x), self.classifier.weight)))
