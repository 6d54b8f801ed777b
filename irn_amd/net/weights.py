"""Seeded random checkpoints for the CAM / IRNet backbones.

Neither build box nor GPU box has network access or trained ``.pth`` files, so tests, golden
fixtures and bench.py use random-init weights of the reference architectures.  The returned
dicts have exactly the reference's state-dict keys (aliases included), so they load into the
reference's modules and into ours alike.
"""
import math

import torch


def _fill(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() == 4:                                   # conv kernel: He-normal
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan_in))
            elif name.endswith("weight"):                      # norm scale
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            else:                                              # norm / conv bias
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in module.named_buffers():
            if name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
    return module


def random_resnet50_state(seed=0):
    from .resnet50 import ResNet50Trunk
    return _fill(ResNet50Trunk(strides=(2, 2, 2, 1)), seed).state_dict()


def random_cam_state(seed=1):
    from .resnet50_cam import CAM
    return _fill(CAM(), seed).state_dict()


def random_irn_state(seed=2):
    from .resnet50_irn import EdgeDisplacement
    return _fill(EdgeDisplacement(), seed).state_dict()
