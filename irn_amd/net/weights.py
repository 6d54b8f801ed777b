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


class skip_param_init:
    """Context manager: build a network without running the random initialisers of its parameters (0.2-0.3 s per ResNet-50 on
    the host) — for modules whose every parameter is about to be overwritten by a checkpoint.  Buffers and constant
    initialisers run as usual.  `load_checkpoint` below falls back to the normal construction when a checkpoint turns out
    not to cover every parameter."""
    _NAMES = ("kaiming_uniform_", "kaiming_normal_", "uniform_", "normal_", "xavier_uniform_", "xavier_normal_", "trunc_normal_")

    def __enter__(self):
        import torch.nn.init as init
        self._saved = {n: getattr(init, n) for n in self._NAMES}
        for n in self._NAMES:
            setattr(init, n, lambda tensor, *a, **k: tensor)
        return self

    def __exit__(self, *exc):
        import torch.nn.init as init
        for n, f in self._saved.items():
            setattr(init, n, f)
        return False


def load_checkpoint(factory, path, strict):
    """`factory()` with the state dict at `path` loaded — what the steps do at the start of `run(args)` (reference
    step/make_cam.py:63-65: construct, torch.load, load_state_dict, eval) without paying for an initialisation that the
    checkpoint overwrites, and reading the file through a memory map when its format allows."""
    try:
        state = torch.load(path, map_location="cpu", mmap=True, weights_only=True)
    except Exception:
        state = torch.load(path, map_location="cpu")
    with skip_param_init():
        model = factory()
    result = model.load_state_dict(state, strict=strict)
    if getattr(result, "missing_keys", None):
        model = factory()                                   # a partial checkpoint: the rest keeps its regular initialisation
        model.load_state_dict(state, strict=strict)
    model.eval()
    return model
