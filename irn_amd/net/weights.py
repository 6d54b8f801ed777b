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


def _meta_build_covered(factory, state):
    """factory() on the `meta` device (no storage, no initialiser runs, nothing global is patched — safe beside other
    threads that build modules) and whether `state` covers every parameter and buffer of it."""
    with torch.device("meta"):
        model = factory()
    own = dict(model.state_dict())
    covered = all(k in state and tuple(state[k].shape) == tuple(v.shape) for k, v in own.items())
    return model, covered


def load_checkpoint(factory, path, strict):
    """`factory()` with the state dict at `path` loaded — what the steps do at the start of `run(args)` (reference
    step/make_cam.py:63-65: construct, torch.load, load_state_dict, eval) without paying for a random initialisation
    that the checkpoint overwrites (0.2-0.3 s per ResNet-50 on the host), and reading the file through a memory map when
    its format allows.  The network is built on the `meta` device and materialised from the checkpoint's tensors
    (`load_state_dict(assign=True)`); a checkpoint that does not cover every parameter and buffer (a partial one, or
    another architecture) goes through the regular construction + `load_state_dict`, errors and all."""
    try:
        state = torch.load(path, map_location="cpu", mmap=True, weights_only=True)
    except (RuntimeError, ValueError, TypeError) as e:      # a legacy (non-zip) file cannot be memory-mapped
        if "mmap" not in str(e).lower():
            raise
        state = torch.load(path, map_location="cpu", weights_only=True)
    model, covered = _meta_build_covered(factory, state)
    if covered:
        # assign=True makes the checkpoint's own tensors the parameters (clone: a memory-mapped file must be releasable,
        # and aliased entries of the reference's state dicts — stage1.0 == resnet50.conv1 — must stay ONE parameter)
        seen = {}
        own = {}
        for k, v in state.items():
            key = (v.data_ptr(), tuple(v.shape))
            if key not in seen:
                seen[key] = v.clone()
            own[k] = seen[key]
        model.load_state_dict(own, strict=strict, assign=True)
        if any(p.is_meta for p in model.parameters()) or any(b.is_meta for b in model.buffers()):
            covered = False                                  # something the state dict does not name (non-persistent buffer)
    if not covered:
        model = factory()
        model.load_state_dict(state, strict=strict)
    model.eval()
    return model
