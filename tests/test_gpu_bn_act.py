"""The trunk's fused inference epilogue (irn_bn_act, irn_amd/csrc/bn_act.hip) against the composed operations it
replaces — FixedBatchNorm -> `out += residual` -> ReLU of reference net/resnet50.py:11-14, :34-54, :87-89.

The kernel does one fused multiply-add per element with constants folded in double precision, so it is compared with
the exact (fp64) value of the same expression at fp32 rounding accuracy, with PyTorch's own batch_norm / add / relu at
the accuracy those have among themselves, and through the whole CAM / IRNet forwards with the fusion on and off."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


def _case(shape, seed, residual):
    g = torch.Generator().manual_seed(seed)
    c = shape[1]
    x = torch.randn(shape, generator=g)
    res = torch.randn(shape, generator=g) if residual else None
    scale = torch.rand(c, generator=g) * 2 - 0.5
    shift = torch.randn(c, generator=g)
    return x, res, scale, shift


# planes that are / are not a multiple of four long, a piece straddling two planes at every offset, planes shorter than
# a piece, a flat tail, one channel, one image, many channels
SHAPES = [(2, 8, 16, 16), (3, 5, 7, 9), (2, 3, 1, 5), (4, 6, 1, 1), (1, 1, 3, 1), (1, 7, 1, 3), (2, 64, 33, 47), (1, 2048, 2, 3),
          (16, 3, 5, 5), (1, 1, 1, 1027), (2, 1, 31, 2)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_bn_act_equals_the_exact_expression(shape, residual, relu):
    from irn_amd import ops
    dev = _dev()
    x, res, scale, shift = _case(shape, 11 + len(shape) * shape[1], residual)
    view = (1, -1) + (1,) * (len(shape) - 2)
    exact = x.double() * scale.double().view(view) + shift.double().view(view)
    fma32 = exact.float()                                     # the kernel's fmaf: one rounding of the exact value
    if residual:
        exact = fma32.double() + res.double()                 # then one fp32 addition
    want = exact.float()
    if relu:
        want = torch.clamp_min(want, 0)
    xd = x.to(dev)
    out = ops.bn_act_(xd, scale.to(dev), shift.to(dev), None if res is None else res.to(dev), relu)
    assert out.data_ptr() == xd.data_ptr()                    # in place
    assert torch.equal(out.cpu(), want), float((out.cpu() - want).abs().max())


def test_bn_act_nan_and_signed_zero():
    from irn_amd import ops
    dev = _dev()
    x = torch.tensor([[[float("nan"), -1.0, 2.0, float("inf"), -float("inf"), 0.0, -0.0, 1.0]]], device=dev).view(1, 1, 8)
    out = ops.bn_act_(x.clone(), torch.ones(1, device=dev), torch.zeros(1, device=dev), None, True).cpu().view(-1)
    ref = torch.relu(x.cpu().view(-1))
    assert torch.isnan(out[0]) and torch.isnan(ref[0])
    assert torch.equal(out[1:], ref[1:])


def test_bn_act_refuses_what_it_cannot_do():
    from irn_amd import ops
    dev = _dev()
    x = torch.zeros(2, 4, 3, 3, device=dev)
    s = torch.ones(4, device=dev)
    with pytest.raises(ValueError):
        ops.bn_act_(x.cpu(), s, s)
    with pytest.raises(ValueError):
        ops.bn_act_(x.permute(0, 1, 3, 2)[:, :, :, :2], s, s)                 # not contiguous
    with pytest.raises(ValueError):
        ops.bn_act_(x, torch.ones(3, device=dev), s)
    with pytest.raises(ValueError):
        ops.bn_act_(x, s, s, residual=torch.zeros(2, 4, 3, 2, device=dev))
    with pytest.raises(ValueError):
        ops.bn_act_(x.double(), s, s)
    assert ops.bn_act_(torch.zeros(0, 4, 3, 3, device=dev), s, s).numel() == 0


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (3, 5, 7, 9), (2, 3, 1, 5), (4, 6, 1, 1), (2, 64, 33, 47)])
@pytest.mark.parametrize("relu", [False, True])
def test_bn_act_with_a_batch_norm_on_the_residual(shape, relu):
    """The projection shortcut's batch norm in the same pass: fl(fma(x, s, b)) + fl(fma(res, rs, rb)), bit for bit."""
    from irn_amd import ops
    dev = _dev()
    x, res, scale, shift = _case(shape, 31 + shape[1], True)
    g = torch.Generator().manual_seed(77)
    rs, rb = torch.rand(shape[1], generator=g) * 2 - 0.5, torch.randn(shape[1], generator=g)
    view = (1, -1, 1, 1)
    a = (x.double() * scale.double().view(view) + shift.double().view(view)).float()
    b = (res.double() * rs.double().view(view) + rb.double().view(view)).float()
    want = a + b
    if relu:
        want = torch.clamp_min(want, 0)
    got = ops.bn_act_(x.to(dev), scale.to(dev), shift.to(dev), res.to(dev), relu, (rs.to(dev), rb.to(dev)))
    assert torch.equal(got.cpu(), want), float((got.cpu() - want).abs().max())
    with pytest.raises(ValueError):
        ops.bn_act_(x.to(dev), scale.to(dev), shift.to(dev), None, relu, (rs.to(dev), rb.to(dev)))


def test_bottleneck_with_projection_fused_vs_composed(monkeypatch):
    from irn_amd.net import resnet50 as R
    dev = _dev()
    torch.manual_seed(8)
    unit = R.Bottleneck(64, 32, stride=2, project=True).to(dev).eval()
    with torch.no_grad():
        for m in unit.modules():
            if isinstance(m, R.FrozenBatchNorm):
                m.running_mean.normal_()
                m.running_var.uniform_(0.3, 2.0)
                m.weight.uniform_(-1.0, 1.5)
                m.bias.normal_()
    x = torch.randn(3, 64, 19, 23, device=dev)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(R, "FUSED_EPILOGUE", fused)
        with torch.no_grad():
            outs[fused] = unit(x)
    assert outs[True].shape == (3, 128, 10, 12)
    assert float((outs[True] - outs[False]).abs().max()) < 1e-5 * max(1.0, float(outs[False].abs().max()))
    y = unit(x.requires_grad_(True))                                             # autograd on: composed path, differentiable
    assert y.requires_grad and float((y.detach() - outs[False]).abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2, 16, 9, 13), (1, 32, 8, 8)])
def test_frozen_batch_norm_apply_equals_composed_ops(shape):
    """FrozenBatchNorm.apply_ on the device (fused) vs F.batch_norm -> + skip -> relu on the device and on the CPU."""
    from irn_amd.net import resnet50 as R
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    bn = R.FrozenBatchNorm(shape[1])
    with torch.no_grad():
        bn.weight.copy_(torch.rand(shape[1], generator=g) + 0.5)
        bn.bias.copy_(torch.randn(shape[1], generator=g))
        bn.running_mean.copy_(torch.randn(shape[1], generator=g))
        bn.running_var.copy_(torch.rand(shape[1], generator=g) + 0.1)
    x = torch.randn(shape, generator=g)
    skip = torch.randn(shape, generator=g)
    with torch.no_grad():
        want = F.relu(bn(x) + skip)
        bnd = bn.to(dev)
        got = bnd.apply_(x.to(dev), residual=skip.to(dev), relu=True)
        composed = F.relu(bnd(x.to(dev)) + skip.to(dev))
    assert float((got.cpu() - want).abs().max()) < 2e-6
    assert float((got - composed).abs().max()) < 2e-6
    # with autograd on the composed path runs and the input is left alone
    xg = x.to(dev).requires_grad_(True)
    y = bnd.apply_(xg, residual=skip.to(dev), relu=True)
    assert y.requires_grad and y.data_ptr() != xg.data_ptr()
    assert float((y.detach() - composed).abs().max()) == 0.0


def test_folded_constants_follow_the_parameters():
    from irn_amd.net import resnet50 as R
    dev = _dev()
    bn = R.FrozenBatchNorm(4).to(dev)
    s0, b0 = bn.folded()
    assert bn.folded()[0] is s0                                                # cached
    with torch.no_grad():
        bn.running_var.fill_(4.0)
        bn.running_mean.fill_(1.0)
    s1, b1 = bn.folded()
    assert s1 is not s0
    np.testing.assert_allclose(s1.cpu().numpy(), 1 / np.sqrt(4.0 + bn.eps), rtol=1e-7)
    np.testing.assert_allclose(b1.cpu().numpy(), -1 / np.sqrt(4.0 + bn.eps), rtol=1e-7)
    bn.load_state_dict({"weight": torch.full((4,), 2.0), "bias": torch.zeros(4), "running_mean": torch.zeros(4),
                        "running_var": torch.ones(4), "num_batches_tracked": torch.tensor(0)})
    np.testing.assert_allclose(bn.folded()[0].cpu().numpy(), 2 / np.sqrt(1.0 + bn.eps), rtol=1e-7)


@pytest.mark.parametrize("size", [(64, 96), (75, 101)])
def test_backbones_fused_vs_composed(size, monkeypatch):
    """The whole CAM and IRNet forwards with the fused epilogue on and off (same weights, same MIOpen convolutions):
    what the fusion changes is one rounding per layer."""
    from irn_amd.net import resnet50 as R, resnet50_cam, resnet50_irn, weights
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 3) + size, generator=g).to(dev)
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(seed=1), strict=True)
    cam = cam.to(dev).eval()
    irn = resnet50_irn.EdgeDisplacement(crop_size=128)
    irn.load_state_dict(weights.random_irn_state(seed=2), strict=False)
    irn = irn.to(dev).eval()
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(R, "FUSED_EPILOGUE", fused)
        with torch.no_grad():
            e, d = irn(x)
            outs[fused] = (cam(x).cpu(), e.cpu(), d.cpu())
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


# ------------------------------------------------------------------------------------------------
# stem: batch norm + ReLU + max pool in one pass
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 3, 1, 1), (2, 5, 2, 3), (3, 64, 37, 51), (1, 4, 64, 130), (1, 2, 7, 129)])
def test_stem_pool_equals_the_composed_ops(shape):
    from irn_amd import ops
    dev = _dev()
    x, _, scale, shift = _case(shape, 3 + shape[2], False)
    fma32 = (x.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).float()
    want = F.max_pool2d(torch.clamp_min(fma32, 0), 3, 2, 1)
    xd = x.to(dev)
    got = ops.stem_pool(xd, scale.to(dev), shift.to(dev))
    assert torch.equal(xd.cpu(), x)                                             # the input is not written
    assert got.shape == want.shape and torch.equal(got.cpu(), want)


def test_stem_pool_nan_and_refusals():
    from irn_amd import ops
    dev = _dev()
    x = torch.zeros(1, 1, 6, 6)
    x[0, 0, 3, 3] = float("nan")                                               # a tap of four windows
    one, zero = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    got = ops.stem_pool(x.to(dev), one, zero).cpu()
    want = F.max_pool2d(torch.relu(x), 3, 2, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.isnan(got).sum() == 4
    with pytest.raises(ValueError):
        ops.stem_pool(x, one, zero)                                             # CPU tensor
    with pytest.raises(ValueError):
        ops.stem_pool(x.to(dev), torch.ones(2, device=dev), zero)
    assert ops.stem_pool(torch.zeros(0, 1, 6, 6, device=dev), one, zero).shape == (0, 1, 3, 3)


def test_stem_module_fused_vs_composed(monkeypatch):
    from irn_amd.net import resnet50 as R
    dev = _dev()
    torch.manual_seed(4)
    trunk = R.ResNet50Trunk(strides=(2, 2, 2, 1)).to(dev).eval()
    with torch.no_grad():
        trunk.bn1.running_mean.normal_()
        trunk.bn1.running_var.uniform_(0.2, 2.0)
        trunk.bn1.weight.uniform_(-1.0, 1.5)                                    # negative scales too
    st = R.Stem(trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool)
    x = torch.randn(2, 3, 75, 101, device=dev)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(R, "FUSED_EPILOGUE", fused)
        with torch.no_grad():
            outs[fused] = st(x)
    assert outs[True].shape == outs[False].shape == (2, 64, 19, 26)
    assert float((outs[True] - outs[False]).abs().max()) < 1e-5
    assert list(st.state_dict().keys()) == list(torch.nn.Sequential(trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool).state_dict().keys())


# ------------------------------------------------------------------------------------------------
# IRNet heads: bilinear upsampling + ReLU
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape,factor", [((2, 32, 16, 16), 2), ((2, 32, 8, 8), 4), ((1, 3, 5, 7), 2), ((1, 3, 5, 7), 4), ((2, 2, 1, 1), 4),
                                          ((1, 4, 9, 5), 3), ((1, 2, 6, 3), 1), ((3, 1, 33, 2), 2), ((1, 1, 2, 129), 4)])
@pytest.mark.parametrize("relu", [False, True])
def test_upsample_bilinear_equals_torch(shape, factor, relu):
    from irn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(shape[2] * 7 + factor)
    x = torch.randn(shape, generator=g)
    up = torch.nn.Upsample(scale_factor=factor, mode="bilinear", align_corners=False)
    want = up(x)
    want_dev = up(x.to(dev)).cpu()
    if relu:
        want, want_dev = torch.relu(want), torch.relu(want_dev)
    got = ops.upsample_bilinear(x.to(dev), factor, relu=relu).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))        # ATen on the CPU (the reference's run)
    assert float((got - want_dev).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))    # ATen on the device
    # exact statement of the documented expression
    h, w = shape[2:]
    r = np.float32(1.0 / factor)
    def axis(n_out, n_in):
        src = np.maximum(r * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0)).astype(np.float32)
        i0 = src.astype(np.int64)
        return i0, i0 + (i0 < n_in - 1), (src - i0.astype(np.float32)).astype(np.float32)
    y0, y1, ly = axis(h * factor, h)
    x0, x1, lx = axis(w * factor, w)
    a = x.numpy()
    ly, lx = ly[:, None], lx[None, :]
    one = np.float32(1)
    top = (one - lx) * a[..., y0[:, None], x0[None, :]] + lx * a[..., y0[:, None], x1[None, :]]
    bot = (one - lx) * a[..., y1[:, None], x0[None, :]] + lx * a[..., y1[:, None], x1[None, :]]
    exact = ((one - ly) * top + ly * bot).astype(np.float32)
    if relu:
        exact = np.maximum(exact, 0)
    assert np.array_equal(got.numpy(), exact)


def test_upsample_refusals():
    from irn_amd import ops
    dev = _dev()
    with pytest.raises(ValueError):
        ops.upsample_bilinear(torch.zeros(1, 1, 4, 4), 2)
    with pytest.raises(ValueError):
        ops.upsample_bilinear(torch.zeros(1, 1, 4, 4, device=dev).transpose(2, 3)[..., :2], 2)
    for bad in (0, 2.5, 65):
        with pytest.raises(ValueError):
            ops.upsample_bilinear(torch.zeros(1, 1, 4, 4, device=dev), bad)
    assert ops.upsample_bilinear(torch.zeros(0, 3, 4, 4, device=dev), 2).shape == (0, 3, 8, 8)


# ------------------------------------------------------------------------------------------------
# against the oracle and the reference's own modules (tests/golden/trunk_ops.npz)
# ------------------------------------------------------------------------------------------------

def test_trunk_tails_hip_vs_oracle_and_reference_golden(golden):
    """irn_bn_act / irn_stem_pool / irn_upsample_bilinear through the C ABI on the reference's fixtures: bit-identical to
    the oracle's folded restatement (bn, residual, pool) and within fp32 rounding of the outputs of the reference's
    FixedBatchNorm / MaxPool2d / Upsample modules."""
    from oracle import irn_oracle as O
    from irn_amd import ops
    dev = _dev()
    g = golden("trunk_ops")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def fold(prefix):
        return O.fold_batch_norm(g[prefix + "_w"], g[prefix + "_b"], g[prefix + "_mean"], g[prefix + "_var"], float(g[prefix + "_eps"]))

    def close(a, ref, tol=2e-6):
        assert float(np.abs(a - ref).max()) <= tol * max(1.0, float(np.abs(ref).max()))

    for tag in "abc":
        x, res = g["x_" + tag], g["res_" + tag]
        s, b = fold("bn_" + tag)
        sd, bd = fold("bnd_" + tag)
        cases = (("bn_plain_", None, False, None), ("bn_relu_", None, True, None), ("bn_add_relu_", res, True, None),
                 ("bn_addbn_relu_", res, True, (sd, bd)))
        for key, r, relu, aff in cases:
            got = ops.bn_act_(T(x), T(s), T(b), None if r is None else T(r), relu,
                              None if aff is None else (T(aff[0]), T(aff[1]))).cpu().numpy()
            assert np.array_equal(got, O.bn_act(x, s, b, res=r, relu=relu, res_affine=aff)), key + tag
            close(got, g[key + tag], 4e-6)
    for tag in ("s1", "s2", "s3"):
        s, b = fold("stem_" + tag)
        got = ops.stem_pool(T(g["stem_x_" + tag]), T(s), T(b)).cpu().numpy()
        assert np.array_equal(got, O.stem_pool(g["stem_x_" + tag], s, b))
        close(got, g["stem_out_" + tag])
    for tag in ("u2", "u4", "u2b"):
        f = int(g["up_f_" + tag])
        got = ops.upsample_bilinear(T(g["up_x_" + tag]), f, relu=True).cpu().numpy()
        close(got, O.head_upsample_relu(g["up_x_" + tag], f), 3e-7)
        close(got, g["up_out_" + tag], 3e-7)


def test_module_folding_equals_the_oracle_folding():
    from oracle import irn_oracle as O
    from irn_amd.net import resnet50 as R
    dev = _dev()
    torch.manual_seed(12)
    bn = R.FrozenBatchNorm(37)
    with torch.no_grad():
        bn.weight.normal_()
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.01, 3.0)
    s, b = O.fold_batch_norm(bn.weight.detach().numpy(), bn.bias.detach().numpy(), bn.running_mean.numpy(), bn.running_var.numpy(), bn.eps)
    sd, bd = bn.to(dev).folded()
    assert np.array_equal(sd.cpu().numpy(), s) and np.array_equal(bd.cpu().numpy(), b)


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (3, 64, 7, 9), (1, 2048, 2, 3), (2, 256, 33, 47), (4, 4, 1, 5)])
@pytest.mark.parametrize("mode", ["plain", "residual", "residual_bn"])
@pytest.mark.parametrize("relu", [False, True])
def test_bn_act_channels_last_equals_nchw_bitwise(shape, mode, relu):
    """irn_bn_act_nhwc (the pass over a torch.channels_last tensor, for the trunk on MIOpen's NHWC solvers): one fmaf per
    element with the same constants, so it must equal the NCHW pass bit for bit, residual forms included."""
    from irn_amd import ops
    x, res, scale, shift = _case(shape, 11, mode != "plain")
    g = torch.Generator().manual_seed(5)
    aff = None
    if mode == "residual_bn":
        aff = ((torch.rand(shape[1], generator=g) + 0.5).to(_dev()), torch.randn(shape[1], generator=g).to(_dev()))
    scale, shift = scale.to(_dev()), shift.to(_dev())
    a = ops.bn_act_(x.to(_dev()).clone(), scale, shift, None if res is None else res.to(_dev()), relu, aff)
    xc = x.to(_dev()).contiguous(memory_format=torch.channels_last)
    rc = None if res is None else res.to(_dev()).contiguous(memory_format=torch.channels_last)
    assert not xc.is_contiguous() or 1 in shape[2:]
    b = ops.bn_act_(xc, scale, shift, rc, relu, aff)
    assert b.data_ptr() == xc.data_ptr() and b.stride() == xc.stride()
    assert torch.equal(a, b.contiguous())


def test_bn_act_channels_last_refusals():
    from irn_amd import ops
    x = torch.randn(2, 6, 5, 5, device=_dev()).contiguous(memory_format=torch.channels_last)     # 6 channels: not a multiple of 4
    s = torch.ones(6, device=_dev())
    with pytest.raises(ValueError):
        ops.bn_act_(x, s, s)
    x = torch.randn(2, 8, 5, 5, device=_dev()).contiguous(memory_format=torch.channels_last)
    s = torch.ones(8, device=_dev())
    with pytest.raises(ValueError):
        ops.bn_act_(x, s, s, torch.randn(2, 8, 5, 5, device=_dev()))                             # residual in another memory format


_CL_SCRIPT = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from irn_amd.net import resnet50 as r50, resnet50_cam, resnet50_irn, weights
assert r50.CHANNELS_LAST_MODE == os.environ.get("IRN_CHANNELS_LAST")
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
x = torch.randn(4, 3, 160, 208, generator=g).to(dev)
cam = resnet50_cam.CAM(); cam.load_state_dict(weights.random_cam_state(1)); cam = cam.to(dev).eval()
irn = resnet50_irn.EdgeDisplacement(); irn.load_state_dict(weights.random_irn_state(2), strict=False)
irn = irn.to(dev).eval()
with torch.no_grad():
    c = cam.forward_batch(x)
    e = irn.forward_batch([x[:2], x[2:, :, :150, :199]])
np.savez(sys.argv[1], cam=c.cpu().numpy(), e0=e[0][0].cpu().numpy(), d0=e[0][1].cpu().numpy(), e1=e[1][0].cpu().numpy(), d1=e[1][1].cpu().numpy())
"""


def test_backbones_channels_last_mode_equals_nchw(tmp_path):
    """IRN_CHANNELS_LAST=1 (the trunk's stages on channels-last activations, everything around them converted at the
    seams) gives the CAM and IRNet outputs of the default NCHW run up to MIOpen's choice of solver."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "cl.py"
    script.write_text(_CL_SCRIPT % root)
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, IRN_CHANNELS_LAST=mode)
        out = subprocess.run([sys.executable, str(script), str(tmp_path / ("m%s.npz" % mode))], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        outs[mode] = np.load(tmp_path / ("m%s.npz" % mode))
    worst = {}
    for k in outs["0"].files:
        a, b = outs["0"][k], outs["1"][k]
        scale = max(float(np.abs(a).max()), 1e-6)
        worst[k] = float(np.abs(a - b).max()) / (scale if k == "cam" else 1.0)
    print("channels-last vs NCHW trunk: max deviation", {k: "%.2e" % v for k, v in worst.items()})
    assert worst["cam"] <= 1e-4 and max(worst["e0"], worst["e1"]) <= 1e-4 and max(worst["d0"], worst["d1"]) <= 2e-3
