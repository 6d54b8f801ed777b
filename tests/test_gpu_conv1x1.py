"""The trunk's 1x1 convolutions as GEMMs with a fused epilogue (irn_conv1x1_nhwc, irn_amd/csrc/conv1x1.cpp) against the
operations they replace — conv1 -> bn1 -> ReLU and conv3 -> bn3 -> (+ residual | downsample(x)) -> ReLU of reference
net/resnet50.py:34-54 with FixedBatchNorm (:11-14) folded into weight and bias.

Compared with the exact (fp64) value of the same expression at fp32 GEMM accuracy, with the composed PyTorch modules, and
for what the steps rely on: the same bits on every call and for every entry of the rank table."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


CASES = [  # n, cin, cout, h, w — ragged pixel counts, tiny and trunk-sized channel counts
    (2, 64, 256, 24, 32), (2, 256, 64, 24, 32), (1, 8, 4, 3, 5), (3, 128, 512, 13, 19), (2, 2048, 512, 6, 8), (16, 64, 64, 32, 32), (1, 4, 12, 1, 1)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("bias,residual,relu", [(True, False, True), (True, True, True), (False, False, False), (True, True, False), (False, True, True)])
def test_conv1x1_equals_the_exact_expression(case, bias, residual, relu):
    from irn_amd import ops
    n, cin, cout, h, w = case
    dev = _dev()
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, cout, h, w, generator=g) if residual else None
    want = torch.einsum("nchw,oc->nohw", x.double(), wt.double())
    if bias:
        want = want + b.double().view(1, -1, 1, 1)
    if residual:
        want = want + r.double()
    if relu:
        want = want.clamp_min(0)
    got = ops.conv1x1_nhwc(_cl(x.to(dev)), wt.to(dev), None if b is None else b.to(dev), None if r is None else _cl(r.to(dev)), relu)
    assert got.shape == (n, cout, h, w) and got.is_contiguous(memory_format=torch.channels_last)
    err = float((got.cpu().double() - want).abs().max())
    assert err <= 2e-6 * cin ** 0.5 + 1e-6, (case, err)            # an fp32 dot product of cin terms, correctly accumulated


def test_conv1x1_in_place_on_the_residual_and_into_a_given_output():
    """The projection unit writes conv3's result over the shortcut GEMM's output (C = D); `out=` is honoured."""
    from irn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = _cl(torch.randn(2, 32, 9, 11, generator=g).to(dev))
    wt = (torch.randn(48, 32, generator=g) / 6).to(dev)
    b = torch.randn(48, generator=g).to(dev)
    sc = _cl(torch.randn(2, 48, 9, 11, generator=g).to(dev))
    want = ops.conv1x1_nhwc(x, wt, b, sc.clone(memory_format=torch.channels_last), True)
    buf = sc.clone(memory_format=torch.channels_last)
    got = ops.conv1x1_nhwc(x, wt, b, buf, True, out=buf)
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, want)
    other = torch.empty_like(want)
    assert ops.conv1x1_nhwc(x, wt, b, sc, True, out=other).data_ptr() == other.data_ptr() and torch.equal(other, want)


def test_conv1x1_same_bits_every_call_and_for_every_listed_rank():
    """What the steps rely on: the kernel is a function of the problem and the shipped rank table only, and the
    GEMMs accumulate in a fixed order — repeated calls give identical bits, for rank 0 and for every rank the table names."""
    from irn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    problems = [(4096, 1024, 256, 1, 0, 1), (16384, 256, 1024, 1, 1, 1)] + [k for k in list(ops.gemm_ranks())[:4]]
    for (m, cin, cout, hb, hr, relu) in problems:
        if m > 70000:
            continue
        x = torch.randn(1, cin, m, 1, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev) if hb else None
        r = torch.randn(1, cout, m, 1, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if hr else None
        n_algo = ops.conv1x1_algo_count(m, cin, cout, hb, hr, relu)
        assert n_algo >= 1
        rank = ops.gemm_ranks().get((m, cin, cout, hb, hr, relu), 0)
        for k in sorted({0, min(rank, n_algo - 1)}):
            first = ops.conv1x1_nhwc(x, wt, b, r, bool(relu), algo_rank=k)
            for _ in range(3):
                assert torch.equal(ops.conv1x1_nhwc(x, wt, b, r, bool(relu), algo_rank=k), first), (m, cin, cout, k)
        with pytest.raises(Exception):
            ops.conv1x1_nhwc(x, wt, b, r, bool(relu), algo_rank=n_algo)            # outside hipBLASLt's list


def test_conv1x1_refuses_what_it_cannot_do():
    from irn_amd import ops
    dev = _dev()
    x = _cl(torch.randn(2, 8, 4, 4, device=dev))
    wt = torch.randn(6, 8, device=dev)
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x.contiguous(), wt)                                        # NCHW activation
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x.cpu(), wt.cpu())                                         # no CPU path
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x, torch.randn(6, 7, device=dev))                          # channel mismatch
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x, wt, torch.randn(5, device=dev))                         # bias length
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x, wt, residual=torch.randn(2, 6, 4, 4, device=dev))       # residual not channels-last
    with pytest.raises(ValueError):
        ops.conv1x1_nhwc(x.double(), wt.double())


@pytest.mark.parametrize("project,stride", [(False, 1), (True, 1), (True, 2)])
def test_bottleneck_gemm_path_equals_the_composed_unit(project, stride, monkeypatch):
    """Bottleneck.forward on a channels-last activation (GEMMs + MIOpen 3x3) against the same unit with the fused paths off
    (MIOpen convolutions + torch batch norm / add / ReLU), identity and projection shortcuts, stride 1 and 2."""
    from irn_amd.net import resnet50 as r50
    dev = _dev()
    torch.manual_seed(3)
    c_in = 64 if project else 128
    unit = r50.Bottleneck(c_in, 32, stride=stride, project=project).to(dev).eval()
    for m in unit.modules():
        if isinstance(m, r50.FrozenBatchNorm):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(); m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
    x = torch.randn(4, c_in, 20, 28, device=dev)
    with torch.no_grad():
        monkeypatch.setattr(r50, "FUSED_GEMM", True)
        got = unit(_cl(x))
        assert got.is_contiguous(memory_format=torch.channels_last)
        monkeypatch.setattr(r50, "FUSED_GEMM", False)
        monkeypatch.setattr(r50, "FUSED_EPILOGUE", False)
        want = unit(x)
    err = float((got - want).abs().max() / want.abs().max())
    print("bottleneck project=%s stride=%d: GEMM path vs composed ops, relative max deviation %.2e" % (project, stride, err))
    assert err <= 2e-6
    # the folded operands follow the parameters
    old = unit.gemm_params()["w1"].clone()
    with torch.no_grad():
        unit.bn1.weight.mul_(2.0)
    assert not torch.equal(unit.gemm_params()["w1"], old)
