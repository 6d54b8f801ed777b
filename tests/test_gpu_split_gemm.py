"""Split-precision 1x1 convolutions (round 6; irn_split16 + irn_gemm16_nhwc, irn_amd/csrc/split16.hip, conv1x1.cpp): fp16 hi/lo
operands with 2^-11-scaled low parts, one fp16 MFMA GEMM over 3 cin with fp32 accumulation, the same epilogue as the fp32 GEMM —
conv -> FixedBatchNorm -> (+ residual) -> ReLU of reference net/resnet50.py:11-14,34-54.

Checked against the exact (fp64) value of the same expression at the accuracy of the fp32 GEMM they replace, bit for bit against a
host model of the split, for the bits being the same on every call (what the reproducible mode rests on), and through a whole
bottleneck unit against the composed PyTorch modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def test_split16_is_the_host_model_bit_for_bit():
    """hi = fp16(y), lo' = fp16((y - hi) 2^11) with y = x or relu(fmaf(x, scale, shift)); layout [hi | hi | lo'] per pixel; the
    overflow flag trips on |y| > 65504 and on NaN only."""
    from irn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for (n, c, h, w) in ((2, 64, 7, 9), (1, 8, 1, 1), (3, 2048, 4, 5), (1, 128, 33, 17)):
        x = torch.randn(n, c, h, w, generator=g) * torch.tensor([1e-6, 1e-3, 1.0, 300.0])[torch.randint(0, 4, (n, c, h, w), generator=g)]
        scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        for bn, relu in ((False, False), (True, True), (True, False)):
            got = ops.split16(_cl(x.to(dev)), scale.to(dev) if bn else None, shift.to(dev) if bn else None, relu).cpu()
            y = x.permute(0, 2, 3, 1).reshape(-1, c)
            if bn:
                y = (y.double() * scale.double() + shift.double()).float()        # (the kernel's fmaf rounds once: within an fp32 ulp of this)
                if relu:
                    y = y.clamp_min(0)
            hi = y.to(torch.float16)
            lo = ((y - hi.float()) * 2048.0).to(torch.float16)
            assert got.shape == (n * h * w, 3 * c) and got.dtype == torch.float16
            if not bn:          # (with the batch norm in front the host's fmaf may not be a single rounding: compared below)
                assert torch.equal(got[:, :c], hi) and torch.equal(got[:, c:2 * c], hi) and torch.equal(got[:, 2 * c:], lo), (n, c, bn)
            else:
                assert torch.equal(got[:, :c], got[:, c:2 * c])
            # hi + 2^-11 lo' reproduces y to 22 bits wherever lo' is a normal number
            rec = got[:, :c].double() + got[:, 2 * c:].double() / 2048.0
            big = y.abs() > 1e-3
            assert float(((rec - y.double()).abs() / y.abs().double().clamp_min(1e-30))[big].max()) <= 2.0 ** -21
    assert not ops.split_overflowed()
    bad = torch.ones(1, 8, 2, 2)
    bad[0, 3, 1, 1] = 7e4
    ops.split16(_cl(bad.to(dev)))
    assert ops.split_overflowed() and not ops.split_overflowed()            # reported once, then reset
    bad[0, 3, 1, 1] = float("nan")
    ops.split16(_cl(bad.to(dev)))
    assert ops.split_overflowed()


CASES = [(2, 64, 256, 24, 32), (2, 256, 64, 24, 32), (3, 128, 512, 13, 19), (2, 2048, 512, 6, 8), (1, 1024, 2048, 8, 8), (16, 64, 64, 32, 32), (1, 8, 12, 1, 1)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("bias,residual,relu", [(True, False, True), (True, True, True), (False, False, False)])
def test_split_gemm_equals_the_exact_expression_like_the_fp32_gemm(case, bias, residual, relu):
    from irn_amd import ops
    n, cin, cout, h, w = case
    dev = _dev()
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)) * 3.0
    wt = torch.randn(cout, cin, generator=g) / cin ** 0.5 * 0.05            # folded-batch-norm-sized weights
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, cout, h, w, generator=g) if residual else None
    want = torch.einsum("nchw,oc->nohw", x.double(), wt.double())
    if bias:
        want = want + b.double().view(1, -1, 1, 1)
    if residual:
        want = want + r.double()
    if relu:
        want = want.clamp_min(0)
    b16, alpha = ops.split_weight(wt.double().to(dev))
    assert b16.shape == (cout, 3 * cin) and float(b16.float().abs().max()) < 2.0 ** 14
    a16 = ops.split16(_cl(x.to(dev)))
    got = ops.gemm16_nhwc(a16, b16, (n, cout, h, w), None if b is None else b.to(dev), None if r is None else _cl(r.to(dev)), relu, alpha)
    assert got.shape == (n, cout, h, w) and got.is_contiguous(memory_format=torch.channels_last) and got.dtype == torch.float32
    err = float((got.cpu().double() - want).abs().max())
    f32 = ops.conv1x1_nhwc(_cl(x.to(dev)), wt.to(dev), None if b is None else b.to(dev), None if r is None else _cl(r.to(dev)), relu)
    err32 = float((f32.cpu().double() - want).abs().max())
    scale = float(want.abs().max())
    print("%s bias %d residual %d relu %d: split %.2e, fp32 GEMM %.2e (max |value| %.2f)" % (case, bias, residual, relu, err, err32, scale))
    assert err <= 4e-7 * scale * max(1.0, cin ** 0.5 / 8) + 1e-6, (case, err, err32)      # the bound the fp32 GEMM is held to, scaled to these magnitudes
    assert err <= 4.0 * err32 + 1e-6, (case, err, err32)
    # same bits on every call, and in place over the residual
    again = ops.gemm16_nhwc(a16, b16, (n, cout, h, w), None if b is None else b.to(dev), None if r is None else _cl(r.to(dev)), relu, alpha)
    assert torch.equal(got, again)
    if residual:
        buf = _cl(r.to(dev)).clone(memory_format=torch.channels_last)
        out = ops.gemm16_nhwc(a16, b16, (n, cout, h, w), None if b is None else b.to(dev), buf, relu, alpha, out=buf)
        assert out.data_ptr() == buf.data_ptr() and torch.equal(out, got)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("n,c,cout,h,w", [(2, 64, 32, 9, 11), (1, 512, 512, 16, 16), (3, 128, 256, 7, 5), (2, 8, 8, 1, 1), (1, 16, 8, 1, 6)])
def test_conv3x3_as_nine_accumulating_split_gemms(n, c, cout, h, w, fused, monkeypatch):
    """ops.conv3x3_split (irn_split16_pad + irn_conv3x3_split_gemm: nine GEMMs over 3 cin, or — row-fused, the default — three over
    9 cin reading the same zero-bordered operand with overlapping rows) against the
    exact 3x3 / pad 1 convolution, at the accuracy of MIOpen's fp32 convolution; the bordered result read back through the
    batch-norm + ReLU + split pass equals the dense one; same bits on every call, also after other shapes used the buffers."""
    import torch.nn.functional as F
    from irn_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n * 100 + c + h)
    x = torch.relu(torch.randn(n, c, h, w, generator=g)) * 2.0
    wt = torch.randn(cout, c, 3, 3, generator=g) / (9 * c) ** 0.5
    want = F.conv2d(x.double(), wt.double(), None, 1, 1)
    monkeypatch.setattr(ops, "CONV3X3_ROW_FUSED", fused)
    w16, alpha = ops.split_weight_3x3(wt.double().to(dev))
    assert w16.shape == ((3, cout, 9 * c) if fused else (9, cout, 3 * c))
    xd = _cl(x.to(dev))
    pad = ops.conv3x3_split(xd, w16, alpha)
    assert pad.shape == (n * (h + 2) * (w + 2), cout)
    got = pad.view(n, h + 2, w + 2, cout)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).cpu().double()
    f32 = F.conv2d(xd, _cl(wt.to(dev)), None, 1, 1).cpu().double()
    err, err32, scale = float((got - want).abs().max()), float((f32 - want).abs().max()), float(want.abs().max())
    print("3x3 %s -> %d (%s): split GEMMs %.2e, MIOpen fp32 %.2e from fp64 (max |value| %.2f)" % ((n, c, h, w), cout, "three over 9 cin" if fused else "nine over 3 cin", err, err32, scale))
    assert err <= 4.0 * err32 + 2e-6 * scale and err <= 1e-5 * max(1.0, scale)
    interior = lambda t: t.view(n, h + 2, w + 2, cout)[:, 1:-1, 1:-1]       # (border rows hold garbage by contract)
    first = interior(pad).clone()
    # the bordered form through the tail pass: batch norm + ReLU + split of the interior == the same pass on the dense tensor
    scale_c, shift_c = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    if cout % 8 == 0:
        dense = _cl(pad.view(n, h + 2, w + 2, cout)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).contiguous())
        a = ops.split16_pad(pad, (n, cout, h, w), scale_c, shift_c, relu=True, in_padded=True)
        b = ops.split16(dense, scale_c, shift_c, relu=True)
        assert torch.equal(a, b)
    ops.conv3x3_split(_cl(torch.randn(1, c, h + 1, w + 2, generator=g).to(dev)), w16, alpha)       # another shape in between
    again = ops.conv3x3_split(xd, w16, alpha)
    assert torch.equal(interior(again), first)
    if fused:
        # a hipBLASLt build that refuses the overlapping-row operand: the same call falls back to nine GEMMs on the same taps
        monkeypatch.setattr(ops, "_ROW_FUSED_REFUSED", True)
        nine = interior(ops.conv3x3_split(xd, w16, alpha))
        assert float((nine - first).abs().max()) <= 2e-6 * max(1.0, scale)
    assert not ops.split_overflowed()


def test_bottleneck_split_path_vs_composed_modules_and_fp32_gemm_path(monkeypatch):
    """A whole unit (identity and projection, stride 1) on the channels-last inference path with IRN_SPLIT_GEMM on: against the
    composed PyTorch modules (conv -> FrozenBatchNorm -> ReLU ... in NCHW, what the reference runs) and against the fp32-GEMM
    path, at the accuracy the fp32 path itself has; the same bits on every call."""
    from irn_amd.net import resnet50 as r50
    dev = _dev()
    torch.manual_seed(11)
    for (c_in, planes, project) in ((256, 64, False), (1024, 512, True), (2048, 512, False), (64, 64, True)):
        unit = r50.Bottleneck(c_in, planes, stride=1, project=project).to(dev).eval()
        with torch.no_grad():
            for m in unit.modules():
                if isinstance(m, r50.FrozenBatchNorm):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.2)
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 2.0)
            x = torch.relu(torch.randn(2, c_in, 12, 10, device=dev))
            monkeypatch.setattr(r50, "SPLIT_GEMM", True)
            monkeypatch.setattr(r50, "SPLIT_MIN_PLANES", 64)
            monkeypatch.setattr(r50, "SPLIT_MIN_INPUT", 1 << 20)
            monkeypatch.setattr(r50, "SPLIT_MIN_ROWS_3X3", 1)             # the 512-plane units take the nine-GEMM 3x3 here too
            unit._gemm = None
            y_split = unit(_cl(x))
            p = unit.gemm_params()
            assert ("w2_16" in p) == (planes >= r50.SPLIT_MIN_PLANES_3X3)
            assert "w3_16" in p and (("w1_16" in p) == (c_in * (planes + (4 * planes if project else 0)) >= 1 << 20))
            assert torch.equal(y_split, unit(_cl(x)))
            monkeypatch.setattr(r50, "SPLIT_GEMM", False)
            unit._gemm = None
            y_f32 = unit(_cl(x))
            assert "w3_16" not in unit.gemm_params()
            y_ref = unit.double()(x.double())                      # NCHW, fp64: the composed modules
            unit.float()
        e_split, e_f32 = float((y_split.double() - y_ref).abs().max()), float((y_f32.double() - y_ref).abs().max())
        scale = float(y_ref.abs().max())
        print("unit %d -> %d planes, project %s: split path %.2e, fp32 GEMM path %.2e from fp64 (max |value| %.2f)" % (c_in, planes, project, e_split, e_f32, scale))
        assert e_split <= 4.0 * e_f32 + 1e-6 * scale and e_split <= 2e-5 * max(1.0, scale)


def test_production_passes_vs_the_reference_goldens(golden, monkeypatch):
    """The trunk exactly as the steps run it — 16-row channels-last passes, split-precision 1x1 GEMMs, row-fused split 3x3 in
    stages 2-4 — against the REFERENCE's own CPU forwards (tests/golden/nets512.npz, nets_scales.npz: net/resnet50_cam.py:55-70,
    net/resnet50_irn.py:216-234 run by tests/golden/make_golden.py) at the north star's 1e-4: CAM at 256^2 / 512^2 / 1024^2 with
    the golden pair among seven others, EdgeDisplacement with ragged images padded to the 512^2 crop."""
    from irn_amd import ops, synth
    from irn_amd.net import resnet50 as r50, resnet50_cam, resnet50_irn, weights
    dev = _dev()
    monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "1")
    monkeypatch.setattr(r50, "SPLIT_GEMM", True)
    calls = {"gemm16": 0, "conv3x3": 0}
    real16, real3 = ops.gemm16_nhwc, ops.conv3x3_split
    monkeypatch.setattr(ops, "gemm16_nhwc", lambda *a, **k: (calls.__setitem__("gemm16", calls["gemm16"] + 1), real16(*a, **k))[1])
    monkeypatch.setattr(ops, "conv3x3_split", lambda *a, **k: (calls.__setitem__("conv3x3", calls["conv3x3"] + 1), real3(*a, **k))[1])
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(seed=1), strict=True)
    cam = cam.to(dev).eval()
    norm = lambda t: t / (t.max(axis=(1, 2), keepdims=True) + 1e-5)
    rel = lambda a, ref: float(np.abs(a - ref).max() / max(1.0, np.abs(ref).max()))
    for gname, key in (("nets512", "cam512"), ("nets_scales", "cam256"), ("nets_scales", "cam1024")):
        g = golden(gname)
        h, w, seed = (int(v) for v in g[key + "_seed"])
        pairs = [torch.from_numpy(synth.image_pair(h, w, seed + 100 * i)) for i in range(8)]       # pair 0 is the reference's input
        before = dict(calls)
        with torch.no_grad():
            y = cam.forward_batch(torch.cat(pairs[3:] + pairs[:3]).to(dev)).cpu().numpy()          # ... at position 5 of the pass
        assert calls["gemm16"] - before["gemm16"] >= 16 and calls["conv3x3"] - before["conv3x3"] >= (11 if h >= 512 else 3), (key, calls)      # the stride-1 3x3 of stages 2-4: 3 + 5 + 3 units (at 256^2 only stage 2 has >= 8192 rows)
        ref = g[key + "_out"]
        e_abs, e_norm = rel(y[5], ref), float(np.abs(norm(y[5]) - norm(ref)).max())
        print("%s in a 16-row split-precision pass: %.2e relative, %.2e on the normalised CAM (bar 1e-4)" % (key, e_abs, e_norm))
        assert y.shape == (8,) + ref.shape and e_abs <= 1e-4 and e_norm <= 1e-4, (key, e_abs, e_norm)
    g5 = golden("nets512")
    irn = resnet50_irn.EdgeDisplacement(crop_size=512)
    irn.load_state_dict(weights.random_irn_state(seed=2), strict=False)
    irn = irn.to(dev).eval()
    keys = ("irn", "irn512", "irn", "irn512", "irn", "irn", "irn512", "irn")
    items = []
    for key in keys:
        h, w, seed = (int(v) for v in g5[key + "_seed"])
        items.append(torch.from_numpy(synth.image_pair(h, w, seed)).to(dev))
    before = dict(calls)
    with torch.no_grad():
        outs = irn.forward_batch(items)
    assert calls["conv3x3"] - before["conv3x3"] >= 11
    worst = 0.0
    for key, (edge, dp) in zip(keys, outs):
        worst = max(worst, float(np.abs(edge.cpu().numpy() - g5[key + "_edge"]).max()), rel(dp.cpu().numpy(), g5[key + "_dp"]))
    print("EdgeDisplacement, 8 ragged images in one split-precision pass: worst deviation from the reference %.2e (bar 1e-4)" % worst)
    assert worst <= 1e-4
    assert not ops.split_overflowed()
