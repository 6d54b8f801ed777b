"""Device input pipeline (irn_bicubic_resize_u8 / irn_msf_pack) through the C ABI: bit-exact against the
golden outputs of the reference's dataset class over Pillow, the oracle restatement, and — where Pillow is
installed — live Image.resize calls at VOC sizes."""
import numpy as np
import pytest
import torch

from oracle import msf_oracle as M

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_msf_pack_equals_reference_items(golden):
    from irn_amd import ops
    d = golden("msf")
    for name in "ab":
        outs = ops.msf_pack(_dev(d[name + "_img"]), tuple(d["scales"]))
        for i, o in enumerate(outs):
            g = d["%s_item%d" % (name, i)]
            assert o.dtype == torch.float32 and tuple(o.shape) == g.shape
            assert np.array_equal(o.cpu().numpy(), g), (name, i)


def test_bicubic_resize_equals_pillow_golden(golden):
    from irn_amd import ops
    d = golden("msf")
    for i in range(5):
        want = d["resize%d_out" % i]
        got = ops.bicubic_resize(_dev(d["resize%d_img" % i]), want.shape[:2])
        assert np.array_equal(got.cpu().numpy(), want), i
    got = ops.bicubic_resize(_dev(d["gray_img"]), (61, 33))
    assert np.array_equal(got.cpu().numpy(), d["gray_out"])


@pytest.mark.parametrize("h,w", [(375, 500), (500, 334), (281, 500), (512, 512)])
def test_msf_pack_voc_sizes_vs_oracle(h, w):
    from irn_amd import ops, synth
    img = synth.photo(h, w, seed=h + w)
    scales = (1.0, 0.5, 1.5, 2.0)
    outs = ops.msf_pack(_dev(img), scales)
    want = M.msf_item(img, scales)
    for o, g in zip(outs, want):
        assert np.array_equal(o.cpu().numpy(), g)
    # size-independent properties: entry 1 is the horizontal flip of entry 0; scale 1 is the normalised image
    for o in outs:
        assert torch.equal(o[1], torch.flip(o[0], dims=(-1,)))
    assert np.array_equal(outs[0][0].cpu().numpy(), np.transpose(M.normalize(img), (2, 0, 1)))


def test_bicubic_resize_vs_live_pillow_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    from irn_amd import ops
    rng = np.random.default_rng(11)
    for t in range(30):
        h, w = (int(v) for v in rng.integers(1, 300, 2))
        hs, ws = (int(v) for v in rng.integers(1, 400, 2))
        ch = (3, 1, 4)[t % 3]
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        pil = Image.fromarray(img[..., 0] if ch == 1 else img, {1: "L", 3: "RGB", 4: "RGBX"}[ch])
        want = np.asarray(pil.resize((ws, hs), Image.BICUBIC))
        got = ops.bicubic_resize(_dev(img), (hs, ws)).cpu().numpy()
        if ch == 1:
            got = got[..., 0]
        assert np.array_equal(got, want), (h, w, hs, ws, ch)


def test_bicubic_resize_large_and_constant():
    """1024^2 -> 2048^2 and back down: a constant image stays constant (weights sum to 1 << 22 up to the
    rounding the clip absorbs) and black/white saturate without wrap-around."""
    from irn_amd import ops
    for v in (0, 255, 77):
        img = torch.full((1024, 1024, 3), v, dtype=torch.uint8, device="cuda")
        up = ops.bicubic_resize(img, (2048, 2048))
        assert int(up.min()) == v and int(up.max()) == v
        down = ops.bicubic_resize(up, (300, 700))
        assert int(down.min()) == v and int(down.max()) == v


def test_msf_pack_argument_errors():
    from irn_amd import ops
    with pytest.raises(ValueError):
        ops.msf_pack(torch.zeros((4, 4, 3), device="cuda"), (1.0,))              # not uint8
    with pytest.raises(ValueError):
        ops.msf_pack(torch.zeros((4, 4), dtype=torch.uint8, device="cuda"), (1.0,))
    with pytest.raises(RuntimeError, match="positive"):
        ops.msf_pack(torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"), (0.01,))   # rounds to 0 x 0
