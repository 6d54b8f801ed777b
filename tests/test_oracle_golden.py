"""The oracle restatement is pinned on outputs of the reference itself (tests/golden/*.npz, written
by tests/golden/make_golden.py which imports /root/reference and runs it on CPU)."""
import numpy as np
import pytest

from oracle import irn_oracle as O


def _unpack(d, key):
    shape = tuple(d[key + "_shape"])
    return np.unpackbits(d[key])[:int(np.prod(shape))].reshape(shape).astype(bool)


@pytest.mark.parametrize("r,n_dirs,n_cells", [(2, 4, 12), (3, 12, 56), (5, 34, 242), (7, 72, 714), (10, 152, 2134)])
def test_path_tables(golden, r, n_dirs, n_cells):
    pt = golden("path_tables")
    groups, dst = O.search_paths_dst(r)
    assert len(dst) == n_dirs and sum(g.shape[0] * g.shape[1] for g in groups) == n_cells
    assert np.array_equal(dst, pt["r%d_dst" % r])
    assert np.array_equal([g.shape[1] for g in groups], pt["r%d_group_lens" % r])
    assert np.array_equal([g.shape[0] for g in groups], pt["r%d_group_counts" % r])
    assert np.array_equal(np.concatenate([g.reshape(-1, 2) for g in groups]), pt["r%d_paths_flat" % r])
    pi = O.PathIndexOracle(r, tuple(pt["r%d_size" % r]))
    assert np.array_equal(pi.src_indices, pt["r%d_src_indices" % r])
    assert np.array_equal(pi.dst_indices, pt["r%d_dst_indices" % r])
    assert np.array_equal(np.concatenate([p.reshape(-1) for p in pi.path_indices]), pt["r%d_path_indices_flat" % r])


def test_r5_reference_channel_order():
    # SURVEY.md §3.4 item 1 lists the r=5 channel order explicitly
    expect = [(0, 1), (1, 0), (0, 2), (2, 0), (0, 3), (1, -1), (1, 1), (3, 0), (0, 4), (4, 0), (1, -2), (1, 2),
              (2, -1), (2, 1), (2, -2), (2, 2), (1, -3), (1, 3), (2, -3), (2, 3), (3, -2), (3, -1), (3, 1), (3, 2),
              (1, -4), (1, 4), (3, -3), (3, 3), (4, -1), (4, 1), (2, -4), (2, 4), (4, -2), (4, 2)]
    _, dst = O.search_paths_dst(5)
    assert [tuple(d) for d in dst] == expect
    assert O.path_cells(1, 1) == [(1, 1), (0, 1), (1, 0), (0, 0)]


@pytest.mark.parametrize("r", [3, 5, 10])
def test_edge_to_affinity_exact(golden, r):
    af = golden("affinity")
    e = af["r%d_edge" % r]
    h, w = e.shape
    ep = np.ones((h + r, w + 2 * r), np.float32)
    ep[:h, r:r + w] = e
    pi = O.PathIndexOracle(r, (h + r, w + 2 * r))
    a = O.edge_to_affinity(ep.reshape(-1), pi.path_indices)
    assert np.array_equal(a, af["r%d_aff" % r])


def _walk_cases(golden):
    wk = golden("walk")
    return wk, sorted(k[:-3] for k in wk.files if k.endswith("_rw"))


def test_walk_stencil_matches_reference(golden):
    """fp64 stencil sweeps == the reference's dense fp32 squaring, to the reference's own fp32
    noise: <= 1e-4 max-abs (north-star tolerance) and identical channel argmax."""
    wk, names = _walk_cases(golden)
    assert len(names) >= 10
    for n in names:
        h, w, c, r, b, e = wk[n + "_params"]
        ref = wk[n + "_rw"]
        st = O.propagate_to_edge_stencil(wk[n + "_cam"], wk[n + "_edge"], r, b, e)
        assert st.shape == ref.shape
        assert np.abs(st - ref).max() <= 1e-4, n
        assert np.array_equal(np.argmax(st[:, 0], 0), np.argmax(ref[:, 0], 0)), n


def test_walk_dense_restatement_small(golden):
    """The line-by-line dense restatement (own BLAS, fp32) agrees with the reference to fp32 matmul
    noise on the grids where N^3 is cheap."""
    wk, names = _walk_cases(golden)
    for n in names:
        h, w, c, r, b, e = wk[n + "_params"]
        if h * w > 800:
            continue
        de = O.propagate_to_edge_dense(wk[n + "_cam"], wk[n + "_edge"], r, b, e)
        assert np.abs(de - wk[n + "_rw"]).max() <= 5e-4, n


def test_dense_torch_restatement_vs_reference_golden(golden):
    """oracle/dense_ref.py (the reference's own dense algorithm op for op on torch CPU tensors: what bench.py times as
    `cpu_baseline.reference_algorithm`) reproduces the reference's outputs bit for bit where the same torch kernels
    run the same sizes, i.e. on every small golden case."""
    from oracle import dense_ref
    wk, names = _walk_cases(golden)
    n_checked = 0
    for n in names:
        h, w, c, r, b, e = wk[n + "_params"]
        if h * w > 1100:
            continue
        x = wk[n + "_cam"]
        if n.endswith("_ck"):
            x = x.reshape(2, c // 2, h, w)
        de = dense_ref.propagate_to_edge(x, wk[n + "_edge"][None], int(r), int(b), int(e)).numpy()
        assert de.shape == wk[n + "_rw"].shape
        assert np.abs(de - wk[n + "_rw"]).max() <= 1e-6, n
        n_checked += 1
    assert n_checked >= 5


def test_stencil_conserves_degree_weighted_mass(golden):
    wk, _ = _walk_cases(golden)
    n = "r5_b10_e8"
    h, w, c, r, b, e = wk[n + "_params"]
    dirs, wts = O.stencil_weights(wk[n + "_edge"], r, b)
    deg = O.stencil_degree(dirs, wts)
    x = (wk[n + "_cam"] * (1 - wk[n + "_edge"])).astype(np.float64)
    y = O.stencil_sweep(x, dirs, wts.astype(np.float64), deg)
    assert np.allclose((deg * x).sum(axis=(1, 2)), (deg * y).sum(axis=(1, 2)), rtol=1e-12)


def test_sem_seg_epilogue_bit_exact(golden):
    wk, sg = golden("walk"), golden("semseg")
    names = sorted(k[:-6] for k in sg.files if k.endswith("_label"))
    assert len(names) == 4
    for n in names:
        H, W = sg[n + "_size"]
        up, lab, _ = O.sem_seg_epilogue(wk[n + "_rw"], (H, W), sg[n + "_keys"], float(sg[n + "_bg"]))
        assert np.array_equal(up, sg[n + "_rw_up"]), n
        assert np.array_equal(lab, sg[n + "_label"]), n


@pytest.mark.parametrize("name", ["a", "b", "c", "cen64", "cen_ragged"])
def test_centroids_and_clusters_bit_exact(golden, name):
    ins = golden("instance")
    dp = ins[name + "_dp"]
    cen = O.find_centroids_with_refinement(dp)
    assert np.array_equal(cen, ins[name + "_centroids"])
    assert np.array_equal(O.cluster_centroids(cen, dp), _unpack(ins, name + "_instance_map"))


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_instance_pipeline(golden, name):
    ins = golden("instance")
    H, W = ins[name + "_size"]
    cen, inst, rw, idx, det = O.instance_labels(ins[name + "_cam"], ins[name + "_keys"], ins[name + "_edge"],
                                                ins[name + "_dp"], (H, W))
    assert np.abs(rw - ins[name + "_rw"]).max() <= 1e-4
    assert np.array_equal(idx, ins[name + "_argmax"])
    assert np.array_equal(det["mask"], _unpack(ins, name + "_det_mask"))
    assert np.array_equal(det["class"], ins[name + "_det_class"])
    assert np.abs(np.asarray(det["score"], np.float32) - ins[name + "_det_score"]).max() <= 1e-4


def test_label4_matches_scipy():
    import scipy.ndimage
    rng = np.random.RandomState(0)
    for shape, p in (((17, 23), 0.5), ((40, 31), 0.62), ((8, 8), 0.9), ((5, 7), 0.0)):
        m = rng.rand(*shape) < p
        assert np.array_equal(O.label4(m), scipy.ndimage.label(m)[0])


@pytest.mark.parametrize("name", ["a", "b"])
def test_cam_merge(golden, name):
    cm = golden("cam_merge")
    outs = [cm["%s_out%d" % (name, i)] for i in range(4)]
    keys, lo, hi = O.cam_merge(outs, tuple(cm[name + "_size"]), cm[name + "_label"])
    assert np.array_equal(keys, cm[name + "_keys"])
    # 1-2 ulp: ATen contracts the source-index arithmetic differently per layout; the parity bar
    # on `cam` is 1e-4 (north star), this is 100x tighter
    assert np.abs(lo - cm[name + "_cam"]).max() <= 1e-6
    assert np.abs(hi - cm[name + "_high_res"]).max() <= 1e-6


@pytest.mark.parametrize("r", [3, 5])
def test_affinity_backward_restatement_vs_reference_autograd(golden, r):
    """oracle.edge_to_affinity_backward against the gradient autograd produced through the reference's
    AffinityDisplacementLoss.to_affinity (tests/golden/make_golden.py gen_affinity_grad)."""
    ag = golden("affinity_grad")
    ge = O.edge_to_affinity_backward(ag["r%d_edge" % r], ag["r%d_gout" % r], r)
    assert np.abs(ge - ag["r%d_gedge" % r]).max() <= 1e-5 * np.abs(ag["r%d_gedge" % r]).max()


# ------------------------------------------------------------------------------------------------
# Multi-scale input pipeline (oracle/msf_oracle.py): Pillow's 8-bit bicubic + the reference's item
# construction, pinned on outputs of the reference's own dataset class over the installed Pillow.
# ------------------------------------------------------------------------------------------------
def test_msf_item_restatement_vs_reference_dataset(golden):
    from oracle import msf_oracle as M
    d = golden("msf")
    for name in "ab":
        items = M.msf_item(d[name + "_img"], d["scales"])
        assert len(items) == 4
        for i, it in enumerate(items):
            g = d["%s_item%d" % (name, i)]
            assert it.dtype == np.float32 and it.shape == g.shape
            assert np.array_equal(it, g), (name, i)


def test_pil_bicubic_restatement_vs_pillow_golden(golden):
    from oracle import msf_oracle as M
    d = golden("msf")
    for i in range(5):
        want = d["resize%d_out" % i]
        assert np.array_equal(M.pil_bicubic_resize(d["resize%d_img" % i], want.shape[:2]), want), i
    assert np.array_equal(M.pil_bicubic_resize(d["gray_img"][..., None], (61, 33))[..., 0], d["gray_out"])


def test_pil_bicubic_restatement_vs_live_pillow():
    """Random sizes and scale factors against the installed Pillow (skipped where Pillow is absent)."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import msf_oracle as M
    rng = np.random.default_rng(3)
    for t in range(24):
        h, w = (int(v) for v in rng.integers(3, 90, 2))
        s = float(rng.choice([0.5, 1.5, 2.0, 0.37, 1.01, 0.99, 3.3, 0.1]))
        hs, ws = (max(v, 1) for v in M.rescale_size(h, w, s))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ws, hs), Image.BICUBIC))
        assert np.array_equal(M.pil_bicubic_resize(img, (hs, ws)), want), (h, w, s)


def test_rescale_size_rounds_half_to_even():
    from oracle import msf_oracle as M
    assert M.rescale_size(375, 500, 0.5) == (188, 250)      # 187.5 -> 188
    assert M.rescale_size(375, 500, 1.5) == (562, 750)      # 562.5 -> 562
    assert M.rescale_size(281, 500, 0.5) == (140, 250)      # 140.5 -> 140


@pytest.mark.parametrize("r", [3, 5, 10])
def test_pair_displacement_restatement_vs_reference_autograd(golden, r):
    """oracle.pair_displacement (+ backward) against AffinityDisplacementLoss.to_pair_displacement run under
    autograd (tests/golden/make_golden.py gen_pair_disp): the forward is a subtraction and exact."""
    pd = golden("pair_disp")
    disp, want = pd["r%d_disp" % r], pd["r%d_pair" % r]
    assert np.array_equal(O.pair_displacement(disp, r), want)
    g = np.random.RandomState(11 + r).randn(*want.shape).astype(np.float32)
    gd = O.pair_displacement_backward(g, r, disp.shape[2:])
    assert np.abs(gd - pd["r%d_gdisp" % r]).max() <= 1e-6 * np.abs(pd["r%d_gdisp" % r]).max()


def test_pair_displacement_backward_is_the_adjoint():
    """<J x, g> == <x, J^T g> for random x, g: the backward restatement is the transpose of the (linear) forward."""
    rng = np.random.RandomState(5)
    for r, (hp, wp) in ((3, (9, 11)), (5, (13, 21))):
        x = rng.randn(2, 2, hp, wp).astype(np.float32)
        y = O.pair_displacement(x, r)
        g = rng.randn(*y.shape).astype(np.float32)
        lhs = float((y.astype(np.float64) * g).sum())
        rhs = float((x.astype(np.float64) * O.pair_displacement_backward(g, r, (hp, wp))).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


def test_msf_oracle_identity_and_flip_properties():
    from irn_amd import synth
    from oracle import msf_oracle as M
    img = synth.photo(33, 47, seed=9)
    assert M.pil_bicubic_resize(img, (33, 47)) is img                      # both passes skipped (misc/imutils.py:9-10)
    items = M.msf_item(img, (1.0, 1.5))
    for it in items:
        assert np.array_equal(it[1], it[0][..., ::-1])
    # resizing commutes with a horizontal mirror only up to rounding of the centre taps; it does commute with channel
    # permutation exactly (bands are independent)
    perm = img[..., ::-1]
    assert np.array_equal(M.pil_bicubic_resize(perm, (50, 70)), M.pil_bicubic_resize(img, (50, 70))[..., ::-1])


def test_trunk_tails_vs_reference_modules(golden):
    """The oracle's folded batch norm / residual / ReLU, the stem's pool and the heads' upsampling against the outputs of
    the reference's own FixedBatchNorm (net/resnet50.py:11-14) and the torch modules it builds (tests/golden/trunk_ops.npz).
    Folding the batch norm into one multiply-add changes its rounding (ATen normalises first): <= 2e-6 relative; the
    pool is then exact on equal inputs.  ATen's CPU bilinear kernel contracts its multiply-adds differently from one
    loop specialisation to the next (the [C,1,h,w] x4 call of the label epilogue is pinned bit for bit elsewhere; these
    [N,C,h,w] calls land in another one), so the heads' upsampling is held to 2 ulp."""
    g = golden("trunk_ops")

    def fold(prefix):
        return O.fold_batch_norm(g[prefix + "_w"], g[prefix + "_b"], g[prefix + "_mean"], g[prefix + "_var"], float(g[prefix + "_eps"]))

    def close(a, ref, tol=2e-6):
        assert a.shape == ref.shape and a.dtype == np.float32
        assert float(np.abs(a - ref).max()) <= tol * max(1.0, float(np.abs(ref).max()))

    for tag in "abc":
        x, res = g["x_" + tag], g["res_" + tag]
        s, b = fold("bn_" + tag)
        close(O.bn_act(x, s, b, relu=False), g["bn_plain_" + tag])
        close(O.bn_act(x, s, b, relu=True), g["bn_relu_" + tag])
        close(O.bn_act(x, s, b, res=res, relu=True), g["bn_add_relu_" + tag])
        close(O.bn_act(x, s, b, res=res, relu=True, res_affine=fold("bnd_" + tag)), g["bn_addbn_relu_" + tag], 4e-6)
    for tag in ("s1", "s2", "s3"):
        s, b = fold("stem_" + tag)
        close(O.stem_pool(g["stem_x_" + tag], s, b), g["stem_out_" + tag])
    for tag in ("u2", "u4", "u2b"):
        close(O.head_upsample_relu(g["up_x_" + tag], int(g["up_f_" + tag])), g["up_out_" + tag], 3e-7)


def test_cam_ref_is_the_reference_forward_bit_for_bit(golden):
    """oracle/cam_ref.py (bench.py's `cpu_baseline.cam`): the functional restatement of net/resnet50.py:17-108 +
    net/resnet50_cam.py:55-70 gives the reference's own output on the reference's input with the same seeded weights —
    max-abs 0.0, not a tolerance; its flop count is what the product's networks count on meta tensors."""
    import torch
    from irn_amd.net import weights
    from oracle import cam_ref
    g = golden("nets")
    with torch.no_grad():
        y = cam_ref.cam_forward(weights.random_cam_state(seed=1), torch.from_numpy(g["cam_in"]))
    assert y.shape == g["cam_out"].shape and np.array_equal(y.numpy(), g["cam_out"])
    gflop = sum(cam_ref.flops_per_pair(s, s) for s in (512, 256, 768, 1024)) / 1e9
    assert abs(gflop - 974.04) < 0.01
    import bench
    assert abs(bench.backbone_gflop_per_image((1.0, 0.5, 1.5, 2.0), 512, False) - gflop) < 1e-6
    assert abs(bench.backbone_gflop_per_image((1.0, 0.5, 1.5, 2.0), 512, True) - gflop - 149.61) < 0.01
