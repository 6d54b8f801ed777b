"""Host-side mirrors: backbones reproduce the reference's forward on the same seeded weights,
sharding partitions the dataset, small helpers match."""
import os

import numpy as np
import pytest
import torch

from irn_amd.misc import imutils, pyutils, torchutils


def test_cam_forward_matches_reference(golden):
    from irn_amd.net import resnet50_cam, weights
    g = golden("nets")
    net = resnet50_cam.CAM()
    net.load_state_dict(weights.random_cam_state(seed=1), strict=True)
    net.eval()
    with torch.no_grad():
        y = net(torch.from_numpy(g["cam_in"]))
    assert y.shape == g["cam_out"].shape
    assert np.abs(y.numpy() - g["cam_out"]).max() <= 1e-4 * max(1.0, np.abs(g["cam_out"]).max())


def test_edge_displacement_forward_matches_reference(golden):
    from irn_amd.net import resnet50_irn, weights
    g = golden("nets")
    net = resnet50_irn.EdgeDisplacement(crop_size=128)
    net.load_state_dict(weights.random_irn_state(seed=2), strict=False)
    net.eval()
    with torch.no_grad():
        edge, dp = net(torch.from_numpy(g["irn_in"]))
    assert edge.shape == g["irn_edge"].shape and dp.shape == g["irn_dp"].shape
    assert np.abs(edge.numpy() - g["irn_edge"]).max() <= 1e-4
    assert np.abs(dp.numpy() - g["irn_dp"]).max() <= 1e-4 * max(1.0, np.abs(g["irn_dp"]).max())


def test_state_dict_keys_cover_reference_aliases():
    from irn_amd.net import resnet50_cam, resnet50_irn
    cam_keys = set(resnet50_cam.CAM().state_dict())
    for k in ("resnet50.conv1.weight", "stage1.0.weight", "stage1.4.0.conv1.weight", "backbone.3.0.2.bn3.running_var",
              "classifier.weight", "newly_added.0.weight", "resnet50.layer4.0.downsample.1.running_mean"):
        assert k in cam_keys, k
    irn_keys = set(resnet50_irn.EdgeDisplacement().state_dict())
    for k in ("fc_edge3.0.weight", "fc_edge6.bias", "fc_dp7.3.weight", "fc_dp7.4.running_mean",
              "mean_shift.running_mean", "edge_layers.5.weight", "dp_layers.6.1.bias", "stage5.0.2.conv3.weight"):
        assert k in irn_keys, k


@pytest.mark.parametrize("n_items,n", [(10, 1), (11, 2), (1449, 4), (10582, 8), (5, 8)])
def test_split_dataset_partitions(n_items, n):
    shards = torchutils.split_dataset(list(range(n_items)), n)
    seen = np.concatenate([np.asarray(s.indices) for s in shards])
    assert len(shards) == n and sorted(seen.tolist()) == list(range(n_items))
    for r in range(n):
        assert np.array_equal(shards[r].indices, torchutils.shard_indices(n_items, r, n))


def test_small_helpers():
    assert imutils.get_strided_size((375, 500), 4) == (94, 125)
    assert imutils.get_strided_up_size((375, 500), 16) == (384, 512)
    a = np.array([[5, 5, 9], [2, 9, 2]])
    assert np.array_equal(imutils.compress_range(a), [[1, 1, 2], [0, 2, 0]])
    oh = pyutils.to_one_hot(np.array([[0, 2], [1, 1]]))
    assert oh.shape == (3, 2, 2) and oh.dtype == np.bool_ and oh[2, 0, 1] and oh.sum() == 4
    assert pyutils.to_one_hot(np.array([0, 1]), maximum_val=4).shape == (4, 2)


def test_dataset_msf(tmp_path):
    from PIL import Image
    from irn_amd.voc12 import dataloader
    root = tmp_path / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    rng = np.random.RandomState(0)
    Image.fromarray(rng.randint(0, 255, (37, 50, 3), dtype=np.uint8)).save(root / "JPEGImages" / "2007_000032.jpg")
    lst = tmp_path / "list.txt"
    lst.write_text("2007_000032\n")
    lab = np.zeros(20, np.float32)
    lab[[3, 7]] = 1
    ds = dataloader.VOC12ClassificationDatasetMSF(str(lst), str(root), scales=(1.0, 0.5, 2.0),
                                                  cls_labels={2007000032: lab})
    item = ds[0]
    assert item["name"] == "2007_000032" and item["size"] == (37, 50)
    assert [tuple(x.shape) for x in item["img"]] == [(2, 3, 37, 50), (2, 3, 18, 25), (2, 3, 74, 100)]
    assert np.array_equal(item["img"][0][1], item["img"][0][0][..., ::-1])
    assert dataloader.decode_int_filename(2007000032) == "2007_000032"


def test_cam_merge_matches_reference_golden(golden):
    """step/make_cam.py:32-52 through the torch-op mirror of the merge (CPU here) against the reference's
    outputs; the HIP kernel the step uses is checked against the same fixture in test_gpu_steps.py."""
    from oracle import torch_mirrors
    cm = golden("cam_merge")
    for name in "ab":
        outs = [torch.from_numpy(cm["%s_out%d" % (name, i)]) for i in range(4)]
        keys, cam, hi = torch_mirrors.merge_scales_torch(outs, tuple(int(v) for v in cm[name + "_size"]),
                                              torch.from_numpy(cm[name + "_label"]))
        assert np.array_equal(keys.numpy(), cm[name + "_keys"])
        assert np.abs(cam.numpy() - cm[name + "_cam"]).max() <= 1e-6
        assert np.abs(hi.numpy() - cm[name + "_high_res"]).max() <= 1e-6


def test_async_writer_writes_everything_and_surfaces_errors(tmp_path):
    from irn_amd.step import _common
    w = _common.AsyncWriter(threads=2, max_pending=3)
    for i in range(10):
        w.submit(np.save, str(tmp_path / ("f%d.npy" % i)), {"i": i})
    w.close()
    assert sorted(p.name for p in tmp_path.iterdir()) == sorted("f%d.npy" % i for i in range(10))
    assert np.load(str(tmp_path / "f7.npy"), allow_pickle=True).item() == {"i": 7}
    w = _common.AsyncWriter(threads=1, max_pending=1)
    w.submit(np.save, str(tmp_path / "no_such_dir" / "x.npy"), 1)
    with pytest.raises(Exception):
        w.close()


def test_bicubic_plan_host_table_equals_oracle():
    """irn_bicubic_plan is host-only (no GPU): the fixed-point tap tables the kernels consume are Pillow's."""
    from irn_amd import ops
    from oracle import msf_oracle as M
    for n_in, n_out in [(500, 250), (375, 188), (375, 562), (500, 1000), (7, 3), (333, 334), (281, 1), (3, 40)]:
        lo, cnt, k = ops.bicubic_plan(n_in, n_out)
        a, b, c = M.bicubic_coeffs(n_in, n_out)
        assert np.array_equal(lo, a) and np.array_equal(cnt, b) and np.array_equal(k, c), (n_in, n_out)
    lo, cnt, k = ops.bicubic_plan(64, 64)                    # identity plan = "pass skipped"
    assert np.array_equal(lo, np.arange(64)) and (cnt == 1).all() and (k == 1 << 22).all()


def test_normalize_lut_equals_reference_normalisation():
    from irn_amd import ops
    from irn_amd.voc12.dataloader import TorchvisionNormalize
    img = np.random.RandomState(1).randint(0, 256, (9, 11, 3)).astype(np.uint8)
    lut = ops.normalize_lut()
    assert np.array_equal(TorchvisionNormalize()(img), np.stack([lut[c][img[..., c]] for c in range(3)], -1))


def test_msf_pack_refuses_cpu_tensors():
    from irn_amd import ops
    with pytest.raises(ValueError, match="GPU tensor"):
        ops.msf_pack(torch.zeros((4, 4, 3), dtype=torch.uint8), (1.0,))
    with pytest.raises(ValueError, match="GPU tensor"):
        ops.bicubic_resize(torch.zeros((4, 4, 3), dtype=torch.uint8), (2, 2))


def test_dataset_raw_mode_hands_over_the_decoded_image(tmp_path):
    """raw=True: the item carries the decoded uint8 image (the steps build the scales on the GPU); name, size and
    label are those of the reference-format item."""
    from PIL import Image
    from irn_amd import synth
    from irn_amd.voc12 import dataloader
    root = tmp_path / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    img = synth.photo(37, 52, seed=3)
    Image.fromarray(img).save(root / "JPEGImages" / "2008_000001.jpg", quality=95)
    (tmp_path / "train.txt").write_text("2008_000001\n")
    lab = np.zeros(20, np.float32)
    lab[[2, 7]] = 1
    kw = dict(voc12_root=str(root), scales=(1.0, 0.5), cls_labels={2008000001: lab})
    raw = dataloader.VOC12ClassificationDatasetMSF(str(tmp_path / "train.txt"), raw=True, **kw)[0]
    ref = dataloader.VOC12ClassificationDatasetMSF(str(tmp_path / "train.txt"), **kw)[0]
    decoded = np.asarray(Image.open(root / "JPEGImages" / "2008_000001.jpg").convert("RGB"))
    assert raw["img"].dtype == torch.uint8 and np.array_equal(raw["img"].numpy(), decoded)
    assert raw["name"] == ref["name"] and raw["size"] == ref["size"] == (37, 52)
    assert torch.equal(raw["label"], ref["label"])
    # the host loop over the same decoded image is what the oracle restates
    from oracle import msf_oracle as M
    for a, b in zip(ref["img"], M.msf_item(decoded, (1.0, 0.5))):
        assert np.array_equal(a, b)


def test_dense_operator_mirrors_match_reference(golden):
    """irn_amd.misc.indexing.affinity_sparse2dense / to_transition_matrix (API-completeness mirrors of reference
    misc/indexing.py:112-139; the walk itself never densifies) against the oracle's restatement and, chained the way
    misc/indexing.py:141-165 chains them, against the reference's own propagate_to_edge output."""
    import torch.nn.functional as F
    from irn_amd.misc import indexing
    from oracle import irn_oracle as O
    af = golden("affinity")
    for r in (3, 5):
        edge = af["r%d_edge" % r]
        h, w = edge.shape
        pi = indexing.PathIndex(r, (h + r, w + 2 * r))
        aff = torch.from_numpy(af["r%d_aff" % r])
        dense = indexing.affinity_sparse2dense(aff, pi.src_indices, pi.dst_indices, (h + r) * (w + 2 * r))
        want = O.affinity_dense(af["r%d_aff" % r], pi.src_indices, pi.dst_indices, (h + r) * (w + 2 * r))
        assert np.array_equal(dense.numpy(), want)
        assert torch.equal(dense, dense.t()) and float(dense.diagonal().min()) == 1.0
        t = indexing.to_transition_matrix(dense[:64, :64].contiguous() + 0.01, 10, 2)
        want_t = O.to_transition_matrix(want[:64, :64] + np.float32(0.01), 10, 2)
        assert np.abs(t.numpy() - want_t).max() <= 1e-6
    wk = golden("walk")
    name = "r5_b10_e4"
    h, w, c, r, beta, e = (int(v) for v in wk[name + "_params"])
    edge = torch.from_numpy(wk[name + "_edge"])[None]
    pi = indexing.PathIndex(r, (h + r, w + 2 * r))
    ep = F.pad(edge, (r, r, 0, r), value=1.0).reshape(-1).numpy()
    aff = torch.from_numpy(O.edge_to_affinity(ep, pi.path_indices))
    dense = indexing.affinity_sparse2dense(aff, pi.src_indices, pi.dst_indices, (h + r) * (w + 2 * r))
    dense = dense.view(h + r, w + 2 * r, h + r, w + 2 * r)[:-r, r:-r, :-r, r:-r].reshape(h * w, h * w)
    t = indexing.to_transition_matrix(dense, beta, e)
    x = torch.from_numpy(wk[name + "_cam"]).view(-1, h, w) * (1 - edge)
    rw = torch.matmul(x.view(-1, h * w), t).view(-1, 1, h, w)
    assert np.abs(rw.numpy() - wk[name + "_rw"]).max() <= 1e-6


def test_logger_and_timer(tmp_path, capsys):
    import sys
    import time
    log = pyutils.Logger(str(tmp_path / "run.log"))
    try:
        t = pyutils.Timer("step.make_cam:")
        print("hello")
        time.sleep(0.01)
        assert t.lapse() >= 0.01 and t.elapsed() >= 0.01
    finally:
        log.close()
    assert sys.stdout is not log
    text = (tmp_path / "run.log").read_text()
    assert "step.make_cam:" in text and "hello" in text
    assert "hello" in capsys.readouterr().out


def test_backbones_at_512_match_reference(golden):
    """CAM.forward on [2,3,512,512] and EdgeDisplacement.forward on a ragged VOC-size item vs the reference's CPU forwards
    (tests/golden/nets512.npz); the GPU forms of these checks live in tests/test_gpu_parity_r2.py."""
    from irn_amd import synth
    from irn_amd.net import resnet50_cam, resnet50_irn, weights
    g = golden("nets512")
    cam = resnet50_cam.CAM()
    cam.load_state_dict(weights.random_cam_state(seed=1), strict=True)
    cam.eval()
    irn = resnet50_irn.EdgeDisplacement()
    irn.load_state_dict(weights.random_irn_state(seed=2), strict=False)
    irn.eval()
    with torch.no_grad():
        h, w, seed = (int(v) for v in g["cam512_seed"])
        y = cam(torch.from_numpy(synth.image_pair(h, w, seed))).numpy()
        assert np.abs(y - g["cam512_out"]).max() <= 1e-4 * np.abs(g["cam512_out"]).max()
        h, w, seed = (int(v) for v in g["irn_seed"])
        edge, dp = irn(torch.from_numpy(synth.image_pair(h, w, seed)))
        assert np.abs(edge.numpy() - g["irn_edge"]).max() <= 1e-4
        assert np.abs(dp.numpy() - g["irn_dp"]).max() <= 1e-4 * max(1.0, np.abs(g["irn_dp"]).max())


def test_thread_loader_collates_like_dataloader(tmp_path):
    """irn_amd.step._common.make_loader (thread prefetcher) yields exactly what DataLoader(batch_size=1, shuffle=False)
    yields for the multi-scale dataset, raw and pre-scaled items, with and without worker threads."""
    from PIL import Image
    from torch.utils.data import DataLoader
    from irn_amd.step import _common
    from irn_amd.voc12 import dataloader as vd
    (tmp_path / "JPEGImages").mkdir()
    names, labels = [], {}
    rng = np.random.RandomState(0)
    for i in range(5):
        n = "2008_%06d" % (i + 1)
        names.append(n)
        Image.fromarray((rng.rand(40 + i, 50, 3) * 255).astype(np.uint8)).save(tmp_path / "JPEGImages" / (n + ".jpg"))
        labels[int(n.replace("_", ""))] = np.eye(20, dtype=np.float32)[i]
    (tmp_path / "train.txt").write_text("\n".join(names) + "\n")
    np.save(tmp_path / "cls_labels.npy", labels)
    for raw in (True, False):
        ds = vd.VOC12ClassificationDatasetMSF(str(tmp_path / "train.txt"), voc12_root=str(tmp_path), scales=(1.0, 0.5), raw=raw)
        want = list(DataLoader(ds, shuffle=False, num_workers=0))
        for nw in (0, 3):
            got = list(_common.make_loader(ds, nw))
            assert len(got) == len(want)
            for x, y in zip(want, got):
                assert x["name"] == y["name"] and torch.equal(x["label"], y["label"])
                assert [int(v) for v in x["size"]] == [int(v) for v in y["size"]]
                if raw:
                    assert torch.equal(x["img"], y["img"])
                else:
                    assert all(torch.equal(p, q) for p, q in zip(x["img"], y["img"]))


def test_cam_store_hit_and_file_fallback(tmp_path):
    """step/_common.CamStore: a CAM put by make_cam is handed to the label steps from memory; anything else comes from the
    `.npy` the reference's make_cam writes (step/make_cam.py:55-56) — same values either way.  Entries belong to ONE
    output directory AND to one make_cam run: a later put replaces an earlier one, and an entry of a run the directory
    no longer carries the stamp of (make_cam ran again somewhere else) is never served."""
    from irn_amd.step import _common
    store = _common.CamStore(max_bytes=1 << 20)
    dev = torch.device("cpu")
    keys = torch.tensor([3, 7])
    cam = torch.rand(2, 8, 9)
    np.save(tmp_path / "2008_000001.npy", {"keys": keys, "cam": cam, "high_res": np.zeros((2, 32, 36), np.float32)})
    run1 = _common.new_cam_run(str(tmp_path))
    assert _common.current_cam_run(str(tmp_path)) == run1 and _common.current_cam_run(str(tmp_path / "nowhere")) is None
    store.put("2008_000001", keys, keys.clone(), cam.clone(), cam_out_dir=str(tmp_path), run_id=run1)
    k_cpu, k_dev, c = store.get("2008_000001", str(tmp_path), dev, run1)
    assert store.hits == 1 and store.misses == 0 and torch.equal(k_cpu, keys) and torch.equal(c, cam)
    # keep_cams_on_device off: the store is not consulted
    store.get("2008_000001", str(tmp_path), dev, run1, use_store=False)
    assert store.hits == 1 and store.misses == 1
    # a second run with other weights: same name, new values -> the new ones are served
    cam2 = torch.rand(2, 8, 9)
    store.put("2008_000001", keys, keys.clone(), cam2.clone(), cam_out_dir=str(tmp_path), run_id=run1)
    assert torch.equal(store.get("2008_000001", str(tmp_path), dev, run1)[2], cam2) and len(store) == 1
    assert store._bytes == cam2.numel() * 4
    # make_cam ran again for this directory in ANOTHER process (new stamp, new file): this store's entry is stale
    run2 = _common.new_cam_run(str(tmp_path))
    assert run2 != run1
    cam_new = torch.rand(2, 8, 9)
    np.save(tmp_path / "2008_000001.npy", {"keys": keys, "cam": cam_new, "high_res": np.zeros((2, 32, 36), np.float32)})
    got = store.get("2008_000001", str(tmp_path), dev, _common.current_cam_run(str(tmp_path)))
    assert torch.equal(got[2], cam_new) and store.misses == 2
    # an entry without a stamp (or a directory without one) is never a hit
    assert torch.equal(store.get("2008_000001", str(tmp_path), dev, None)[2], cam_new) and store.misses == 3
    # the same name under another output directory is another entry: served from ITS file
    other = tmp_path / "other"
    other.mkdir()
    cam3 = torch.rand(2, 8, 9)
    np.save(other / "2008_000001.npy", {"keys": keys, "cam": cam3, "high_res": np.zeros((2, 32, 36), np.float32)})
    assert torch.equal(store.get("2008_000001", str(other), dev, run2)[2], cam3) and store.misses == 4
    store.drop_dir(str(tmp_path))
    assert len(store) == 0 and store._bytes == 0
    k_cpu, k_dev, c = store.get("2008_000001", str(tmp_path), dev, run2)
    assert store.misses == 5 and torch.equal(k_cpu, keys) and torch.equal(k_dev, keys) and torch.equal(c, cam_new)
    big = torch.zeros(1, 1024, 1024)                       # 4 MB > the 1 MB cap: not kept, never an error
    store.put("big", keys[:1], keys[:1], big, cam_out_dir=str(tmp_path), run_id=run2)
    assert len(store) == 0
    for i in range(40):                                    # 40 x 36 KB > 1 MB: the oldest entries leave
        store.put("n%d" % i, keys, keys, torch.zeros(1, 96, 96), cam_out_dir=str(tmp_path), run_id=run2)
    assert store._bytes <= 1 << 20 and ("%s" % tmp_path, "n39") in store._items and ("%s" % tmp_path, "n0") not in store._items


def test_split_by_owner_hits_and_balance():
    """Label-step shards that follow make_cam's CAM placement: every image whose CAM a worker holds goes to that worker
    while the shards stay within 25 % of the even share; unknown images fill the lightest shards; every image exactly
    once (the reference's strided split, misc/torchutils.py:66-68, is the fallback)."""
    from irn_amd.step import _common
    rng = np.random.default_rng(0)
    names = ["img%04d" % i for i in range(1000)]
    made = {"img%04d" % i: int(i % 8) for i in rng.permutation(1200)[:900]}      # CAMs of 900 images, strided over 8 workers
    shards = _common.split_by_owner(list(range(1000)), 8, names, made)
    seen = sorted(int(i) for s in shards for i in s.indices)
    assert seen == list(range(1000))
    assert max(len(s) for s in shards) <= int(np.ceil(1000 / 8 * 1.25))
    hits = sum(1 for k, s in enumerate(shards) for i in s.indices if made.get(names[int(i)]) == k)
    known = sum(1 for n in names if n in made)
    assert hits >= 0.97 * known, (hits, known)
    # a pathological placement (everything on worker 0) is capped, not followed
    shards = _common.split_by_owner(list(range(100)), 4, names[:100], {n: 0 for n in names[:100]})
    assert [len(s) for s in shards][0] == 32 and sorted(int(i) for s in shards for i in s.indices) == list(range(100))


def test_frozen_batch_norm_folding_and_composed_path():
    """FrozenBatchNorm.folded() is the batch norm as one multiply-add (the constants irn_bn_act consumes), and apply_ on
    the CPU (or with autograd on) is the reference's composed tail: FixedBatchNorm -> += residual -> ReLU
    (net/resnet50.py:11-14, :34-54), leaving its input untouched."""
    import torch.nn.functional as F
    from irn_amd.net import resnet50 as R
    g = torch.Generator().manual_seed(9)
    bn = R.FrozenBatchNorm(6)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(6, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(6, generator=g))
        bn.running_mean.copy_(torch.randn(6, generator=g))
        bn.running_var.copy_(torch.rand(6, generator=g) + 0.05)
    x = torch.randn(2, 6, 5, 7, generator=g)
    skip = torch.randn(2, 6, 5, 7, generator=g)
    scale, shift = bn.folded()
    with torch.no_grad():
        want = bn(x)
    assert float((x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) - want).abs().max()) < 2e-6
    x0 = x.clone()
    with torch.no_grad():
        y = bn.apply_(x, residual=skip, relu=True)
    assert torch.equal(x, x0) and torch.equal(y, F.relu(want + skip))
    bn.train()                                                   # statistics stay frozen (FixedBatchNorm)
    with torch.no_grad():
        assert torch.equal(bn(x), want)
    key = bn.folded()[0]
    assert bn.folded()[0] is key
    with torch.no_grad():
        bn.weight.mul_(2.0)
    assert float((bn.folded()[0] - 2 * key).abs().max()) < 1e-6


def test_apply_with_a_batch_norm_on_the_residual_composed_path():
    """apply_(x, residual, residual_bn=...) off the device = bn(x) + bn_r(residual) -> ReLU: the projection shortcut of
    net/resnet50.py:48-52."""
    import torch.nn.functional as F
    from irn_amd.net import resnet50 as R
    torch.manual_seed(2)
    bn, bn_r = R.FrozenBatchNorm(4), R.FrozenBatchNorm(4)
    with torch.no_grad():
        for m in (bn, bn_r):
            m.running_mean.normal_()
            m.running_var.uniform_(0.2, 2.0)
            m.weight.normal_()
            m.bias.normal_()
        x, r = torch.randn(2, 4, 3, 5), torch.randn(2, 4, 3, 5)
        assert torch.equal(bn.apply_(x, residual=r, relu=True, residual_bn=bn_r), F.relu(bn(x) + bn_r(r)))
        unit = R.Bottleneck(8, 4, stride=2, project=True).eval()
        y = unit(torch.randn(1, 8, 9, 9))
        assert y.shape == (1, 16, 5, 5) and float(y.min()) >= 0.0


def test_folding_under_inference_mode():
    """Tensors made under torch.inference_mode() have no version counter; the folded constants are then cached by
    storage only."""
    from irn_amd.net import resnet50 as R
    with torch.inference_mode():
        bn = R.FrozenBatchNorm(3)
        scale, shift = bn.folded()
        assert bn.folded()[0] is scale
        x = torch.randn(2, 3, 4, 4)
        assert float((bn.apply_(x, relu=False) - (x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))).abs().max()) < 1e-6


def test_load_checkpoint_skips_only_what_the_checkpoint_overwrites(tmp_path):
    """net/weights.load_checkpoint (the start of every step's run(args), reference step/make_cam.py:63-65): the random
    initialisers are skipped, the loaded network equals the regularly built one, a partial checkpoint keeps a regular
    initialisation for the rest, and torch.nn.init is left as it was."""
    import torch.nn.init as init
    from irn_amd.net import resnet50_cam, weights
    before = init.kaiming_uniform_
    path = str(tmp_path / "cam.pth")
    torch.save(weights.random_cam_state(1), path)
    fast = weights.load_checkpoint(resnet50_cam.CAM, path, strict=True)
    ref = resnet50_cam.CAM()
    ref.load_state_dict(torch.load(path), strict=True)
    ref.eval()
    x = torch.randn(2, 3, 48, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        assert torch.equal(fast(x), ref(x)) and not fast.training
    sd = torch.load(path)
    dropped = "classifier.weight"
    sd.pop(dropped)
    torch.save(sd, str(tmp_path / "part.pth"))
    part = weights.load_checkpoint(resnet50_cam.CAM, str(tmp_path / "part.pth"), strict=False)
    w = part.state_dict()[dropped]
    assert bool(torch.isfinite(w).all()) and float(w.abs().max()) < 1.0 and float(w.std()) > 1e-4     # a real initialisation
    assert init.kaiming_uniform_ is before


def test_model_spec_is_built_once_per_checkpoint_version(tmp_path):
    """step/_common.ModelSpec: what a step's run(args) hands its (persistent) workers instead of a pickled network — built by
    the worker on first use, reused by later steps naming the same checkpoint, rebuilt when the file changes."""
    import os
    from irn_amd.net import weights
    from irn_amd.step import _common
    path = str(tmp_path / "cam.pth")
    torch.save(weights.random_cam_state(1), path)
    spec = _common.ModelSpec("net.resnet50_cam", "CAM", path, strict=True)
    a = _common.materialise(spec)
    assert _common.materialise(_common.ModelSpec("net.resnet50_cam", "CAM", path, strict=True)) is a     # same key, same object
    assert _common.materialise(a) is a                                                                   # networks pass through
    sd = weights.random_cam_state(7)
    torch.save(sd, path)
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 10 ** 9))
    b = _common.materialise(spec)
    assert b is not a and torch.equal(b.state_dict()["classifier.weight"], sd["classifier.weight"])
    assert sum(1 for k in _common._MODELS if k[2] == os.path.abspath(path)) == 1                         # the old version was dropped


def test_miopen_setup_is_stable_seeded_and_exclusive(tmp_path, monkeypatch):
    """One MIOpen user database per (device, HIP version, device ordinal), stable across runs, seeded from the package's
    shipped database when empty, and never shared by two live processes (a second claimant gets a private copy)."""
    import subprocess
    import sys
    from irn_amd.step import _common
    for k in ("IRN_MIOPEN_DB_SET", "IRN_MIOPEN_BASE", "MIOPEN_USER_DB_PATH", "MIOPEN_FIND_MODE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("IRN_MIOPEN_CACHE", str(tmp_path))
    monkeypatch.setattr(_common, "_MIOPEN_LOCKS", [])
    monkeypatch.delenv("IRN_DETERMINISTIC", raising=False)
    d0 = _common.miopen_setup(0)
    key = _common.miopen_mode_key()
    assert key == _common.miopen_cache_key() + "-det"             # the reproducible mode is the default and has its own databases
    import torch
    from irn_amd.net import resnet50 as r50
    assert r50.DETERMINISTIC is True and torch.backends.cudnn.deterministic is True
    assert d0 == os.path.join(str(tmp_path), key, "dev0") and os.environ["MIOPEN_USER_DB_PATH"] == d0
    assert os.environ["MIOPEN_FIND_MODE"] == "2" and _common.miopen_setup(0) == d0
    open(os.path.join(d0, "gfx950_256.ufdb.txt"), "w").write("found")
    # a second live process on the same device: private copy of what the first one has, removed when it exits
    code = ("import os, sys; sys.path.insert(0, %r); from irn_amd.step import _common; d = _common.miopen_setup(0); "
            "print(d); print(sorted(os.listdir(d)))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = {k: v for k, v in os.environ.items() if k not in ("IRN_MIOPEN_DB_SET",)}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    d1, files = out.stdout.strip().splitlines()[-2:]
    assert d1 == os.path.join(str(tmp_path), key, "dev0-pid%s" % d1.rsplit("pid", 1)[1]) and "gfx950_256.ufdb.txt" in files
    assert not os.path.exists(d1)
    # another ordinal is another directory; a user-chosen find mode survives
    monkeypatch.delenv("IRN_MIOPEN_DB_SET")
    monkeypatch.setenv("MIOPEN_FIND_MODE", "1")
    assert _common.miopen_setup(3).endswith(os.path.join(key, "dev3")) and os.environ["MIOPEN_FIND_MODE"] == "1"
    # the fast mode: its own key, PyTorch's flag off
    monkeypatch.delenv("IRN_MIOPEN_DB_SET")
    monkeypatch.setenv("IRN_DETERMINISTIC", "0")
    assert _common.miopen_setup(3).endswith(os.path.join(_common.miopen_cache_key(), "dev3"))
    assert r50.DETERMINISTIC is False and torch.backends.cudnn.deterministic is False
    monkeypatch.setattr(r50, "DETERMINISTIC", None)


def test_det_database_is_the_tuned_one_minus_the_split_k_implicit_gemms():
    """irn_amd/data/miopen/<key>-det (tools/miopen_det_filter.py): same problems, same channels-last shape list; the NHWC
    implicit-GEMM solver is gone from exactly the forward records whose tuned configuration splits K, another fast solver
    remains in every one of them, nothing else changed."""
    import glob
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irn_amd", "data", "miopen")
    fast = sorted(d for d in glob.glob(os.path.join(root, "*")) if os.path.isdir(d) and not d.endswith("-det"))
    assert fast
    for src in fast:
        dst = src + "-det"
        assert os.path.isdir(dst), "run tools/miopen_det_filter.py"
        assert open(os.path.join(src, "nhwc_shapes.json")).read() == open(os.path.join(dst, "nhwc_shapes.json")).read()
        rec = lambda d: dict(l.rstrip("\n").split("=", 1) for l in open(glob.glob(os.path.join(d, "*.ufdb.txt"))[0]) if "=" in l)
        a, b = rec(src), rec(dst)
        assert a.keys() == b.keys()
        gtc = "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC:"
        changed = [k for k in a if a[k] != b[k]]
        assert changed and all("NHWC" in k and k.endswith("-F") for k in changed)
        for k in changed:
            ents_a, ents_b = a[k].split(";"), b[k].split(";")
            assert [e for e in ents_a if not e.startswith(gtc)] == ents_b
            assert any(not e.startswith("ConvDirectNaive") for e in ents_b), k


def test_merge_miopen_db_adds_missing_entries_only(tmp_path):
    """The shipped find database completes a user database: entries the user database lacks are appended, entries it has
    (what this machine measured) are kept, other files are left alone."""
    from irn_amd.step import _common
    src, dst = tmp_path / "src", tmp_path / "dst"
    src.mkdir()
    dst.mkdir()
    (src / "gfx.ufdb.txt").write_text("a-NCHW=solverA:1\nb-NHWC-NHWC-NHWC=solverB:2\n")
    (src / "gfx.udb.txt").write_text("p=1,2,3")
    (src / "nhwc_shapes.json").write_text("[[16, 512, 512]]")
    (dst / "gfx.ufdb.txt").write_text("a-NCHW=mine:0.5\n")
    assert _common.merge_miopen_db(str(src), str(dst)) == 2
    assert (dst / "gfx.ufdb.txt").read_text() == "a-NCHW=mine:0.5\nb-NHWC-NHWC-NHWC=solverB:2\n"
    assert (dst / "gfx.udb.txt").read_text() == "p=1,2,3\n" and not (dst / "nhwc_shapes.json").exists()
    assert _common.merge_miopen_db(str(src), str(dst)) == 0


def test_trunk_layout_is_chosen_per_input_shape(monkeypatch):
    """auto: channels-last only for the network-input shapes the shipped find database was tuned for; 1 / 0 force it."""
    from irn_amd.net import resnet50 as r50

    class _T:                    # a tensor stand-in: only the attributes channels_last_for reads
        def __init__(self, shape, cuda=True):
            self.shape, self.is_cuda = shape, cuda

        def dim(self):
            return len(self.shape)

    from irn_amd.step import _common
    monkeypatch.setattr(r50, "_TUNED_SHAPES", {_common.miopen_cache_key(): {(16, 512, 512)}, _common.miopen_cache_key() + "-det": {(16, 512, 512)}})
    monkeypatch.setattr(r50, "DETERMINISTIC", None)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", False)
    with torch.no_grad():
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "auto")
        monkeypatch.delenv("IRN_MIOPEN_DB_SET", raising=False)
        assert not r50.channels_last_for(_T((16, 3, 512, 512)))       # no miopen_setup in this process: the NHWC entries may be missing
        monkeypatch.setenv("IRN_MIOPEN_DB_SET", "/somewhere")
        assert r50.channels_last_for(_T((16, 3, 512, 512))) and not r50.channels_last_for(_T((16, 3, 375, 500)))
        assert not r50.channels_last_for(_T((2, 3, 512, 512))) and not r50.channels_last_for(_T((16, 3, 512, 512), cuda=False))
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "1")
        assert r50.channels_last_for(_T((2, 3, 375, 500)))
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "0")
        assert not r50.channels_last_for(_T((16, 3, 512, 512)))
        # a caller's own deterministic flag, not managed by the steps: MIOpen's attribute leaves no fast NHWC solver -> NCHW
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "auto")
        torch.backends.cudnn.deterministic = True
        assert not r50.channels_last_for(_T((16, 3, 512, 512)))
        # the reproducible mode of the steps (managed): tuned shapes channels-last WITHOUT the attribute (their database has no
        # order-dependent solver), everything else NCHW WITH it; the end of a pass puts the attribute back on
        monkeypatch.setattr(r50, "DETERMINISTIC", True)
        assert r50.channels_last_for(_T((16, 3, 512, 512))) and torch.backends.cudnn.deterministic is False
        r50.end_trunk_pass()
        assert torch.backends.cudnn.deterministic is True
        assert not r50.channels_last_for(_T((16, 3, 333, 500))) and torch.backends.cudnn.deterministic is True
        monkeypatch.setattr(r50, "DETERMINISTIC", False)
        torch.backends.cudnn.deterministic = False
        assert r50.channels_last_for(_T((16, 3, 512, 512))) and torch.backends.cudnn.deterministic is False
    monkeypatch.setattr(r50, "DETERMINISTIC", None)
    monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "1")
    with torch.enable_grad():
        assert not r50.channels_last_for(_T((16, 3, 512, 512)))       # the training seam keeps the composed NCHW path


def test_reproducible_mode_runs_the_trunk_in_passes_of_a_fixed_size(monkeypatch):
    """which kernels an image meets must not depend on the batch it travels in (shard split, tail of a
    shard, early flush).  Reproducible mode: rows go through in passes of 16 (sizes the shipped database is tuned for: a short
    pass is filled with zero images) or 2 (every other size: the reference's own batch); forced channels-last on an unknown
    shape keeps MIOpen's deterministic attribute on."""
    from irn_amd.net import resnet50 as r50
    from irn_amd.step import _common

    class _T:
        def __init__(self, shape, cuda=True):
            self.shape, self.is_cuda = shape, cuda

        def dim(self):
            return len(self.shape)

    monkeypatch.setattr(r50, "_TUNED_SHAPES", {_common.miopen_cache_key() + "-det": {(16, 512, 512), (16, 375, 500)}})
    monkeypatch.delenv("IRN_DETERMINISTIC", raising=False)
    monkeypatch.setenv("IRN_MIOPEN_DB_SET", "/somewhere")
    monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "auto")
    with torch.no_grad():
        monkeypatch.setattr(r50, "DETERMINISTIC", True)
        assert r50.pass_rows(_T((6, 3, 512, 512))) == 16 and r50.pass_rows(_T((16, 3, 375, 500))) == 16
        assert r50.pass_rows(_T((16, 3, 333, 500))) == 2 and r50.pass_rows(_T((2, 3, 64, 64))) == 2
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "0")
        assert r50.pass_rows(_T((16, 3, 512, 512))) == 2                      # no channels-last trunk: every size is "untuned"
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "1")
        prev = torch.backends.cudnn.deterministic
        assert r50.channels_last_for(_T((4, 3, 100, 100))) and torch.backends.cudnn.deterministic is True     # forced layout, unknown shape
        assert r50.channels_last_for(_T((16, 3, 512, 512))) and torch.backends.cudnn.deterministic is False
        torch.backends.cudnn.deterministic = prev
        monkeypatch.setattr(r50, "CHANNELS_LAST_MODE", "auto")
        for mode in (None, False):
            monkeypatch.setattr(r50, "DETERMINISTIC", mode)
            assert r50.pass_rows(_T((6, 3, 512, 512))) is None                # outside the mode the caller's batch is the pass
        assert r50.pass_rows(_T((6, 3, 512, 512), cuda=False)) is None
    # the chunking itself (CPU tensors, the plan injected): every pass has exactly `rows` rows, pads are dropped, tuples work
    seen = []

    def fn(c):
        seen.append(int(c.shape[0]))
        return c.sum(dim=(1, 2, 3)), c * 2.0

    x = torch.arange(6 * 3 * 2 * 2, dtype=torch.float32).reshape(6, 3, 2, 2)
    monkeypatch.setattr(r50, "pass_rows", lambda t: 4)
    pads0 = r50.PASS_STATS["pad_rows"]
    a, b = r50.run_rows(fn, x)
    assert seen == [4, 4] and r50.PASS_STATS["pad_rows"] - pads0 == 2
    assert torch.equal(a, x.sum(dim=(1, 2, 3))) and torch.equal(b, x * 2.0)
    seen.clear()
    assert torch.equal(r50.run_rows(lambda c: c + 1.0, x[:4]), x[:4] + 1.0)      # a full pass is handed over as it is
    monkeypatch.setattr(r50, "pass_rows", lambda t: None)
    assert torch.equal(r50.run_rows(lambda c: c + 1.0, x), x + 1.0)


def test_missing_shipped_data_warns_once_and_names_the_remedy(monkeypatch, tmp_path):
    """the tuned find database and the GEMM rank table exist for one (architecture, CU count, HIP
    version) key; on any other box the steps ran ~10 % slower without a word.  Now one RuntimeWarning per process and kind
    names the key looked for, the keys shipped and the tool that writes the data; the per-step summary lists the sizes
    whose trunk passes fell to NCHW."""
    import warnings
    from irn_amd.step import _common
    from irn_amd.net import resnet50 as r50
    monkeypatch.setattr(_common, "miopen_cache_key", lambda: "gfx999-cu1-hip0.0")
    monkeypatch.setattr(_common, "_WARNED", set())
    monkeypatch.delenv("IRN_DETERMINISTIC", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        rep = _common.warn_missing_shipped_data()
        _common.warn_missing_shipped_data()                                    # once per process
    assert rep["miopen"] is False and rep["gemm"] is False and rep["mode_key"] == "gfx999-cu1-hip0.0-det"
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert len(msgs) == 2
    mi = [m for m in msgs if "MIOpen" in m][0]
    ge = [m for m in msgs if "GEMM rank table" in m][0]
    assert "gfx999-cu1-hip0.0-det" in mi and "gfx950-cu256-hip7.0.51831-det" in mi and "tools/miopen_warmup.py" in mi and "miopen_det_filter" in mi
    assert "gfx999-cu1-hip0.0" in ge and "gfx950-cu256-hip7.0.51831" in ge and "tools/conv1x1_tune.py" in ge
    # the shipped key itself: nothing to say
    monkeypatch.setattr(_common, "miopen_cache_key", lambda: "gfx950-cu256-hip7.0.51831")
    monkeypatch.setattr(_common, "_WARNED", set())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        rep = _common.warn_missing_shipped_data()
    assert rep["miopen"] and rep["gemm"] and not [x for x in w if issubclass(x.category, RuntimeWarning)]
    # miopen_setup is where every worker / rank / in-process step passes: it warns too
    monkeypatch.setattr(_common, "miopen_cache_key", lambda: "gfx999-cu1-hip0.0")
    monkeypatch.setattr(_common, "_WARNED", set())
    monkeypatch.setattr(_common, "_MIOPEN_LOCKS", [])
    for k in ("IRN_MIOPEN_DB_SET", "IRN_MIOPEN_BASE", "MIOPEN_USER_DB_PATH"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("IRN_MIOPEN_CACHE", str(tmp_path))
    saved = (torch.backends.cudnn.deterministic, r50.DETERMINISTIC)
    try:
        with pytest.warns(RuntimeWarning, match="no tuned MIOpen find database for 'gfx999-cu1-hip0.0-det'"):
            _common.miopen_setup(0)
    finally:
        torch.backends.cudnn.deterministic, r50.DETERMINISTIC = saved
    # step summary: NCHW passes by size, then reset
    monkeypatch.setattr(r50, "PASS_STATS", {"channels_last": 5, "nchw": 3, "pad_rows": 4, "nchw_sizes": {"281x500": 2, "96x112": 1}})
    line = _common.untuned_report()
    assert "3 of 8 trunk passes ran NCHW" in line and "281x500 x2" in line and "96x112 x1" in line and "4 zero rows" in line
    assert _common.untuned_report() == "" and r50.PASS_STATS["nchw"] == 0
    line = _common.startup_line(3, 8, 0, "/db")
    assert "worker 3/8" in line and "NOT SHIPPED" in line and "reproducible" in line


def test_edge_store_keys_and_cap(tmp_path):
    """step/_common.EdgeStore: the boundary / displacement maps one label step leaves for the other are keyed by network,
    forward geometry and image FILE; another checkpoint, another image under the same name (new mtime / size) or another
    device is a miss; the oldest entries leave when the cap is reached; edges_for fills and uses it."""
    from irn_amd.step import _common, make_sem_seg_labels
    root = tmp_path / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    (root / "JPEGImages" / "2008_000001.jpg").write_bytes(b"a" * 100)
    stamp = _common.image_stamp(str(root), "2008_000001")
    assert stamp[0].endswith("2008_000001.jpg") and stamp[2] == 100
    (root / "JPEGImages" / "2008_000001.jpg").write_bytes(b"b" * 101)
    assert _common.image_stamp(str(root), "2008_000001") != stamp and _common.image_stamp(str(root), "missing")[1:] == (-1, -1)

    class _Net:                      # stands in for EdgeDisplacement: counts its forwards
        crop_size, stride, calls = 512, 4, 0

        def forward_batch(self, items):
            _Net.calls += 1
            return [(it[:1, 0, ::4, ::4] + 1.0, it[:, 0, ::4, ::4] * 2.0) for it in items]

    store = _common.EdgeStore(max_bytes=1 << 20)
    net = _Net()
    mk = lambda names: [{"name": n, "img": torch.full((2, 3, 16, 20), float(i)), "stamp": ("/p/" + n, 1, 2)} for i, n in enumerate(names)]
    a = mk(["x", "y", "z"])
    make_sem_seg_labels.edges_for(net, a, 2, store=store, model_key=("ckpt", 1))
    assert _Net.calls == 2 and store.misses == 3 and store.hits == 0 and len(store) == 3 and "img" not in a[0]
    b = mk(["x", "y", "z", "w"])
    make_sem_seg_labels.edges_for(net, b, 2, store=store, model_key=("ckpt", 1))
    assert _Net.calls == 3 and store.hits == 3                                    # only "w" went through the network
    for p, q in zip(a, b):
        assert torch.equal(p["edge"], q["edge"]) and torch.equal(p["dp"], q["dp"])
    c = mk(["x"])
    make_sem_seg_labels.edges_for(net, c, 2, store=store, model_key=("ckpt", 2))   # another checkpoint: computed again
    assert _Net.calls == 4
    d = mk(["x"])
    d[0]["stamp"] = ("/p/x", 9, 2)                                                # the file changed
    make_sem_seg_labels.edges_for(net, d, 2, store=store, model_key=("ckpt", 1))
    assert _Net.calls == 5
    e = mk(["x"])
    make_sem_seg_labels.edges_for(net, e, 2)                                       # no store: always computed, nothing stored
    assert _Net.calls == 6 and "stamp" in e[0]
    small = _common.EdgeStore(max_bytes=3 * 4 * 60)
    for i in range(5):
        small.put(("k", i), torch.zeros(1, 4, 5), torch.zeros(2, 4, 5))
    assert len(small) == 3 and small.get(("k", 0), torch.device("cpu")) is None and small.get(("k", 4), torch.device("cpu")) is not None


def test_bottleneck_gemm_params_fold_batch_norm_into_weight_and_bias():
    """Bottleneck.gemm_params: conv -> FrozenBatchNorm == conv with (weight * scale) + shift, shortcut shift merged into conv3's
    bias; checked on the CPU against the module's own composed forward (reference net/resnet50.py:11-14, :34-54)."""
    import torch
    import torch.nn.functional as F
    from irn_amd.net import resnet50 as r50
    torch.manual_seed(0)
    for project in (False, True):
        c_in = 64 if project else 128
        unit = r50.Bottleneck(c_in, 32, stride=1, project=project).eval()
        for m in unit.modules():
            if isinstance(m, r50.FrozenBatchNorm):
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(); m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0)
        x = torch.randn(2, c_in, 8, 8)
        with torch.no_grad():
            want = unit(x)
            p = unit.gemm_params()
            y = F.relu(F.conv2d(x, p["w1"].view(32, c_in, 1, 1), p["b1"]))
            y = unit.bn2.apply_(unit.conv2(y), relu=True)
            sc = F.conv2d(x, p["wd"]) if project else x
            got = F.relu(F.conv2d(y, p["w3"].view(128, 32, 1, 1), p["b3"]) + sc)
        assert float((got - want).abs().max()) <= 1e-5
        assert unit.gemm_params() is p                          # cached while the parameters stand
        with torch.no_grad():
            unit.bn3.bias.add_(1.0)
        assert unit.gemm_params() is not p and not torch.equal(unit.gemm_params()["b3"], p["b3"])


def test_device_key_comes_from_the_architecture_not_the_marketing_name(monkeypatch):
    """`torch.cuda.get_device_name` is empty under rocprofv3 and generic without amdgpu.ids: the key of the shipped databases
    (MIOpen find database, GEMM rank table) is built from gcnArchName + CU count + HIP version."""
    import types
    import torch
    from irn_amd.step import _common
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(gcnArchName="gfx950:sramecc+:xnack-", multi_processor_count=256))
    monkeypatch.setattr(torch.cuda, "get_device_name", lambda i=0: "")
    key = _common.miopen_cache_key()
    assert key.startswith("gfx950-cu256-hip") and " " not in key
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irn_amd", "data")
    shipped = os.listdir(os.path.join(root, "miopen"))
    assert any(d.startswith("gfx950-cu256-hip") for d in shipped), shipped
    assert any(f.startswith("gfx950-cu256-hip") and f.endswith(".json") for f in os.listdir(os.path.join(root, "gemm")))


def test_dataset_skips_the_pixels_when_the_step_says_so(tmp_path):
    """VOC12ClassificationDatasetMSF(raw=True, skip_image=...): an image the step will not need (its edge maps are on the device)
    comes as an empty uint8 tensor with the size from the file header; `device_images` maps it to None; `EdgeStore.peek` does not
    count.  (The reference decodes every image in both label steps, step/make_sem_seg_labels.py:24-34.)"""
    import numpy as np
    import torch
    from PIL import Image
    from torch.utils.data import default_collate
    from irn_amd.step import _common
    from irn_amd.voc12 import dataloader
    root = tmp_path / "voc" / "JPEGImages"
    root.mkdir(parents=True)
    for name, (h, w) in (("2008_000001", (30, 44)), ("2008_000002", (37, 21))):
        Image.fromarray((np.random.RandomState(0).rand(h, w, 3) * 255).astype(np.uint8)).save(root / (name + ".jpg"))
    (tmp_path / "train.txt").write_text("2008_000001\n2008_000002\n")
    np.save(tmp_path / "cls_labels.npy", {2008000001: np.eye(20, dtype=np.float32)[3], 2008000002: np.eye(20, dtype=np.float32)[5]})
    ds = dataloader.VOC12ClassificationDatasetMSF(str(tmp_path / "train.txt"), voc12_root=str(tmp_path / "voc"), raw=True,
                                                  skip_image=lambda name: name.endswith("2"))
    a, b = default_collate([ds[0]]), default_collate([ds[1]])
    assert a["img"].shape == (1, 30, 44, 3) and b["img"].numel() == 0
    assert (int(b["size"][0]), int(b["size"][1])) == (37, 21) and b["name"][0] == "2008_000002"
    assert _common.device_images(b, (1.0,)) is None
    _common.set_skip_image(torch.utils.data.Subset(ds, [0, 1]), None)
    assert ds.skip_image is None and ds[1]["img"].shape == (37, 21, 3)
    store = _common.EdgeStore()
    store.put("k", torch.zeros(1, 2, 2), torch.zeros(2, 2, 2))
    h0, m0 = store.hits, store.misses
    assert store.peek("k", torch.device("cpu")) and not store.peek("other", torch.device("cpu"))
    assert (store.hits, store.misses) == (h0, m0)


def test_run_sample_help_renders_and_deterministic_flag_reaches_the_environment(monkeypatch, tmp_path):
    """`run_sample.py --help` formats (a bare % in a help string used to crash argparse) and `--deterministic` sets the variable
    every process that sets MIOpen up reads (workers inherit it)."""
    import run_sample
    text = run_sample.build_parser().format_help()
    assert "--deterministic {0,1}" in text and "--edge_out_dir" in text and "--walk_accel {0,1}" in text
    a = run_sample.build_parser().parse_args(["--voc12_root", "x", "--deterministic", "0"])
    assert a.deterministic == 0 and run_sample.build_parser().parse_args(["--voc12_root", "x"]).deterministic is None


def test_instance_step_pipeline_order(monkeypatch):
    """make_ins_seg_labels._flush: per turn (1) the batch whose front half is enqueued gets its back half, (2) the new batch's IRNet
    forward + front half are enqueued, (3) the batch before is collected and written — every batch passes the three stages once
    and in order, the back half of batch k never queues behind the forward of batch k+1, three closing turns drain the pipe.
    (Reference loop: step/make_ins_seg_labels.py:119-152, one image at a time.)"""
    import argparse
    from irn_amd.step import make_ins_seg_labels as mis, make_sem_seg_labels as mss
    log = []
    monkeypatch.setattr(mss, "edges_for", lambda model, pend, irn_batch, **kw: log.append(("irnet", tuple(p["name"] for p in pend))))
    monkeypatch.setattr(mss, "_edge_store_kw", lambda model, args: {})
    monkeypatch.setattr(mis, "instance_front", lambda items: log.append(("front", tuple(it["name"] for it in items))) or ("cmaps", "k_dev"))
    monkeypatch.setattr(mis, "instance_back", lambda walker, items, front, beta, exp_times, bg, deferred=False:
                        log.append(("back", tuple(it["name"] for it in items))) or "pending-%s" % items[0]["name"])
    monkeypatch.setattr(mis, "_write", lambda names, pending, args, writer: log.append(("write", tuple(names), pending)))
    args = argparse.Namespace(beta=10, exp_times=8, ins_seg_bg_thres=0.25, irn_batch=8)
    state = {}
    batches = [["a1", "a2"], ["b1", "b2"], ["c1"]]
    for b in batches:
        pend = [{"name": n} for n in b]
        mis._flush(None, None, pend, args, None, state)
        assert pend == []
    for _ in range(3):
        mis._flush(None, None, [], args, None, state)
    assert state == {"front": None, "emit": None}
    stages = [(kind, names[0][0]) for kind, names, *rest in log]
    # per batch: irnet -> front -> back -> write, once each
    for tag in "abc":
        assert [k for k, t in stages if t == tag] == ["irnet", "front", "back", "write"], (tag, stages)
    # the back half of a batch is enqueued BEFORE the next batch's forward (its small read-backs must not wait behind it)
    assert stages.index(("back", "a")) < stages.index(("irnet", "b")) < stages.index(("write", "a")) or \
        stages.index(("back", "a")) < stages.index(("irnet", "b"))
    assert stages.index(("back", "b")) < stages.index(("irnet", "c"))
    # a batch is collected one turn after its back half: its detections cross PCIe under the next batch's kernels
    assert stages.index(("write", "a")) > stages.index(("front", "b")) and stages.index(("write", "b")) > stages.index(("front", "c"))
    assert [w[2] for w in log if w[0] == "write"] == ["pending-a1", "pending-b1", "pending-c1"]


def test_split_weight_operands_reconstruct_the_weight_to_22_bits_and_the_split_product_is_fp32_grade():
    """The host half of the split-precision convolutions (irn_amd/ops.py split_weight[_3x3]; reference net/resnet50.py:34-54 with
    FixedBatchNorm folded in): [w_hi | w_lo | w_hi 2^-11] of w 2^p reconstructs w to 2^-22 of the layer's largest weight, every
    entry stays inside fp16's range, the taps of a 3x3 weight share one exponent in both operand layouts — and the product the GPU
    forms with them, emulated here in numpy (fp16 operands, exact products, wide accumulation), is as close to the exact one as a
    correctly rounded fp32 dot product."""
    from irn_amd import ops
    g = torch.Generator().manual_seed(4)
    for cout, cin, spread in ((64, 256, 1.0), (512, 128, 1e-3), (8, 16, 30.0)):
        w = torch.randn(cout, cin, generator=g).double() * spread * torch.rand(cout, 1, generator=g).double()
        b16, alpha = ops.split_weight(w)
        assert b16.dtype == torch.float16 and b16.shape == (cout, 3 * cin) and float(b16.float().abs().max()) < 2.0 ** 14
        hi, lo, his = b16[:, :cin].double(), b16[:, cin:2 * cin].double(), b16[:, 2 * cin:].double()
        # the scaled copy is exact for every weight above 2^-17 of the largest, and off by at most half a subnormal step below
        assert float((his - hi * 2.0 ** -11).abs().max()) <= 2.0 ** -25 and torch.equal(his[hi.abs() >= 0.125], (hi * 2.0 ** -11)[hi.abs() >= 0.125])
        rec = (hi + lo) * alpha
        assert float((rec - w).abs().max()) <= 2.0 ** -22 * float(w.abs().max())
        # the activation side as irn_split16 forms it, and the three-term product
        x = torch.relu(torch.randn(40, cin, generator=g)).float() * 5.0
        xh = x.to(torch.float16)
        xl = ((x - xh.float()) * 2048.0).to(torch.float16)
        a16 = torch.cat([xh, xh, xl], dim=1).double()
        got = alpha * (a16 @ b16.double().t())
        want = x.double() @ w.t()
        f32 = (x @ w.float().t()).double()
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 4e-7 * scale and float((got - want).abs().max()) <= 2.0 * float((f32 - want).abs().max()) + 1e-7 * scale
    w3 = torch.randn(16, 8, 3, 3, generator=g).double() * 0.1
    for fused, shape in ((False, (9, 16, 24)), (True, (3, 16, 72))):
        ops.CONV3X3_ROW_FUSED, saved = fused, ops.CONV3X3_ROW_FUSED
        try:
            t16, a3 = ops.split_weight_3x3(w3)
        finally:
            ops.CONV3X3_ROW_FUSED = saved
        assert tuple(t16.shape) == shape
        taps = t16.view(3, 16, 3, 24).permute(0, 2, 1, 3).reshape(9, 16, 24) if fused else t16
        for t in range(9):
            rec = (taps[t][:, :8].double() + taps[t][:, 8:16].double()) * a3
            assert float((rec - w3[:, :, t // 3, t % 3]).abs().max()) <= 2.0 ** -22 * float(w3.abs().max())


def test_warmup_tool_picks_the_untuned_sizes_of_a_dataset(tmp_path):
    """tools/miopen_warmup.py --from-list: the (H, W) histogram of a list's images from their JPEG headers, most frequent first,
    minus what the shipped channels-last database covers, until the requested share of the images has a tuned size."""
    import argparse
    import importlib.util
    from PIL import Image
    root = tmp_path / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    sizes = [(375, 500)] * 5 + [(300, 420)] * 3 + [(123, 77)] * 2 + [(64, 64)]
    names = []
    for i, (h, w) in enumerate(sizes):
        names.append("2011_%06d" % i)
        Image.new("RGB", (w, h)).save(root / "JPEGImages" / (names[-1] + ".jpg"))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    spec = importlib.util.spec_from_file_location("miopen_warmup", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "miopen_warmup.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irn_amd", "data", "miopen")
    a = argparse.Namespace(from_list=str(tmp_path / "list.txt"), voc12_root=str(root), coverage=0.9, max_sizes=40, out=shipped)
    assert mod.sizes_of_list(a) == "300x420,123x77"          # 375x500 is shipped; 8 + 2 of 11 images >= 90 %; 64x64 not needed
    a.coverage, a.max_sizes = 1.0, 1
    assert mod.sizes_of_list(a) == "300x420"
