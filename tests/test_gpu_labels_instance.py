"""GPU parity of the label epilogue and the instance front-end (through the C ABI): bit-exact
against the reference's outputs in tests/golden/ and against the oracle on fresh inputs."""
import numpy as np
import pytest
import torch

from oracle import irn_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _unpack(d, key):
    shape = tuple(d[key + "_shape"])
    return np.unpackbits(d[key])[:int(np.prod(shape))].reshape(shape).astype(bool)


def test_sem_seg_labels_bit_exact_vs_reference(golden):
    from irn_amd import ops
    wk, sg = golden("walk"), golden("semseg")
    names = sorted(k[:-6] for k in sg.files if k.endswith("_label"))
    rws = [torch.from_numpy(wk[n + "_rw"]).to(_dev()) for n in names]
    sizes = [tuple(int(v) for v in sg[n + "_size"]) for n in names]
    keys = [torch.from_numpy(sg[n + "_keys"]).to(_dev()) for n in names]
    # one bg threshold per call: run the two thresholds separately
    for bg in sorted({float(sg[n + "_bg"]) for n in names}):
        idx = [i for i, n in enumerate(names) if float(sg[n + "_bg"]) == bg]
        out = ops.label_epilogue([rws[i] for i in idx], [sizes[i] for i in idx], bg, keys=[keys[i] for i in idx],
                                 want_argmax=True, want_rw_up=True)
        for j, i in enumerate(idx):
            n = names[i]
            assert np.array_equal(out["rw_up"][j].cpu().numpy(), sg[n + "_rw_up"]), n
            assert np.array_equal(out["labels"][j].cpu().numpy(), sg[n + "_label"]), n


def test_epilogue_full_size_vs_oracle():
    from irn_amd import ops, synth
    rng = np.random.RandomState(5)
    rw = (synth.cam_blobs(4, 128, 128, seed=9) * rng.uniform(0.2, 1.0, (4, 1, 1))).astype(np.float32)[:, None]
    keys = np.array([1, 4, 9, 19])
    for size in ((512, 512), (500, 375), (509, 1)):
        out = ops.label_epilogue([torch.from_numpy(rw).to(_dev())], [size], 0.25,
                                 keys=[torch.from_numpy(keys).to(_dev())], want_argmax=True, want_rw_up=True)
        up, lab, idx = O.sem_seg_epilogue(rw, size, keys, 0.25)
        assert np.array_equal(out["rw_up"][0].cpu().numpy(), up)
        assert np.array_equal(out["labels"][0].cpu().numpy(), lab)
        assert np.array_equal(out["argmax"][0].cpu().numpy(), idx)


def test_argmax_tie_rule_first_maximum_wins():
    from irn_amd import ops
    rw = torch.zeros((3, 1, 4, 4), device=_dev())
    rw[0] = 0.25
    rw[1] = 1.0
    rw[2] = 1.0                                      # channels 1 and 2 tie at the max -> channel 1
    out = ops.label_epilogue([rw], [(16, 16)], 1.0, want_labels=False, want_argmax=True)
    assert int(out["argmax"][0].max()) == 0          # background (1.0) ties with the max -> background first
    out = ops.label_epilogue([rw], [(16, 16)], 0.5, want_labels=False, want_argmax=True)
    assert torch.all(out["argmax"][0] == 2)


@pytest.mark.parametrize("name", ["a", "b", "c", "cen64", "cen_ragged"])
def test_centroids_and_clusters_bit_exact_vs_reference(golden, name):
    from irn_amd import ops
    ins = golden("instance")
    dp = torch.from_numpy(ins[name + "_dp"]).to(_dev())
    cen = ops.find_centroids_with_refinement(dp)
    assert np.array_equal(cen.cpu().numpy(), ins[name + "_centroids"])
    oh = ops.cluster_centroids(cen, dp, as_one_hot=True)
    assert np.array_equal(oh.cpu().numpy(), _unpack(ins, name + "_instance_map"))


def test_centroids_128_vs_oracle():
    from irn_amd import ops, synth
    dp = synth.displacement_field(128, 128, seed=77, strength=0.3)
    cen = ops.find_centroids_with_refinement(torch.from_numpy(dp).to(_dev())).cpu().numpy()
    assert np.array_equal(cen, O.find_centroids_with_refinement(dp))


def test_label4_vs_oracle_random_masks():
    from irn_amd import ops
    rng = np.random.RandomState(1)
    for shape, p in (((3, 37, 41), 0.55), ((2, 64, 64), 0.62), ((1, 128, 128), 0.59), ((2, 9, 200), 0.7),
                     ((1, 16, 16), 0.0), ((1, 16, 16), 1.0)):
        m = rng.rand(*shape) < p
        labels, counts = ops.label4(torch.from_numpy(m).to(_dev()))
        labels, counts = labels.cpu().numpy(), counts.cpu().numpy()
        for i in range(shape[0]):
            ref = O.label4(m[i])
            assert np.array_equal(labels[i], ref), (shape, p, i)
            assert counts[i] == ref.max()
    # snake: one long component forces deep union-find chains
    m = np.zeros((1, 64, 64), bool)
    m[0, ::2] = True
    m[0, 1::4, -1] = True
    m[0, 3::4, 0] = True
    labels, counts = ops.label4(torch.from_numpy(m).to(_dev()))
    assert int(counts[0]) == 1 and np.array_equal(labels[0].cpu().numpy(), O.label4(m[0]))


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_instance_labels_end_to_end_vs_reference(golden, name):
    from irn_amd.misc import indexing
    from irn_amd.step import make_ins_seg_labels as mis
    ins = golden("instance")
    H, W = (int(v) for v in ins[name + "_size"])
    walker = indexing.RandomWalk(5, _dev())
    det = mis.instance_labels(walker, torch.from_numpy(ins[name + "_edge"])[None].to(_dev()),
                              torch.from_numpy(ins[name + "_dp"]).to(_dev()),
                              torch.from_numpy(ins[name + "_cam"]).to(_dev()),
                              torch.from_numpy(ins[name + "_keys"]), (H, W), 10.0, 8, 0.25)
    assert np.array_equal(det["mask"], _unpack(ins, name + "_det_mask"))
    assert np.array_equal(det["class"], ins[name + "_det_class"])
    assert np.abs(det["score"] - ins[name + "_det_score"]).max() <= 1e-4
    walker.close()


def test_detect_instance_on_device_vs_oracle_random_maps():
    """irn_detect_instance_* against the line-by-line restatement of step/make_ins_seg_labels.py:82-105
    on random class maps (many small components, empty channels, area filter on and off)."""
    from irn_amd import ops
    rng = np.random.RandomState(5)
    for (h, w, c, p_bg, thr) in ((37, 41, 5, 0.4, 0.0), (64, 80, 12, 0.2, 6.5), (128, 128, 3, 0.7, 163.84),
                                 (16, 16, 2, 0.0, 0.0), (33, 9, 40, 0.5, 2.0)):
        blocks = rng.randint(0, c + 1, size=((h + 3) // 4, (w + 3) // 4))
        blocks[:, 0] = np.where(blocks[:, 0] == 2, 0, blocks[:, 0])           # make channel 1 rare / empty at times
        cls = np.kron(blocks, np.ones((4, 4), int))[:h, :w]
        cls[rng.rand(h, w) < p_bg] = 0
        score = rng.rand(c, h, w).astype(np.float32)
        class_ids = np.repeat(np.arange(100, 100 + (c + 1) // 2), 2)[:c]
        one_hot = np.stack([cls == k + 1 for k in range(c)])
        if not one_hot.any():
            continue
        ref = O.detect_instance(score, one_hot, class_ids, max_fragment_size=thr)
        got = ops.detect_instance(torch.from_numpy(score).to(_dev()), torch.from_numpy(cls.astype(np.int32)).to(_dev()),
                                  class_ids, c, max_fragment_size=thr)
        assert got["mask"].shape == ref["mask"].shape, (h, w, c)
        assert np.array_equal(got["mask"], ref["mask"].astype(bool))
        assert np.array_equal(got["class"], ref["class"])
        assert np.array_equal(got["score"], np.asarray(ref["score"], np.float32))
    with pytest.raises(ValueError):
        ops.detect_instance(torch.zeros((2, 8, 8), device=_dev()), torch.zeros((8, 8), dtype=torch.int32, device=_dev()),
                            np.arange(2), 2)


def test_instance_labels_batch_equals_reference_golden(golden):
    """The three golden images through ONE batched walk / epilogue (irn_amd.step.make_ins_seg_labels.
    instance_labels_batch) reproduce the reference's detections exactly as the per-image path does."""
    from irn_amd.misc import indexing
    from irn_amd.step import make_ins_seg_labels as mis
    ins = golden("instance")
    items = []
    for name in "abc":
        H, W = (int(v) for v in ins[name + "_size"])
        items.append({"edge": torch.from_numpy(ins[name + "_edge"])[None].to(_dev()),
                      "dp": torch.from_numpy(ins[name + "_dp"]).to(_dev()),
                      "cam": torch.from_numpy(ins[name + "_cam"]).to(_dev()),
                      "keys": torch.from_numpy(ins[name + "_keys"]), "size": (H, W)})
    walker = indexing.RandomWalk(5, _dev())
    dets = mis.instance_labels_batch(walker, items, 10.0, 8, 0.25)
    for name, det in zip("abc", dets):
        assert not isinstance(det, Exception), det
        assert np.array_equal(det["mask"], _unpack(ins, name + "_det_mask"))
        assert np.array_equal(det["class"], ins[name + "_det_class"])
        assert np.abs(det["score"] - ins[name + "_det_score"]).max() <= 1e-4
    walker.close()


def test_batched_front_end_equals_single_image_calls(golden):
    """irn_find_centroids_batch / irn_cluster_centroids_batch on a ragged batch (golden images + fresh fields) give,
    image by image, what the single-image entry points give — and those are pinned on the reference above."""
    from irn_amd import ops, synth
    ins = golden("instance")
    dps = [ins[n + "_dp"] for n in ("a", "b", "c", "cen64", "cen_ragged")]
    dps += [synth.displacement_field(h, w, seed=s, strength=0.3) for h, w, s in ((128, 128, 5), (7, 9, 6), (94, 125, 7))]
    dev_dps = [torch.from_numpy(d).to(_dev()) for d in dps]
    cens = ops.find_centroids_batch(dev_dps)
    cmaps, ks = ops.cluster_centroids_batch(cens, dev_dps)
    for i, dp in enumerate(dev_dps):
        cen1 = ops.find_centroids_with_refinement(dp)
        assert torch.equal(cens[i], cen1), i
        cmap1, k1 = ops.cluster_centroids(cen1, dp)
        assert ks[i] == k1 and torch.equal(cmaps[i], cmap1), i
    for i, n in enumerate(("a", "b", "c", "cen64", "cen_ragged")):
        oh = cmaps[i][None] == torch.arange(ks[i], device=_dev(), dtype=torch.int32)[:, None, None]
        assert np.array_equal(oh.cpu().numpy(), _unpack(ins, n + "_instance_map"))


def test_detect_instance_batch_vs_oracle_incl_empty_and_fragmented():
    """irn_detect_instance_batch_*: a ragged batch with an all-background image in the middle and a salt-and-pepper
    class map with thousands of one-pixel fragments (the ranking of detections runs over several workgroups)."""
    from irn_amd import ops
    rng = np.random.RandomState(11)
    specs = [(37, 41, 5, 0.4, 0.0), (16, 16, 2, 1.0, 0.0), (96, 120, 7, 0.3, 0.0), (64, 80, 12, 0.2, 6.5)]
    scores, clss, cids, thrs = [], [], [], []
    for (h, w, c, p_bg, thr) in specs:
        if p_bg == 0.3:                                   # iid classes per pixel: ~4000 fragments
            cls = rng.randint(0, c + 1, size=(h, w))
        else:
            cls = np.kron(rng.randint(0, c + 1, size=((h + 3) // 4, (w + 3) // 4)), np.ones((4, 4), int))[:h, :w]
        cls[rng.rand(h, w) < p_bg] = 0
        scores.append(rng.rand(c, h, w).astype(np.float32))
        clss.append(cls.astype(np.int32))
        cids.append(np.arange(50, 50 + c))
        thrs.append(thr)
    got = ops.detect_instance_batch([torch.from_numpy(s).to(_dev()) for s in scores],
                                    [torch.from_numpy(c).to(_dev()) for c in clss], cids, [s[2] for s in specs], thrs)
    assert isinstance(got[1], ValueError)
    n_frag = 0
    for i in (0, 2, 3):
        c = specs[i][2]
        one_hot = np.stack([clss[i] == k + 1 for k in range(c)])
        ref = O.detect_instance(scores[i], one_hot, cids[i], max_fragment_size=thrs[i])
        assert got[i]["mask"].shape == ref["mask"].shape, i
        assert np.array_equal(got[i]["mask"], ref["mask"].astype(bool)), i
        assert np.array_equal(got[i]["class"], ref["class"]), i
        assert np.array_equal(got[i]["score"], np.asarray(ref["score"], np.float32)), i
        n_frag = max(n_frag, len(ref["score"]))
    assert n_frag > 2048
    # and the single-image form agrees on the fragmented map
    one = ops.detect_instance(torch.from_numpy(scores[2]).to(_dev()), torch.from_numpy(clss[2]).to(_dev()), cids[2],
                              specs[2][2], max_fragment_size=0.0)
    assert np.array_equal(one["mask"], got[2]["mask"]) and np.array_equal(one["score"], got[2]["score"])


def test_detect_instance_full_size_maps_crossing_many_tiles():
    """Full-size (512x512 and ragged) class maps whose components wind through many of the labelling's 64 x 16 LDS tiles:
    nested one-pixel rings with L-shaped spurs, interleaved combs of two classes, a checkerboard of 4-pixel blocks and one region covering the
    whole map — detections (masks, classes, scores, their ORDER = raster order of first pixels) equal the restatement of
    step/make_ins_seg_labels.py:82-105."""
    from irn_amd import ops
    rng = np.random.RandomState(23)

    def spiral(h, w):
        m = np.zeros((h, w), np.int32)
        y0, x0, y1, x1 = 0, 0, h - 1, w - 1
        k = 1
        while y0 <= y1 and x0 <= x1:
            m[y0, x0:x1 + 1] = k
            m[y0:y1 + 1, x1] = k
            if y1 > y0:
                m[y1, x0 + 2:x1 + 1] = k
            if x1 > x0 + 2:
                m[y0 + 2:y1 + 1, x0 + 2] = k
            y0, x0, y1, x1 = y0 + 2, x0 + 2, y1 - 2, x1 - 2
            k = 1 + (k % 3)
        return m

    def combs(h, w):
        m = np.zeros((h, w), np.int32)
        m[0, :] = 1
        m[h - 1, :] = 2
        m[1:h - 2, 0::4] = 1            # teeth hanging from the top bar
        m[2:h - 1, 2::4] = 2            # teeth standing on the bottom bar
        return m

    def blocks(h, w):
        b = rng.randint(0, 4, size=((h + 3) // 4, (w + 3) // 4))
        return np.kron(b, np.ones((4, 4), int))[:h, :w].astype(np.int32)

    maps = [spiral(512, 512), combs(512, 512), blocks(512, 512), np.full((375, 500), 2, np.int32), combs(333, 500), spiral(130, 67)]
    scores = [rng.rand(3, *m.shape).astype(np.float32) for m in maps]
    cids = [np.array([3, 7, 11])] * len(maps)
    got = ops.detect_instance_batch([torch.from_numpy(s).to(_dev()) for s in scores], [torch.from_numpy(m).to(_dev()) for m in maps],
                                    cids, [3] * len(maps), [0.0, 10.0, 0.0, 100.0, 0.0, 0.0])
    for i, m in enumerate(maps):
        one_hot = np.stack([m == k + 1 for k in range(3)])
        ref = O.detect_instance(scores[i], one_hot, cids[i], max_fragment_size=[0.0, 10.0, 0.0, 100.0, 0.0, 0.0][i])
        assert got[i]["mask"].shape == ref["mask"].shape, (i, got[i]["mask"].shape, ref["mask"].shape)
        assert np.array_equal(got[i]["mask"], ref["mask"].astype(bool)), i
        assert np.array_equal(got[i]["class"], ref["class"]), i
        assert np.array_equal(got[i]["score"], np.asarray(ref["score"], np.float32)), i


def test_deferred_detections_equal_blocking_and_survive_the_next_batch():
    """detect_instance_batch(deferred=True): the packed transfer of batch A is still in flight on the copy stream while
    batch B (different sizes, so different offsets in its own staging buffer) is counted, emitted and collected; A's
    result, collected last, equals the blocking call's.  An all-background batch gives the per-image ValueErrors."""
    from irn_amd import ops
    rng = np.random.RandomState(5)

    def batch(specs):
        sc, cl, cid = [], [], []
        for (h, w, c) in specs:
            cls = np.kron(rng.randint(0, c + 1, size=((h + 3) // 4, (w + 3) // 4)), np.ones((4, 4), int))[:h, :w]
            sc.append(torch.from_numpy(rng.rand(c, h, w).astype(np.float32)).to(_dev()))
            cl.append(torch.from_numpy(cls.astype(np.int32)).to(_dev()))
            cid.append(np.arange(7, 7 + c))
        return sc, cl, cid, [s[2] for s in specs], [0.0] * len(specs)

    A = batch([(64, 80, 3), (33, 47, 5), (128, 128, 2)])
    B = batch([(96, 96, 4), (16, 24, 1)])
    want_a, want_b = ops.detect_instance_batch(*A), ops.detect_instance_batch(*B)
    pa = ops.detect_instance_batch(*A, deferred=True)
    pb = ops.detect_instance_batch(*B, deferred=True)
    assert isinstance(pa, ops.PendingDetections)
    got_b, got_a = pb.result(), pa.result()
    assert pa.result() is got_a                                                  # collected once
    for got, want in ((got_a, want_a), (got_b, want_b)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(g["mask"], w["mask"]) and np.array_equal(g["score"], w["score"])
            assert np.array_equal(g["class"], w["class"])
    empty = ops.detect_instance_batch([torch.zeros(2, 8, 8, device=_dev())], [torch.zeros(8, 8, dtype=torch.int32, device=_dev())],
                                      [np.arange(2)], [2], [0.0], deferred=True).result()
    assert len(empty) == 1 and isinstance(empty[0], ValueError)


def test_instance_labels_batch_deferred_equals_blocking(golden):
    from irn_amd.misc import indexing
    from irn_amd.step import make_ins_seg_labels as mis
    ins = golden("instance")
    items = []
    for name in "abc":
        H, W = (int(v) for v in ins[name + "_size"])
        items.append({"edge": torch.from_numpy(ins[name + "_edge"])[None].to(_dev()),
                      "dp": torch.from_numpy(ins[name + "_dp"]).to(_dev()),
                      "cam": torch.from_numpy(ins[name + "_cam"]).to(_dev()),
                      "keys": torch.from_numpy(ins[name + "_keys"]), "size": (H, W)})
    walker = indexing.RandomWalk(5, _dev())
    want = mis.instance_labels_batch(walker, items, 10.0, 8, 0.25)
    p1 = mis.instance_labels_batch(walker, items, 10.0, 8, 0.25, deferred=True)
    p2 = mis.instance_labels_batch(walker, items[::-1], 10.0, 8, 0.25, deferred=True)   # enqueued before p1 is collected
    got, got_rev = p1.result(), p2.result()
    walker.close()
    for g1, g2, w in zip(got, got_rev[::-1], want):
        for d in (g1, g2):
            assert np.array_equal(d["mask"], w["mask"]) and np.array_equal(d["score"], w["score"]) and np.array_equal(d["class"], w["class"])
