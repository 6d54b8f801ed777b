"""The run_sample.py step API end to end on a synthetic VOC-shaped directory: make_cam ->
make_sem_seg_labels -> make_ins_seg_labels with random-init checkpoints; checks file schemas through
the reference's readers' access patterns and the step results against the operator tier."""
import argparse
import os

import numpy as np
import pytest
import torch
from PIL import Image

from _parity import label_mismatches

pytestmark = pytest.mark.gpu


os.environ.setdefault("MIOPEN_FIND_MODE", "2")     # fast find: the backbones are plumbing here, not the subject


def _make_voc(tmp, n=4):
    root = tmp / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    rng = np.random.RandomState(0)
    names, labels = [], {}
    for i in range(n):
        name = "2008_%06d" % (i + 1)
        h, w = ((96, 128), (113, 150))[i % 2]            # two sizes only: every new conv shape costs a MIOpen find
        img = (rng.rand(h // 8 + 1, w // 8 + 1, 3) * 255).astype(np.uint8)
        Image.fromarray(img).resize((w, h), Image.BICUBIC).save(root / "JPEGImages" / (name + ".jpg"), quality=95)
        names.append(name)
        lab = np.zeros(20, np.float32)
        lab[rng.choice(20, rng.randint(1, 4), replace=False)] = 1
        labels[int(name.replace("_", ""))] = lab
    (tmp / "lists").mkdir()
    (tmp / "lists" / "train.txt").write_text("\n".join(names) + "\n")
    np.save(tmp / "lists" / "cls_labels.npy", labels)
    return root, names, labels


class _CaptureEdges:
    """Records the edge / displacement maps the label steps actually computed (by wrapping make_sem_seg_labels.edges_for,
    which both label steps call), so that the oracle can be run on exactly the inputs the HIP path saw — a second
    forward of the backbone in THIS process need not be bit-identical (it has run unmanaged convolutions before: LESSONS.md 37)."""

    def __init__(self, module):
        self.module, self.orig, self.edges, self.dps = module, module.edges_for, {}, {}

    def __enter__(self):
        def wrapped(model, pend, irn_batch, **kw):
            self.orig(model, pend, irn_batch, **kw)
            for p in pend:
                self.edges[p["name"]] = p["edge"][0].cpu().numpy()
                self.dps[p["name"]] = p["dp"].cpu().numpy()
        self.module.edges_for = wrapped
        return self

    def __exit__(self, *exc):
        self.module.edges_for = self.orig


def test_steps_end_to_end(tmp_path):
    from irn_amd.net import weights
    from irn_amd.step import make_cam, make_ins_seg_labels, make_sem_seg_labels
    root, names, labels = _make_voc(tmp_path)
    torch.save(weights.random_cam_state(1), tmp_path / "res50_cam.pth")
    torch.save(weights.random_irn_state(2), tmp_path / "res50_irn.pth")
    args = argparse.Namespace(
        num_workers=0, voc12_root=str(root), train_list=str(tmp_path / "lists" / "train.txt"),
        infer_list=str(tmp_path / "lists" / "train.txt"), cam_network="net.resnet50_cam",
        cam_weights_name=str(tmp_path / "res50_cam"), cam_scales=(1.0, 0.5, 1.5),
        irn_network="net.resnet50_irn", irn_weights_name=str(tmp_path / "res50_irn.pth"),
        beta=10, exp_times=8, sem_seg_bg_thres=0.25, ins_seg_bg_thres=0.25,
        cam_out_dir=str(tmp_path / "cam"), sem_seg_out_dir=str(tmp_path / "sem"),
        ins_seg_out_dir=str(tmp_path / "ins"), walk_batch=3)
    os.makedirs(args.cam_out_dir)
    os.makedirs(args.sem_seg_out_dir)
    os.makedirs(args.ins_seg_out_dir)

    make_cam.run(args)
    for n in names:
        d = np.load(os.path.join(args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        W, H = Image.open(root / "JPEGImages" / (n + ".jpg")).size
        k = int(labels[int(n.replace("_", ""))].sum())
        assert d["keys"].dtype == torch.int64 and d["keys"].shape == (k,)
        assert isinstance(d["cam"], torch.Tensor) and d["cam"].shape == (k, (H - 1) // 4 + 1, (W - 1) // 4 + 1)
        assert isinstance(d["high_res"], np.ndarray) and d["high_res"].shape == (k, H, W)
        assert float(d["cam"].max()) <= 1.0 + 1e-6 and float(d["cam"].min()) >= 0.0
        # the reference's readers: keys + 1 padded (make_sem_seg_labels.py:37), cam.cuda() (:39)
        assert np.pad(d["keys"] + 1, (1, 0), mode="constant")[0] == 0

    # default = inputs built on the GPU (irn_msf_pack); the reference's PIL loop in the loader workers (device_preprocess off)
    # gives the SAME FILES, bit for bit: the inputs are bit-identical (tests/test_gpu_msf.py) and the default mode's backbones
    # are a function of their inputs.  Both runs go through a fresh worker process (always_use_workers): this pytest process has
    # run the same convolution shapes unmanaged before, and MIOpen keeps the solver it resolved for a problem per process.
    from irn_amd.step import _common as _c
    try:
        w_args = argparse.Namespace(**{**vars(args), "cam_out_dir": str(tmp_path / "cam_gpu_w"), "worker_devices": "0", "always_use_workers": True})
        ref_args = argparse.Namespace(**{**vars(w_args), "cam_out_dir": str(tmp_path / "cam_pil_w"), "device_preprocess": False})
        for a_ in (w_args, ref_args):
            os.makedirs(a_.cam_out_dir)
            make_cam.run(a_)
    finally:
        _c.shutdown_workers()
    for n in names:
        a = np.load(os.path.join(w_args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        b = np.load(os.path.join(ref_args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        assert torch.equal(a["keys"], b["keys"])
        assert torch.equal(a["cam"], b["cam"]) and np.array_equal(a["high_res"], b["high_res"]), n
        # and the in-process run above (a process with a history) stays inside the parity bar of SURVEY.md §8(d)
        c = np.load(os.path.join(args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        assert (a["cam"] - c["cam"]).abs().max().item() <= 1e-4 and np.abs(a["high_res"] - c["high_res"]).max() <= 1e-4

    from irn_amd.step import _common
    hits0, misses0 = _common.CAM_STORE.hits, _common.CAM_STORE.misses
    _common.EDGE_STORE.clear()
    e_hits0, e_misses0 = _common.EDGE_STORE.hits, _common.EDGE_STORE.misses
    with _CaptureEdges(make_sem_seg_labels) as cap:
        make_sem_seg_labels.run(args)
    edges = cap.edges
    assert _common.CAM_STORE.hits - hits0 == len(names) and _common.CAM_STORE.misses == misses0   # CAMs came from device memory
    # first label step of the run: every edge map computed and left on the device for the other label step
    assert _common.EDGE_STORE.misses - e_misses0 == len(names) and _common.EDGE_STORE.hits == e_hits0 and len(_common.EDGE_STORE) == len(names)
    from oracle import build_oracle, irn_oracle as O
    olib = build_oracle.load()
    for n in names:
        png = np.asarray(Image.open(os.path.join(args.sem_seg_out_dir, n + ".png")))
        W, H = Image.open(root / "JPEGImages" / (n + ".jpg")).size
        assert png.dtype == np.uint8 and png.shape == (H, W)
        present = set(np.nonzero(labels[int(n.replace("_", ""))])[0] + 1) | {0}
        assert set(np.unique(png)) <= present
        # VALUES: the oracle's walk + epilogue (step/make_sem_seg_labels.py:36-49) on the same edge map and the CAM file
        d = np.load(os.path.join(args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        rw = build_oracle.walk(olib, d["cam"].numpy(), edges[n], 5, 10, 256)
        up, want, _ = O.sem_seg_epilogue(rw, (H, W), d["keys"].numpy(), 0.25)
        n_diff, gap = label_mismatches(png, want, up, 0.25, lut=np.concatenate([[0], d["keys"].numpy() + 1]), what=n)
        print("%s: %d of %d label pixels differ from the oracle (ties, largest top-2 gap %.2e)" % (n, n_diff, png.size, gap))

    # the same step reading the CAM files (no device hand-off) and with the walk radius of BASELINE configs[2]
    file_args = argparse.Namespace(**{**vars(args), "sem_seg_out_dir": str(tmp_path / "sem_files")})
    os.makedirs(file_args.sem_seg_out_dir)
    _common.CAM_STORE.clear()
    make_sem_seg_labels.run(file_args)
    assert _common.CAM_STORE.misses - misses0 == len(names)
    assert _common.EDGE_STORE.hits - e_hits0 == len(names)             # ... and this run took its edge maps from the store: no IRNet forward
    # the hand-off off: the maps are recomputed and equal the stored ones up to MIOpen's run-to-run solver choice
    off_args = argparse.Namespace(**{**vars(args), "sem_seg_out_dir": str(tmp_path / "sem_nostore"), "keep_edges_on_device": False})
    os.makedirs(off_args.sem_seg_out_dir)
    e_total = _common.EDGE_STORE.hits + _common.EDGE_STORE.misses
    with _CaptureEdges(make_sem_seg_labels) as cap_off:
        make_sem_seg_labels.run(off_args)
    assert _common.EDGE_STORE.hits + _common.EDGE_STORE.misses == e_total
    worst = max(float(np.abs(cap_off.edges[n] - edges[n]).max()) for n in names)
    print("edge maps recomputed vs handed over in device memory: max deviation %.2e" % worst)
    assert worst <= 1e-4
    for n in names:
        a = np.asarray(Image.open(os.path.join(args.sem_seg_out_dir, n + ".png")))
        b = np.asarray(Image.open(os.path.join(file_args.sem_seg_out_dir, n + ".png")))
        assert np.array_equal(a, b), n
    r10_args = argparse.Namespace(**{**vars(args), "sem_seg_out_dir": str(tmp_path / "sem_r10"), "radius": 10})
    os.makedirs(r10_args.sem_seg_out_dir)
    with _CaptureEdges(make_sem_seg_labels) as cap10:
        make_sem_seg_labels.run(r10_args)
    n = names[0]
    W, H = Image.open(root / "JPEGImages" / (n + ".jpg")).size
    d = np.load(os.path.join(args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
    rw = build_oracle.walk(olib, d["cam"].numpy(), cap10.edges[n], 10, 10, 256)
    up, want, _ = O.sem_seg_epilogue(rw, (H, W), d["keys"].numpy(), 0.25)
    png = np.asarray(Image.open(os.path.join(r10_args.sem_seg_out_dir, n + ".png")))
    n_diff, gap = label_mismatches(png, want, up, 0.25, lut=np.concatenate([[0], d["keys"].numpy() + 1]), what=n + " radius 10")
    print("%s radius 10: %d of %d label pixels differ from the oracle (largest top-2 gap %.2e)" % (n, n_diff, png.size, gap))

    e_hits1 = _common.EDGE_STORE.hits
    with _CaptureEdges(make_sem_seg_labels) as capi:
        make_ins_seg_labels.run(args)
    assert _common.EDGE_STORE.hits - e_hits1 == len(names)             # boundary AND displacement maps of the semantic step reused
    written = [n for n in names if os.path.exists(os.path.join(args.ins_seg_out_dir, n + ".npy"))]
    assert written, "no instance file written"
    for n in written:
        d = np.load(os.path.join(args.ins_seg_out_dir, n + ".npy"), allow_pickle=True).item()
        W, H = Image.open(root / "JPEGImages" / (n + ".jpg")).size
        assert set(d) == {"score", "mask", "class"}
        assert d["mask"].dtype == np.bool_ and d["mask"].shape[1:] == (H, W)
        assert len(d["score"]) == len(d["mask"]) == len(d["class"])
        # masks of one image never overlap (they come from an argmax)
        assert d["mask"].sum(0).max() <= 1
        # VALUES: the oracle's instance pipeline (step/make_ins_seg_labels.py:131-150) on the same edge / dp / CAM
        cd = np.load(os.path.join(args.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        walk = lambda x, e, radius, beta, exp_times: build_oracle.walk(olib, x, e, radius, beta, 2 ** exp_times)
        _, inst, rw_i, _, want = O.instance_labels(cd["cam"].numpy(), cd["keys"].numpy(), capi.edges[n], capi.dps[n], (H, W), walk=walk)
        # the class-id map the detections paint (masks are disjoint): identical up to argmax ties between the fp32 walk and
        # the fp64 oracle — every differing pixel must be such a tie between channels (class x instance) of the oracle's
        # score stack; when no tie moved a fragment, detections agree one by one
        paint = lambda det: (np.asarray(det["mask"]).astype(np.int64) * (np.asarray(det["class"], np.int64) + 1)[:, None, None]).sum(0)
        up_i, _, _ = O.sem_seg_epilogue(rw_i, (H, W), np.zeros(rw_i.shape[0], np.int64), 0.25)
        chan_class = np.concatenate([[0], np.repeat(cd["keys"].numpy(), inst.shape[0]) + 1])
        n_diff, gap = label_mismatches(paint(d), paint(want), up_i, 0.25, lut=chan_class, what=n + " instance classes")
        print("%s: %d of %d instance-class pixels differ from the oracle (largest top-2 gap %.2e)" % (n, n_diff, H * W, gap))
        if len(want["score"]) == len(d["score"]):
            assert np.array_equal(np.asarray(want["class"]), d["class"]), n
            # masks: every pixel at which any detection's mask differs is a tie (< 1e-4) between the two best entries of the
            # oracle's score stack (background included) — proven per pixel, no allowance on the count
            moved = (np.asarray(want["mask"]).astype(bool) != d["mask"]).any(0)
            if moved.any():
                stack = np.sort(np.concatenate([np.full((1, H, W), 0.25, np.float32), up_i], 0)[:, moved], 0)
                assert float((stack[-1] - stack[-2]).max()) < 1e-4, (n, int(moved.sum()), float((stack[-1] - stack[-2]).max()))
            assert np.abs(np.asarray(want["score"], np.float32) - d["score"]).max() <= 1e-3, n


@pytest.mark.parametrize("mode", ["deterministic", "fast"])
def test_steps_two_worker_processes_on_one_device(tmp_path, monkeypatch, mode):
    """The N > 1 path of the steps on a one-GPU box: `worker_devices="0,0"` = two persistent worker processes sharing
    GPU 0 (reference: one process per GPU, step/make_cam.py:71-74).  Exercises spawn, model pickling, HIP + MIOpen
    start-up in the children, the CAM hand-off in EACH worker's device memory across steps (CAM-owner aware shards), two
    resident (cooperative, all-CU) walks contending for one GPU — whichever loses its bounded wait is re-run on the
    streaming sweeps — and checks every output file of the two-worker layout against the one-worker layout:
      deterministic  (IRN_DETERMINISTIC=1, the default: channels-last trunk on the database without split-K solvers for tuned
                     shapes, MIOpen's deterministic attribute + NCHW for everything else — these small images) — the two layouts
                     write the SAME bits: CAMs, edge / displacement maps, label maps, detections.  (Both layouts run in fresh
                     worker processes: MIOpen keeps the solvers it resolved for a problem per process, whatever the attribute
                     says later, so a process that ran the same convolution shapes unmanaged before — this pytest process —
                     is not a deterministic one.)
      fast           (IRN_DETERMINISTIC=0: tuned channels-last trunk with split-K solvers) — the CAMs agree to fp32 rounding; the boundary /
                     displacement maps move by ~1e-5 too, and the instance clustering is discontinuous in them, so the two
                     runs are not compared with each other: EACH run's labels and instance classes are checked against the
                     oracle (fp64 walk + the reference's epilogue / clustering) on that run's OWN CAM files and edge maps
                     (`edge_out_dir`), every differing pixel proven a < 1e-4 tie of the oracle's score stack.
    No allowance on any pixel count."""
    from irn_amd.net import weights
    from irn_amd.step import _common, make_cam, make_ins_seg_labels, make_sem_seg_labels
    from oracle import build_oracle, irn_oracle as O
    monkeypatch.setenv("IRN_DETERMINISTIC", "1" if mode == "deterministic" else "0")      # spawned workers inherit it
    root, names, labels = _make_voc(tmp_path, n=6)
    torch.save(weights.random_cam_state(1), tmp_path / "res50_cam.pth")
    torch.save(weights.random_irn_state(2), tmp_path / "res50_irn.pth")

    def make_args(tag, **kw):
        a = argparse.Namespace(
            num_workers=2, voc12_root=str(root), train_list=str(tmp_path / "lists" / "train.txt"),
            infer_list=str(tmp_path / "lists" / "train.txt"), cam_network="net.resnet50_cam",
            cam_weights_name=str(tmp_path / "res50_cam"), cam_scales=(1.0, 0.5),
            irn_network="net.resnet50_irn", irn_weights_name=str(tmp_path / "res50_irn.pth"),
            beta=10, exp_times=8, sem_seg_bg_thres=0.25, ins_seg_bg_thres=0.25, radius=10,
            cam_out_dir=str(tmp_path / (tag + "_cam")), sem_seg_out_dir=str(tmp_path / (tag + "_sem")),
            ins_seg_out_dir=str(tmp_path / (tag + "_ins")), edge_out_dir=str(tmp_path / (tag + "_edge")),
            walk_batch=2, cam_batch=1, irn_batch=1, **kw)
        for d in (a.cam_out_dir, a.sem_seg_out_dir, a.ins_seg_out_dir):
            os.makedirs(d)
        return a

    def run_layout(args):
        try:
            make_cam.run(args)
            make_sem_seg_labels.run(args)
            stats_sem = _common.pool_stats()
            make_ins_seg_labels.run(args)
            stats = _common.pool_stats()
            pool = _common._POOL[0]
            assert pool is not None and pool.alive() and len(pool.devices) == len(stats)
        finally:
            _common.shutdown_workers()
        return stats_sem, stats

    two = make_args("two", worker_devices="0,0")
    stats_sem, stats = run_layout(two)
    assert len(stats) == 2
    hits = sum(s["cam_store_hits"] for s in stats)
    misses = sum(s["cam_store_misses"] for s in stats)
    print("two workers on device 0 (%s): CAM hand-offs in device memory %d, from files %d; walk batches re-run on the streaming "
          "sweeps: %s" % (mode, hits, misses, [s["walk_fallback_runs"] for s in stats]))
    assert sum(s["cam_store_hits"] for s in stats_sem) == len(names)      # every CAM was found in its worker's memory
    assert hits == 2 * len(names) and misses == 0

    one = make_args("one", worker_devices="0", always_use_workers=True)   # ONE worker process: the other layout
    _, stats1 = run_layout(one)
    assert len(stats1) == 1 and stats1[0]["cam_store_hits"] == 2 * len(names)
    olib = build_oracle.load()
    paint = lambda det: (np.asarray(det["mask"]).astype(np.int64) * (np.asarray(det["class"], np.int64) + 1)[:, None, None]).sum(0)
    walk = lambda x, e, radius, beta, exp_times: build_oracle.walk(olib, x, e, radius, beta, 2 ** exp_times)
    n_px = 0
    cam_dev = 0.0
    ties = {"two labels": 0, "two instance classes": 0, "one labels": 0, "one instance classes": 0}
    for n in names:
        a = np.load(os.path.join(two.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        b = np.load(os.path.join(one.cam_out_dir, n + ".npy"), allow_pickle=True).item()
        assert torch.equal(a["keys"], b["keys"])
        pa = np.asarray(Image.open(os.path.join(two.sem_seg_out_dir, n + ".png")))
        pb = np.asarray(Image.open(os.path.join(one.sem_seg_out_dir, n + ".png")))
        assert pa.shape == pb.shape
        fa, fb = os.path.join(two.ins_seg_out_dir, n + ".npy"), os.path.join(one.ins_seg_out_dir, n + ".npy")
        da = np.load(fa, allow_pickle=True).item() if os.path.exists(fa) else None
        db = np.load(fb, allow_pickle=True).item() if os.path.exists(fb) else None
        ea = np.load(os.path.join(two.edge_out_dir, n + ".npy"), allow_pickle=True).item()
        eb = np.load(os.path.join(one.edge_out_dir, n + ".npy"), allow_pickle=True).item()
        n_px += pa.size
        if mode == "deterministic":
            assert torch.equal(a["cam"], b["cam"]) and np.array_equal(a["high_res"], b["high_res"]), n
            assert np.array_equal(ea["edge"], eb["edge"]) and np.array_equal(ea["dp"], eb["dp"]), n
            assert np.array_equal(pa, pb), n
            assert (da is None) == (db is None), n
            if da is not None:
                assert da["mask"].shape == db["mask"].shape, (n, da["mask"].shape, db["mask"].shape, da["class"], db["class"])
                assert np.array_equal(da["class"], db["class"]), (n, da["class"], db["class"])
                assert np.array_equal(da["mask"], db["mask"]), (n, int((da["mask"] != db["mask"]).sum()))
                assert np.array_equal(da["score"], db["score"]), (n, da["score"], db["score"])
            continue
        # fast mode: the backbones' outputs agree to fp32 rounding (split-K accumulation order), far inside the 1e-4 bar ...
        cam_dev = max(cam_dev, (a["cam"] - b["cam"]).abs().max().item(), float(np.abs(a["high_res"] - b["high_res"]).max()),
                      float(np.abs(ea["edge"] - eb["edge"]).max()))
        assert cam_dev <= 5e-5, (n, cam_dev)
        # ... and each run's outputs equal the oracle's on that run's own inputs except at proven ties
        H, W = pa.shape
        for tag, cam_d, edges, png, det in (("two", a, ea, pa, da), ("one", b, eb, pb, db)):
            keys = cam_d["keys"].numpy()
            rw = build_oracle.walk(olib, cam_d["cam"].numpy(), edges["edge"][0], 10, 10, 256)
            up, want, _ = O.sem_seg_epilogue(rw, (H, W), keys, 0.25)
            ties[tag + " labels"] += label_mismatches(png, want, up, 0.25, lut=np.concatenate([[0], keys + 1]), what="%s %s-worker labels" % (n, tag))[0]
            if det is None:                          # an image without detections writes no file (the reference would crash)
                continue
            _, inst, rw_i, _, want_i = O.instance_labels(cam_d["cam"].numpy(), keys, edges["edge"][0], edges["dp"], (H, W), walk=walk, radius=10)
            up_i, _, _ = O.sem_seg_epilogue(rw_i, (H, W), np.zeros(rw_i.shape[0], np.int64), 0.25)
            chan_class = np.concatenate([[0], np.repeat(keys, inst.shape[0]) + 1])
            ties[tag + " instance classes"] += label_mismatches(paint(det), paint(want_i), up_i, 0.25, lut=chan_class,
                                                                 what="%s %s-worker instance classes" % (n, tag))[0]
    if mode == "deterministic":
        print("two workers vs one worker, deterministic mode: CAMs, edge / displacement maps, label maps and detections of %d images "
              "bit-identical (%d label pixels)" % (len(names), n_px))
    else:
        print("two workers vs one worker, fast mode: backbone outputs within %.1e; pixels differing from the oracle on the run's own inputs "
              "(each proven a < 1e-4 tie) of %d: %s" % (cam_dev, n_px, ties))


def test_run_sample_cli_end_to_end(tmp_path):
    """`python run_sample.py ...` itself (reference run_sample.py:8-137): an existing command line of the reference — its
    training / CRF hyper-parameter flags included — runs the three label-generation passes and writes every output."""
    import sys
    import run_sample
    from irn_amd.misc import pyutils
    from irn_amd.net import weights
    root, names, labels = _make_voc(tmp_path, n=2)
    torch.save(weights.random_cam_state(1), tmp_path / "res50_cam.pth")
    torch.save(weights.random_irn_state(2), tmp_path / "res50_irn.pth")
    lst = str(tmp_path / "lists" / "train.txt")
    stdout = sys.stdout
    try:
        run_sample.main(["--voc12_root", str(root), "--train_list", lst, "--infer_list", lst, "--num_workers", "2",
                         "--cam_weights_name", str(tmp_path / "res50_cam"), "--irn_weights_name", str(tmp_path / "res50_irn.pth"),
                         "--cam_out_dir", str(tmp_path / "cam"), "--sem_seg_out_dir", str(tmp_path / "sem"),
                         "--ins_seg_out_dir", str(tmp_path / "ins"), "--ir_label_out_dir", str(tmp_path / "ir"),
                         "--log_name", str(tmp_path / "log"), "--cam_scales", "1.0", "0.5",
                         "--cam_learning_rate", "0.05", "--irn_batch_size", "16", "--conf_fg_thres", "0.3", "--beta", "10",
                         "--exp_times", "8", "--train_cam_pass", "False", "--eval_cam_pass", "False",
                         # this build's own flags: the schedule switch, the step deadline, the device hand-offs
                         "--walk_accel", "1", "--walk_accel_tol_exp", "7", "--step_timeout", "900",
                         "--keep_cams_on_device", "1", "--keep_edges_on_device", "1"])
    finally:
        if isinstance(sys.stdout, pyutils.Logger):
            sys.stdout.close()
        sys.stdout = stdout
    for n in names:
        assert os.path.exists(tmp_path / "cam" / (n + ".npy")) and os.path.exists(tmp_path / "sem" / (n + ".png"))
        W, H = Image.open(root / "JPEGImages" / (n + ".jpg")).size
        assert np.asarray(Image.open(tmp_path / "sem" / (n + ".png"))).shape == (H, W)
    assert any(os.path.exists(tmp_path / "ins" / (n + ".npy")) for n in names)
    # the reference's own schedule (2^exp_times applications) through the same command line: labels equal except at ties
    stdout = sys.stdout
    try:
        run_sample.main(["--voc12_root", str(root), "--train_list", lst, "--infer_list", lst, "--num_workers", "2",
                         "--cam_weights_name", str(tmp_path / "res50_cam"), "--irn_weights_name", str(tmp_path / "res50_irn.pth"),
                         "--cam_out_dir", str(tmp_path / "cam"), "--sem_seg_out_dir", str(tmp_path / "sem_plain"),
                         "--ins_seg_out_dir", str(tmp_path / "ins_plain"), "--log_name", str(tmp_path / "log_plain"),
                         "--cam_scales", "1.0", "0.5", "--make_cam_pass", "False", "--make_ins_seg_pass", "False",
                         "--walk_accel", "0"])
    finally:
        if isinstance(sys.stdout, pyutils.Logger):
            sys.stdout.close()
        sys.stdout = stdout
    n_px = n_diff = 0
    for n in names:
        a = np.asarray(Image.open(tmp_path / "sem" / (n + ".png")))
        b = np.asarray(Image.open(tmp_path / "sem_plain" / (n + ".png")))
        n_px += a.size
        n_diff += int((a != b).sum())
    print("run_sample.py --walk_accel 1 vs 0: %d of %d semantic label pixels differ" % (n_diff, n_px))
    # two schedules of the same operator on the SAME inputs (the second run takes the first one's edge maps from device
    # memory, the CAM files are shared): deterministic kernels, measured 0 in every session since round 3
    assert n_diff == 0
    with pytest.raises(SystemExit):                      # a pass this build does not implement refuses loudly
        run_sample.main(["--voc12_root", str(root), "--train_irn_pass", "True", "--log_name", str(tmp_path / "log2")])


def test_cam_merge_kernel_vs_oracle_and_reference_golden(golden):
    """irn_cam_merge (step/make_cam.py:38-52) bit for bit against the oracle's restatement, against the
    reference's own output (1e-6; bar 1e-4) and against the torch-op mirror on the GPU."""
    from irn_amd import ops
    from oracle import irn_oracle as O, torch_mirrors
    dev = torch.device("cuda", 0)
    cm = golden("cam_merge")
    for name in "ab":
        outs = [cm["%s_out%d" % (name, i)] for i in range(4)]
        size = tuple(int(v) for v in cm[name + "_size"])
        keys, cam, hi = ops.cam_merge([torch.from_numpy(o).to(dev) for o in outs], size, torch.from_numpy(cm[name + "_label"]))
        ok, olo, ohi = O.cam_merge(outs, size, cm[name + "_label"])
        assert np.array_equal(keys.cpu().numpy(), ok) and np.array_equal(ok, cm[name + "_keys"])
        assert np.array_equal(cam.cpu().numpy(), olo)
        assert np.array_equal(hi.cpu().numpy(), ohi)
        assert np.abs(cam.cpu().numpy() - cm[name + "_cam"]).max() <= 1e-6
        assert np.abs(hi.cpu().numpy() - cm[name + "_high_res"]).max() <= 1e-6
    # ragged sizes, other scale sets, single class
    rng = np.random.RandomState(3)
    for (H, W, shapes, label_idx) in ((375, 500, [(24, 32), (12, 16), (36, 47), (47, 63)], [1, 14]),
                                      (333, 97, [(21, 7), (42, 13)], [19]),
                                      (512, 512, [(32, 32), (16, 16), (48, 48), (64, 64)], [0, 5, 9, 17])):
        outs = [np.abs(rng.randn(20, h, w)).astype(np.float32) for h, w in shapes]
        label = np.zeros(20, np.float32)
        label[label_idx] = 1
        keys, cam, hi = ops.cam_merge([torch.from_numpy(o).to(dev) for o in outs], (H, W), torch.from_numpy(label))
        ok, olo, ohi = O.cam_merge(outs, (H, W), label)
        assert np.array_equal(keys.cpu().numpy(), ok)
        assert np.array_equal(cam.cpu().numpy(), olo) and np.array_equal(hi.cpu().numpy(), ohi)
        tk, tc, th = torch_mirrors.merge_scales_torch([torch.from_numpy(o).to(dev) for o in outs], (H, W),
                                                 torch.from_numpy(label).to(dev))
        assert torch.equal(tk, keys) and (tc - cam).abs().max().item() <= 1e-5 and (th - hi).abs().max().item() <= 1e-5


def test_bench_contract_one_json_line():
    """python bench.py (small batch) prints ONE JSON line with the driver's fields, the `roofline` object of the dominant
    kernel (binding ceiling at the top level), the `cpu_baseline` object (port + reference algorithm) and the `legs`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "IRN_DETERMINISTIC"}        # the line of the DEFAULT mode, whatever this suite runs under
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--cpu-images", "2", "--ref-grids", "64", "--legs", "coco,ins,walk_plain,walk_voc,cam"], capture_output=True, text=True,
                         timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "legs"):
        assert key in r, key
    assert r["unit"] == "images/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["scaling"] == "weak"
    assert r["higher_is_better"] is True and r["vs_baseline"] is None and r["data"] == "synthetic" and r["dtype"] == "f32"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "fp32_vector" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1 and "traffic" in rf
    assert rf["hbm_equivalent"]["unit"] == "GB/s" and rf["hbm_equivalent"]["peak"] == 8000.0
    # `achieved` counts the flops the kernel executes (84 applications of the operator for T^256 at the default truncation bound); 8(d)'s F beside it
    sch = rf["schedule"]
    assert sch["n_sweeps"] == 256 and sch["operator_applications"] == 84 and rf["sweeps_per_launch"] == 84
    pe = rf["power_equivalent"]
    assert abs(pe["achieved"] / rf["achieved"] - 256.0 / 84.0) < 1e-6 and abs(pe["flops"] / rf["flops_per_launch"] - 256.0 / 84.0) < 1e-6
    lp = r["label_parity"]
    assert lp["images"] == 2 and lp["not_a_tie"] == 0 and lp["pixels_differing"] == 0 and lp["max_top2_gap"] < 1e-4
    assert abs(r["value"] - 2 * 8 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-6
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "images/s" and cb["sample"]
    ra = cb["reference_algorithm"]
    assert ra["extrapolated"] is True and ra["value"] > 0 and ra["seconds_measured"]["64x64"]["squarings"] > 0
    assert ra["cores"] == cb["cores"]
    for leg in ("coco", "ins", "walk_plain"):
        assert r["legs"][leg].get("value", 0) > 0, r["legs"][leg]
    assert r["legs"]["walk_plain"]["n_applied"] == 256 and r["legs"]["walk_plain"]["value"] < r["value"]
    # round 5 hygiene: no image twice in the timed batch, the traffic figure says where it comes from, the CPU is named, the
    # ragged leg reports how the persistent launch packed it and that nothing fell back, the backbone leg says which trunk ran
    assert r["config"]["images_per_gpu_per_step"] == 8
    # round 6: re-measured inside the run (two rocprofv3 --pmc passes of one launch each) when rocprofv3 is there, else the static figure
    assert rf["traffic_source"] in (None, "static", "measured") and (rf["traffic"] is None) == (rf["traffic_source"] is None)
    if rf["traffic_source"] == "measured":          # (a box whose rocprofv3 cannot collect counters falls back to the static figure and says why)
        assert 0 < rf["traffic"] < rf["hbm_equivalent"]["algorithmic_bytes_per_launch"], rf.get("traffic_detail")
    print("roofline.traffic: %s (%s)" % (rf["traffic"], rf["traffic_source"]), (rf.get("traffic_detail") or "")[-160:])
    assert isinstance(cb["cpu_model"], str) and cb["cpu_model"]
    wv = r["legs"]["walk_voc"]
    assert wv["value"] > 0 and wv["rounds"] > 0 and wv["fallback_runs"] == 0 and wv["grid_pixels"] > 0
    trunk = r["legs"]["cam"]["trunk"]
    assert trunk["layout"] in ("channels_last", "nchw") and trunk["fused_1x1_gemm"] == (trunk["layout"] == "channels_last")
    assert trunk["deterministic"] is True and trunk["miopen_key"].endswith("-det") and r["legs"]["cam"]["value"] > 0      # the default mode
    assert trunk["split_precision_1x1"] == (trunk["layout"] == "channels_last") and 0 < r["legs"]["cam"]["matrix_fp32_frac"] < 2
    cc = cb["cam"]                      # round 6: the CAM half of the metric on the host cores, beside legs.cam
    assert cc["kind"] == "port" and cc["value"] > 0 and cc["cores"] == cb["cores"] and abs(cc["gflop_per_image"] - 974.04) < 0.01


def test_upload_never_blocks_and_equals_cuda():
    """step/_common.upload: host -> device through a recycled page-locked buffer on the copy stream, ordered into the
    current stream by an event — the same bytes as `.cuda()`, without the host waiting for the stream to drain."""
    import time
    from irn_amd.step import _common
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    items = [torch.randint(0, 256, (375, 500, 3), dtype=torch.uint8, generator=g), torch.randn(2, 3, 33, 47, generator=g),
             torch.randint(0, 256, (1, 1, 3), dtype=torch.uint8, generator=g), torch.arange(7, dtype=torch.int64)]
    with torch.cuda.device(dev):
        for rep in range(20):
            for t in items:
                d = _common.upload(t)
                assert d.device == dev and d.dtype == t.dtype and d.shape == t.shape
                assert torch.equal(d.cpu(), t)
        torch.cuda.synchronize()
        _common.upload(items[0])
        assert len(_common._UPLOADS) <= 2                      # landed copies gave their staging buffers back
        # behind a long-running kernel chain the call returns at once (a pageable .cuda() would wait for the chain)
        a = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        for _ in range(60):
            a = a @ a * 1e-3
        t0 = time.perf_counter()
        d = _common.upload(items[0])
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        assert torch.equal(d.cpu(), items[0])
        print("upload behind 60 queued 4096^3 GEMMs returned after %.2f ms" % (1e3 * dt))
        assert dt < 0.02
