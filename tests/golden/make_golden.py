#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE ITSELF (CPU) in the build container.

The reference (jiwoon-ahn/irn, mounted read-only at /root/reference) has no tests and no
golden vectors of its own, so parity is pinned on outputs of the reference's
own code, imported unmodified from /root/reference and executed on CPU here.  The only
patches are environmental and do not touch arithmetic:

  * ``torch.Tensor.cuda`` -> identity            (misc/indexing.py:99,127 call .cuda())
  * ``skimage.measure.label`` -> a scipy.ndimage.label shim (skimage is not installed;
    both number 4-connected components in raster order of their first pixel)
  * empty ``imageio`` / ``pydensecrf`` modules so that ``step.make_ins_seg_labels`` imports
  * ``net.resnet50.model_zoo.load_url`` -> seeded random state dict (no network)

/root/reference does not exist on the GPU box, so the outputs are committed as small
``.npz`` files next to this script.  Re-run:  python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("IRN_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(OUT), ".."))


def _install_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not found at %s" % REF)
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self          # CPU stand-in for .cuda()
    import scipy.ndimage

    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.measure")

    def label(arr, connectivity=1, background=0):
        assert connectivity == 1 and background == 0
        return scipy.ndimage.label(np.asarray(arr) != 0)[0]

    skm.label = label
    sk.measure = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = skm
    sys.modules["imageio"] = types.ModuleType("imageio")
    pdc = types.ModuleType("pydensecrf")
    pdc_d = types.ModuleType("pydensecrf.densecrf")
    pdc_u = types.ModuleType("pydensecrf.utils")
    pdc_u.unary_from_labels = None
    pdc.densecrf, pdc.utils = pdc_d, pdc_u
    sys.modules["pydensecrf"] = pdc
    sys.modules["pydensecrf.densecrf"] = pdc_d
    sys.modules["pydensecrf.utils"] = pdc_u
    if not hasattr(np, "bool"):
        np.bool = np.bool_                                     # misc/pyutils.py:86 default arg
    os.chdir(REF)                                              # voc12/dataloader.py:22 relative np.load


# ----------------------------------------------------------------------------------------
# synthetic generators (shared with tests/ and bench.py through irn_amd.synth)
# ----------------------------------------------------------------------------------------
from irn_amd import synth  # noqa: E402


def gen_path_tables():
    from misc import indexing
    out = {}
    for r in (2, 3, 5, 7, 10):
        pi = indexing.PathIndex(radius=r, default_size=(r + 6, 2 * r + 7))
        out["r%d_dst" % r] = np.asarray(pi.search_dst, np.int32)
        out["r%d_group_lens" % r] = np.asarray([p.shape[1] for p in pi.search_paths], np.int32)
        out["r%d_group_counts" % r] = np.asarray([p.shape[0] for p in pi.search_paths], np.int32)
        out["r%d_paths_flat" % r] = np.concatenate([p.reshape(-1, 2) for p in pi.search_paths]).astype(np.int32)
        out["r%d_src_indices" % r] = pi.src_indices.astype(np.int64)
        out["r%d_dst_indices" % r] = pi.dst_indices.astype(np.int64)
        out["r%d_path_indices_flat" % r] = np.concatenate([p.reshape(-1) for p in pi.path_indices]).astype(np.int64)
        out["r%d_size" % r] = np.asarray((r + 6, 2 * r + 7), np.int32)
    np.savez_compressed(os.path.join(OUT, "path_tables.npz"), **out)


def gen_affinity():
    from misc import indexing
    out = {}
    for r, (h, w) in ((5, (17, 23)), (10, (21, 26)), (3, (9, 12))):
        edge = torch.from_numpy(synth.edge_field(h, w, seed=100 + r))[None]
        edge_padded = F.pad(edge, (r, r, 0, r), mode="constant", value=1.0)
        pi = indexing.PathIndex(radius=r, default_size=(h + r, w + 2 * r))
        aff = indexing.edge_to_affinity(torch.unsqueeze(edge_padded, 0), pi.path_indices)
        out["r%d_edge" % r] = edge[0].numpy()
        out["r%d_aff" % r] = aff[0].numpy()
    np.savez_compressed(os.path.join(OUT, "affinity.npz"), **out)


def gen_affinity_grad():
    """Training seam: AffinityDisplacementLoss.to_affinity (net/resnet50_irn.py:162-175) under autograd —
    forward values and the gradient of sum(aff * g) w.r.t. the edge map, batch of 2, radius 5 (the
    training configuration's index tensors come from PathIndex like step/train_irn.py:12-21)."""
    from misc import indexing
    from net import resnet50_irn
    out = {}
    for r, (hp, wp) in ((5, (20, 27)), (3, (12, 15))):
        pi = indexing.PathIndex(radius=r, default_size=(hp, wp))
        loss = resnet50_irn.AffinityDisplacementLoss.__new__(resnet50_irn.AffinityDisplacementLoss)
        torch.nn.Module.__init__(loss)
        loss.path_index = pi
        loss.n_path_lengths = len(pi.path_indices)
        for i, pind in enumerate(pi.path_indices):
            loss.register_buffer(resnet50_irn.AffinityDisplacementLoss.path_indices_prefix + str(i), torch.from_numpy(pind))
        edge = torch.stack([torch.from_numpy(synth.edge_field(hp, wp, seed=300 + r + b)) for b in range(2)])[:, None]
        edge.requires_grad_(True)
        aff = loss.to_affinity(edge)
        g = torch.from_numpy(np.random.RandomState(7 + r).randn(*aff.shape).astype(np.float32))
        (aff * g).sum().backward()
        out["r%d_edge" % r] = edge.detach()[:, 0].numpy()
        out["r%d_aff" % r] = aff.detach().numpy()
        out["r%d_gout" % r] = g.numpy()
        out["r%d_gedge" % r] = edge.grad[:, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "affinity_grad.npz"), **out)


def gen_pair_disp():
    """Training seam: AffinityDisplacementLoss.to_pair_displacement (net/resnet50_irn.py:177-193) under autograd:
    forward values and the gradient of sum(pair_disp * g) w.r.t. the displacement field."""
    from misc import indexing
    from net import resnet50_irn
    out = {}
    for r, (hp, wp) in ((5, (20, 27)), (3, (12, 15)), (10, (16, 29))):
        pi = indexing.PathIndex(radius=r, default_size=(hp, wp))
        loss = resnet50_irn.AffinityDisplacementLoss.__new__(resnet50_irn.AffinityDisplacementLoss)
        torch.nn.Module.__init__(loss)
        loss.path_index = pi
        disp = torch.stack([torch.from_numpy(synth.displacement_field(hp, wp, seed=400 + r + b)) for b in range(2)])
        disp.requires_grad_(True)
        pd = loss.to_pair_displacement(disp)
        g = torch.from_numpy(np.random.RandomState(11 + r).randn(*pd.shape).astype(np.float32))
        (pd * g).sum().backward()
        out["r%d_disp" % r] = disp.detach().numpy()
        out["r%d_pair" % r] = pd.detach().numpy()
        out["r%d_gdisp" % r] = disp.grad.numpy()     # g is regenerated from its seed by the tests (RandomState(11 + r))
    np.savez_compressed(os.path.join(OUT, "pair_disp.npz"), **out)


WALK_CASES = [
    # name, h, w, C, radius, beta, exp_times, seed
    ("r5_b10_e8", 32, 32, 3, 5, 10, 8, 1),
    ("r5_b8_e8", 32, 32, 3, 5, 8, 8, 2),
    ("r5_b10_e0", 24, 28, 2, 5, 10, 0, 3),
    ("r5_b10_e1", 24, 28, 2, 5, 10, 1, 4),
    ("r5_b10_e4", 24, 28, 2, 5, 10, 4, 5),
    ("r10_b10_e8", 32, 32, 3, 10, 10, 8, 6),
    ("r10_b10_e8_ragged", 24, 40, 4, 10, 10, 8, 7),
    ("r5_b10_e8_ragged", 47, 31, 1, 5, 10, 8, 8),
    ("r5_b10_e8_rand", 32, 32, 3, 5, 10, 8, 9),      # iid-random edge (the SURVEY probe input)
    ("r5_b10_e8_ck", 28, 36, 6, 5, 10, 8, 10),       # 4-D x [C,K,h,w] like the instance path
    ("r5_b10_e8_64", 64, 64, 3, 5, 10, 8, 11),
    # round 6, the degenerate corner on the reference itself (misc/indexing.py:123-126,135: an isolated pixel keeps a non-zero
    # column sum only through the unit diagonal): 0/1 Bernoulli edges (affinities exactly 0 or 1), and edge = 1 on a block
    # larger than the radius (every pixel inside has no neighbour at all)
    ("r5_b10_e8_bern", 32, 32, 3, 5, 10, 8, 12),
    ("r10_b10_e8_bern", 32, 32, 2, 10, 10, 8, 13),
    ("r5_b10_e8_block", 40, 36, 3, 5, 10, 8, 14),
    ("r10_b10_e8_block", 40, 44, 2, 10, 10, 8, 15),
]


# The headline grid (BASELINE configs[2]: 512^2 images = 128x128 stride-4 grids) on the reference itself: one dense
# propagate_to_edge run is ~5 min and ~5 GB here (18 354^2 fp32 matrices squared 8 times), so these live in their own
# file (walk128.npz) and are generated with `--only walk128`.
WALK_BIG_CASES = [
    ("r10_b10_e8_128", 128, 128, 3, 10, 10, 8, 21),
    ("r5_b10_e8_128", 128, 128, 2, 5, 10, 8, 22),
]

# ragged grids of real VOC images (500x375 and 334x500 photos -> 94x125 and 84x125 stride-4 grids), the reference's own
# call-site setting (radius 5, step/make_sem_seg_labels.py:41) and the headline one (radius 10); round 3
WALK_VOC_CASES = [
    ("r5_b10_e8_voc", 94, 125, 3, 5, 10, 8, 31),
    ("r10_b10_e8_voc", 84, 125, 2, 10, 10, 8, 32),
]


def gen_walk(only=None, cases=None, fname="walk.npz"):
    from misc import indexing
    out = {}
    path = os.path.join(OUT, fname)
    if only and os.path.exists(path):
        out = dict(np.load(path))
    for name, h, w, C, r, beta, e, seed in (cases or WALK_CASES):
        if only and name not in only:
            continue
        t0 = time.time()
        if name.endswith("_rand"):
            g = torch.Generator().manual_seed(seed)
            edge = torch.sigmoid(2 * torch.randn(1, h, w, generator=g))
            cam = torch.rand(C, h, w, generator=g)
        elif name.endswith("_bern"):
            g = torch.Generator().manual_seed(seed)
            edge = (torch.rand(1, h, w, generator=g) < 0.3).float()
            cam = torch.from_numpy(synth.cam_blobs(C, h, w, seed=seed))
        else:
            edge = torch.from_numpy(synth.edge_field(h, w, seed=seed))[None]
            cam = torch.from_numpy(synth.cam_blobs(C, h, w, seed=seed))
            if name.endswith("_block"):
                edge[0, 8:8 + 2 * r + 4, 6:6 + 2 * r + 2] = 1.0
        x = cam
        if name.endswith("_ck"):
            x = cam.view(2, C // 2, h, w)
        rw = indexing.propagate_to_edge(x.clone(), edge.clone(), radius=r, beta=beta, exp_times=e)
        out[name + "_edge"] = edge[0].numpy()
        out[name + "_cam"] = cam.numpy()
        out[name + "_rw"] = rw.numpy()
        out[name + "_params"] = np.asarray([h, w, C, r, beta, e], np.int32)
        print("walk %-22s %.1fs  max=%.4g" % (name, time.time() - t0, float(rw.max())), flush=True)
        np.savez_compressed(path, **out)


def gen_semseg():
    """Epilogue of step/make_sem_seg_labels.py:36-49 executed with the reference's torch ops."""
    d = dict(np.load(os.path.join(OUT, "walk.npz")))
    out = {}
    for name, (H, W), keys_in, bg in (("r5_b10_e8", (125, 127), [0, 7, 14], 0.25),
                                      ("r10_b10_e8_ragged", (93, 160), [1, 3, 8, 19], 0.25),
                                      ("r5_b10_e8_ragged", (188, 121), [11], 0.3),
                                      ("r5_b10_e8_64", (256, 253), [2, 5, 6], 0.25)):
        rw = torch.from_numpy(d[name + "_rw"])
        keys = np.pad(np.asarray(keys_in, np.int64) + 1, (1, 0), mode="constant")
        rw_up = F.interpolate(rw, scale_factor=4, mode="bilinear", align_corners=False)[..., 0, :H, :W]
        rw_up = rw_up / torch.max(rw_up)
        rw_up_bg = F.pad(rw_up, (0, 0, 0, 0, 1, 0), value=bg)
        rw_pred = torch.argmax(rw_up_bg, dim=0).cpu().numpy()
        rw_pred = keys[rw_pred]
        out[name + "_size"] = np.asarray([H, W], np.int32)
        out[name + "_keys"] = np.asarray(keys_in, np.int64)
        out[name + "_bg"] = np.asarray(bg, np.float32)
        out[name + "_rw_up"] = rw_up.numpy()
        out[name + "_label"] = rw_pred.astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "semseg.npz"), **out)


def gen_instance():
    import step.make_ins_seg_labels as mis
    from misc import indexing
    out = {}
    for name, h, w, C, seed, (H, W) in (("a", 30, 38, 2, 20, (120, 150)),
                                        ("b", 41, 29, 1, 37, (161, 116)),
                                        ("c", 32, 32, 3, 33, (128, 128))):
        dp = synth.displacement_field(h, w, seed=seed, strength=0.45)
        edge = torch.from_numpy(synth.edge_field(h, w, seed=seed))[None]
        cams = torch.from_numpy(synth.cam_blobs(C, h, w, seed=seed))
        keys = torch.from_numpy(np.sort(np.random.RandomState(seed).choice(20, C, replace=False)).astype(np.int64))
        cen = mis.find_centroids_with_refinement(dp)
        inst = mis.cluster_centroids(cen, dp)
        icam = mis.separte_score_by_mask(cams, inst)
        rw = indexing.propagate_to_edge(icam, edge, beta=10, exp_times=8, radius=5)
        rw_up = F.interpolate(rw, scale_factor=4, mode="bilinear", align_corners=False)[:, 0, :H, :W]
        rw_up = rw_up / torch.max(rw_up)
        rw_up_bg = F.pad(rw_up, (0, 0, 0, 0, 1, 0), value=0.25)
        nc, ni = len(keys), inst.shape[0]
        shape = torch.argmax(rw_up_bg, 0).cpu().numpy()
        from misc import pyutils
        shape_oh = pyutils.to_one_hot(shape, maximum_val=ni * nc + 1)[1:]
        cls = np.repeat(keys, ni)
        det = mis.detect_instance(rw_up.cpu().numpy(), shape_oh, cls, max_fragment_size=H * W * 0.01)
        out[name + "_dp"] = dp
        out[name + "_edge"] = edge[0].numpy()
        out[name + "_cam"] = cams.numpy()
        out[name + "_keys"] = keys.numpy()
        out[name + "_size"] = np.asarray([H, W], np.int32)
        out[name + "_centroids"] = cen
        out[name + "_instance_map"] = np.packbits(inst, axis=None)
        out[name + "_instance_map_shape"] = np.asarray(inst.shape, np.int32)
        out[name + "_rw"] = rw.numpy()
        out[name + "_argmax"] = shape.astype(np.int32)
        out[name + "_det_score"] = np.asarray(det["score"], np.float32)
        out[name + "_det_class"] = np.asarray(det["class"], np.int64)
        out[name + "_det_mask"] = np.packbits(det["mask"], axis=None)
        out[name + "_det_mask_shape"] = np.asarray(det["mask"].shape, np.int32)
        print("instance", name, "K=", ni, "ndet=", len(det["score"]), flush=True)
    # centroid-only cases (larger, exercises clipping + ties)
    for name, h, w, seed in (("cen64", 64, 64, 31), ("cen_ragged", 94, 125, 32)):
        dp = synth.displacement_field(h, w, seed=seed, strength=0.35)
        cen = mis.find_centroids_with_refinement(dp)
        inst = mis.cluster_centroids(cen, dp)
        out[name + "_dp"] = dp
        out[name + "_centroids"] = cen
        out[name + "_instance_map"] = np.packbits(inst, axis=None)
        out[name + "_instance_map_shape"] = np.asarray(inst.shape, np.int32)
    np.savez_compressed(os.path.join(OUT, "instance.npz"), **out)


def gen_cam_merge():
    """step/make_cam.py:32-52 with the reference's torch ops on synthetic per-scale outputs."""
    from misc import imutils
    out = {}
    for name, (H, W), seed in (("a", (125, 163), 41), ("b", (64, 64), 42)):
        g = torch.Generator().manual_seed(seed)
        size = (H, W)
        strided_size = imutils.get_strided_size(size, 4)
        strided_up_size = imutils.get_strided_up_size(size, 16)
        outputs = []
        for s in (1.0, 0.5, 1.5, 2.0):
            hs, ws = int(np.round(H * s)), int(np.round(W * s))
            fh, fw = (hs + 15) // 16, (ws + 15) // 16
            outputs.append(torch.relu(torch.randn(20, fh, fw, generator=g)))
        label = torch.zeros(20)
        label[[3, 11, 17]] = 1
        strided_cam = torch.sum(torch.stack(
            [F.interpolate(torch.unsqueeze(o, 0), strided_size, mode="bilinear", align_corners=False)[0] for o in outputs]), 0)
        highres_cam = [F.interpolate(torch.unsqueeze(o, 1), strided_up_size, mode="bilinear", align_corners=False) for o in outputs]
        highres_cam = torch.sum(torch.stack(highres_cam, 0), 0)[:, 0, :size[0], :size[1]]
        valid_cat = torch.nonzero(label)[:, 0]
        strided_cam = strided_cam[valid_cat]
        strided_cam /= F.adaptive_max_pool2d(strided_cam, (1, 1)) + 1e-5
        highres_cam = highres_cam[valid_cat]
        highres_cam /= F.adaptive_max_pool2d(highres_cam, (1, 1)) + 1e-5
        for i, o in enumerate(outputs):
            out["%s_out%d" % (name, i)] = o.numpy()
        out[name + "_size"] = np.asarray(size, np.int32)
        out[name + "_label"] = label.numpy()
        out[name + "_keys"] = valid_cat.numpy()
        out[name + "_cam"] = strided_cam.numpy()
        out[name + "_high_res"] = highres_cam.numpy()
    np.savez_compressed(os.path.join(OUT, "cam_merge.npz"), **out)


def gen_nets():
    """CAM / EdgeDisplacement forward of the reference with a seeded random state dict."""
    import net.resnet50 as r50
    from irn_amd.net import weights as wgen

    real_load = r50.model_zoo.load_url
    r50.model_zoo.load_url = lambda *a, **k: dict(wgen.random_resnet50_state(seed=0), **{
        "fc.weight": torch.zeros(1), "fc.bias": torch.zeros(1)})
    try:
        import net.resnet50_cam as rc
        import net.resnet50_irn as ri
        out = {}
        cam = rc.CAM()
        cam.load_state_dict(wgen.random_cam_state(seed=1), strict=True)
        g = torch.Generator().manual_seed(5)
        img = torch.randn(1, 3, 96, 112, generator=g)
        x = torch.cat([img, img.flip(-1)], 0)
        with torch.no_grad():
            y = cam(x)
        out["cam_in"] = x.numpy()
        out["cam_out"] = y.numpy()
        irn = ri.EdgeDisplacement(crop_size=128)
        irn.load_state_dict(wgen.random_irn_state(seed=2), strict=False)
        irn.eval()
        img = torch.randn(1, 3, 101, 122, generator=g)
        x = torch.cat([img, img.flip(-1)], 0)
        with torch.no_grad():
            edge, dp = irn(x)
        out["irn_in"] = x.numpy()
        out["irn_edge"] = edge.numpy()
        out["irn_dp"] = dp.numpy()
        np.savez_compressed(os.path.join(OUT, "nets.npz"), **out)
    finally:
        r50.model_zoo.load_url = real_load


def gen_nets512():
    """The backbones at the headline input size (BASELINE configs[1]/[2]: 512^2 VOC images): the reference's
    CAM.forward on [2,3,512,512] and EdgeDisplacement.forward on a ragged [2,3,375,500] item (crop_size 512), seeded
    random weights.  Inputs are `synth.image_pair(h, w, seed)` and are NOT stored (regenerated by the tests)."""
    import net.resnet50 as r50
    from irn_amd.net import weights as wgen

    real_load = r50.model_zoo.load_url
    r50.model_zoo.load_url = lambda *a, **k: dict(wgen.random_resnet50_state(seed=0), **{
        "fc.weight": torch.zeros(1), "fc.bias": torch.zeros(1)})
    try:
        import net.resnet50_cam as rc
        import net.resnet50_irn as ri
        out = {}
        cam = rc.CAM()
        cam.load_state_dict(wgen.random_cam_state(seed=1), strict=True)
        cam.eval()
        irn = ri.EdgeDisplacement()
        irn.load_state_dict(wgen.random_irn_state(seed=2), strict=False)
        irn.eval()
        with torch.no_grad():
            out["cam512_seed"] = np.asarray([512, 512, 71], np.int32)
            out["cam512_out"] = cam(torch.from_numpy(synth.image_pair(512, 512, 71))).numpy()
            out["cam768_seed"] = np.asarray([768, 768, 72], np.int32)          # the 1.5x scale of make_cam
            out["cam768_out"] = cam(torch.from_numpy(synth.image_pair(768, 768, 72))).numpy()
            out["irn_seed"] = np.asarray([375, 500, 73], np.int32)
            edge, dp = irn(torch.from_numpy(synth.image_pair(375, 500, 73)))
            out["irn_edge"] = edge.numpy()
            out["irn_dp"] = dp.numpy()
            out["irn512_seed"] = np.asarray([512, 512, 74], np.int32)
            edge, dp = irn(torch.from_numpy(synth.image_pair(512, 512, 74)))
            out["irn512_edge"] = edge.numpy()
            out["irn512_dp"] = dp.numpy()
        np.savez_compressed(os.path.join(OUT, "nets512.npz"), **out)
    finally:
        r50.model_zoo.load_url = real_load


def gen_nets_scales():
    """CAM.forward (net/resnet50_cam.py:55-70) at the OTHER two scales make_cam feeds it for a 512^2 image
    (voc12/dataloader.py:191-199 with run_sample.py:31's cam_scales): 0.5x -> [2,3,256,256] and 2.0x -> [2,3,1024,1024]
    (nets512.npz holds 1.0x and 1.5x).  Same seeded weights; inputs regenerated by the tests from the seeds."""
    import net.resnet50 as r50
    from irn_amd.net import weights as wgen

    real_load = r50.model_zoo.load_url
    r50.model_zoo.load_url = lambda *a, **k: dict(wgen.random_resnet50_state(seed=0), **{
        "fc.weight": torch.zeros(1), "fc.bias": torch.zeros(1)})
    try:
        import net.resnet50_cam as rc
        out = {}
        cam = rc.CAM()
        cam.load_state_dict(wgen.random_cam_state(seed=1), strict=True)
        cam.eval()
        with torch.no_grad():
            for size, seed in ((256, 75), (1024, 76)):
                out["cam%d_seed" % size] = np.asarray([size, size, seed], np.int32)
                out["cam%d_out" % size] = cam(torch.from_numpy(synth.image_pair(size, size, seed))).numpy()
        np.savez_compressed(os.path.join(OUT, "nets_scales.npz"), **out)
    finally:
        r50.model_zoo.load_url = real_load


def gen_msf():
    """Multi-scale dataset item: the reference's own VOC12ClassificationDatasetMSF.__getitem__
    (voc12/dataloader.py:185-205) on synthetic photos (imageio.imread stubbed to hand them over), i.e.
    misc/imutils.pil_rescale -> installed Pillow's Image.resize(BICUBIC) -> TorchvisionNormalize -> CHW ->
    flip pair; plus raw Pillow resizes (via the reference's pil_resize) of ragged sizes."""
    import PIL
    from misc import imutils
    import voc12.dataloader as vd
    out = {"pillow_version": np.asarray(PIL.__version__)}
    scales = (1.0, 0.5, 1.5, 2.0)
    out["scales"] = np.asarray(scales)
    cases = {"a": (45, 60), "b": (53, 37)}
    for name, (h, w) in cases.items():
        img = synth.photo(h, w, seed=500 + h)
        sys.modules["imageio"].imread = lambda path, _img=img: _img
        ds = vd.VOC12ClassificationDatasetMSF.__new__(vd.VOC12ClassificationDatasetMSF)
        ds.img_name_list = np.asarray([2007000032])
        ds.label_list = np.zeros((1, 20), np.float32)
        ds.voc12_root = "/nonexistent"
        ds.img_normal = vd.TorchvisionNormalize()
        ds.scales = scales
        item = ds[0]
        out["%s_img" % name] = img
        for i, arr in enumerate(item["img"]):
            out["%s_item%d" % (name, i)] = np.ascontiguousarray(arr)
        assert item["size"] == (h, w)
    rng = np.random.default_rng(7)
    sizes = [((31, 47), (97, 13)), ((64, 64), (17, 64)), ((9, 200), (9, 77)), ((120, 90), (1, 1)), ((5, 3), (40, 41))]
    for i, ((h, w), tgt) in enumerate(sizes):
        img = synth.photo(h, w, seed=600 + i) if i % 2 == 0 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out["resize%d_img" % i] = img
        out["resize%d_out" % i] = np.asarray(imutils.pil_resize(img, tgt, 3))
    gray = synth.photo(40, 50, seed=650)[..., 0].copy()
    out["gray_img"] = gray
    out["gray_out"] = np.asarray(imutils.pil_resize(gray, (61, 33), 3))
    np.savez_compressed(os.path.join(OUT, "msf.npz"), **out)


def gen_trunk_ops():
    """The elementwise tails of the reference's trunk and heads, run by the reference's own modules on seeded tensors
    standing in for the convolutions' outputs: FixedBatchNorm -> += residual -> ReLU (net/resnet50.py:11-14, :34-54, with
    and without the projection shortcut's batch norm :48-49), conv1's tail bn1 -> relu -> maxpool (:94-97), and the heads'
    Upsample -> ReLU (net/resnet50_irn.py:36-48, :72-84)."""
    import net.resnet50 as r50
    g = torch.Generator().manual_seed(21)

    def bn(c):
        m = r50.FixedBatchNorm(c)
        with torch.no_grad():
            m.weight.copy_(torch.rand(c, generator=g) * 2 - 0.5)
            m.bias.copy_(torch.randn(c, generator=g))
            m.running_mean.copy_(torch.randn(c, generator=g))
            m.running_var.copy_(torch.rand(c, generator=g) + 0.05)
        return m

    def params(prefix, m):
        return {prefix + "_w": m.weight.detach().numpy(), prefix + "_b": m.bias.detach().numpy(),
                prefix + "_mean": m.running_mean.numpy(), prefix + "_var": m.running_var.numpy(),
                prefix + "_eps": np.float64(m.eps)}

    out = {}
    relu = torch.nn.ReLU(inplace=True)
    with torch.no_grad():
        for tag, shape in (("a", (2, 6, 9, 13)), ("b", (1, 5, 1, 3)), ("c", (3, 4, 8, 8))):
            x = torch.randn(shape, generator=g)
            res = torch.randn(shape, generator=g)
            m, md = bn(shape[1]), bn(shape[1])
            out.update(params("bn_" + tag, m))
            out.update(params("bnd_" + tag, md))
            out["x_" + tag], out["res_" + tag] = x.numpy().copy(), res.numpy().copy()
            out["bn_relu_" + tag] = relu(m(x)).numpy().copy()                    # Bottleneck.forward :37-43
            y = m(x)
            y += res                                                             # :51
            out["bn_add_relu_" + tag] = relu(y).numpy().copy()                   # :52
            y = m(x)
            y += md(res)                                                         # :48-51 (downsample = conv, FixedBatchNorm)
            out["bn_addbn_relu_" + tag] = relu(y).numpy().copy()
            out["bn_plain_" + tag] = m(x).numpy().copy()
        pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)            # net/resnet50.py:66
        for tag, shape in (("s1", (2, 4, 12, 17)), ("s2", (1, 3, 7, 8)), ("s3", (1, 2, 1, 1))):
            x = torch.randn(shape, generator=g)
            m = bn(shape[1])
            out.update(params("stem_" + tag, m))
            out["stem_x_" + tag] = x.numpy().copy()
            out["stem_out_" + tag] = pool(relu(m(x))).numpy().copy()             # :94-97
        for tag, shape, f in (("u2", (2, 3, 7, 5), 2), ("u4", (1, 4, 6, 9), 4), ("u2b", (1, 2, 1, 1), 2)):
            x = torch.randn(shape, generator=g)
            up = torch.nn.Upsample(scale_factor=f, mode="bilinear", align_corners=False)   # net/resnet50_irn.py:36
            out["up_x_" + tag] = x.numpy().copy()
            out["up_f_" + tag] = np.int64(f)
            out["up_out_" + tag] = relu(up(x)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "trunk_ops.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None,
                    help="subset of: path affinity affinity_grad walk walk128 semseg instance cam_merge nets nets512 nets_scales msf pair_disp trunk_ops; or walk case names")
    a = ap.parse_args()
    _install_reference()
    torch.set_num_threads(os.cpu_count())
    sel = set(a.only) if a.only else None

    def want(k):
        return sel is None or k in sel

    if want("path"):
        gen_path_tables()
    if want("affinity"):
        gen_affinity()
    if want("affinity_grad"):
        gen_affinity_grad()
    walk_names = {c[0] for c in WALK_CASES}
    if want("walk") or (sel and sel & walk_names):
        gen_walk(only=(sel & walk_names) if sel and not want("walk") else None)
    if sel and "walk128" in sel:                      # never part of the default run (10 minutes)
        gen_walk(cases=WALK_BIG_CASES, fname="walk128.npz")
    if sel and "walk_voc" in sel:                     # never part of the default run (minutes of dense matmul each)
        gen_walk(cases=WALK_VOC_CASES, fname="walk_voc.npz")
    if want("semseg"):
        gen_semseg()
    if want("instance"):
        gen_instance()
    if want("cam_merge"):
        gen_cam_merge()
    if want("nets"):
        gen_nets()
    if want("nets512"):
        gen_nets512()
    if want("nets_scales"):
        gen_nets_scales()
    if want("msf"):
        gen_msf()
    if want("pair_disp"):
        gen_pair_disp()
    if want("trunk_ops"):
        gen_trunk_ops()


if __name__ == "__main__":
    main()
