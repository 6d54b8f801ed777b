"""Seeded synthetic VOC-shaped inputs for the pseudo-label hot path.

There is no dataset and no trained checkpoint in the build or GPU containers, so tests, golden
fixtures and bench.py all draw their inputs from these generators (numpy ``RandomState`` only, so
the same seed gives the same tensor on every box):

* ``edge_field``         boundary map in (0,1) as ``EdgeDisplacement`` would emit it
                         (reference net/resnet50_irn.py:231): smooth low field + thin closed ridges
* ``cam_blobs``          K max-normalised activation blobs, the shape ``make_cam`` stores under
                         ``cam`` (reference step/make_cam.py:46-48)
* ``displacement_field`` dp[0]=dy, dp[1]=dx pointing at the nearest of m attractors
                         (reference step/make_ins_seg_labels.py:124)
* ``voc_num_classes``    K drawn from the VOC12 image-level label histogram
"""
import numpy as np
from scipy import ndimage


def _rng(seed):
    return np.random.RandomState(int(seed) & 0x7FFFFFFF)


def edge_field(h, w, seed=0):
    """float32 [h, w] in (0, 1): sigmoid(2 z) of a blurred normal field, plus 2-4 closed ridges
    whose crest is 0.9-1.0 (class boundaries)."""
    r = _rng(seed)
    z = ndimage.gaussian_filter(r.randn(h, w), sigma=2.0, mode="reflect") * 3.0
    e = 1.0 / (1.0 + np.exp(-2.0 * z))
    e = 0.6 * e                                  # interior mostly "no boundary"
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for _ in range(r.randint(2, 5)):
        cy, cx = r.uniform(0.15, 0.85) * h, r.uniform(0.15, 0.85) * w
        ry, rx = r.uniform(0.12, 0.4) * h, r.uniform(0.12, 0.4) * w
        phase, wob = r.uniform(0, 2 * np.pi), r.uniform(0.0, 0.2)
        ang = np.arctan2(yy - cy, xx - cx)
        rad = np.sqrt(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2)
        d = np.abs(rad - (1.0 + wob * np.sin(3 * ang + phase))) * min(ry, rx)
        crest = r.uniform(0.9, 1.0)
        ridge = crest * np.exp(-(d / 0.8) ** 2)
        e = np.maximum(e, ridge)
    return np.clip(e, 1e-4, 1.0 - 1e-4).astype(np.float32)


def cam_blobs(k, h, w, seed=0):
    """float32 [k, h, w], each channel a sum of 1-2 Gaussian blobs, max-normalised to 1."""
    r = _rng(seed + 7919)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((k, h, w), np.float32)
    s = min(h, w) / 128.0
    for c in range(k):
        acc = np.zeros((h, w))
        for _ in range(r.randint(1, 3)):
            cy, cx = r.uniform(0.1, 0.9) * h, r.uniform(0.1, 0.9) * w
            sg = r.uniform(8.0, 24.0) * max(s, 0.15)
            acc += r.uniform(0.5, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
        acc += 0.02 * r.rand(h, w)
        out[c] = (acc / (acc.max() + 1e-5)).astype(np.float32)
    return out


def displacement_field(h, w, seed=0, strength=0.1, noise=0.05):
    """float32 [2, h, w]: strength * (nearest attractor - p) + N(0, noise); 1-4 attractors."""
    r = _rng(seed + 104729)
    m = r.randint(1, 5)
    att = np.stack([r.uniform(0.1, 0.9, m) * (h - 1), r.uniform(0.1, 0.9, m) * (w - 1)], 1)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d2 = (yy[None] - att[:, 0, None, None]) ** 2 + (xx[None] - att[:, 1, None, None]) ** 2
    near = np.argmin(d2, 0)
    dy = strength * (att[near, 0] - yy) + noise * r.randn(h, w)
    dx = strength * (att[near, 1] - xx) + noise * r.randn(h, w)
    return np.stack([dy, dx]).astype(np.float32)


_VOC_K_HIST = ((1, 0.60), (2, 0.29), (3, 0.09), (4, 0.02))   # image-level label count, voc12/cls_labels.npy


def voc_num_classes(seed):
    u = _rng(seed + 15485863).rand()
    acc = 0.0
    for k, p in _VOC_K_HIST:
        acc += p
        if u < acc:
            return k
    return _VOC_K_HIST[-1][0]


def voc_keys(k, seed):
    """Sorted 0-based class ids, the ``keys`` entry of a CAM dict (step/make_cam.py:46)."""
    return np.sort(_rng(seed + 32452843).choice(20, k, replace=False)).astype(np.int64)


# (H, W) of VOC12 JPEGs and their share: the longer side is almost always 500 px; 500x375 landscape dominates, then its
# portrait twin and the 3:2 formats.  No dataset is available in these containers, so this is a coarse stand-in for the
# size histogram of JPEGImages/ (SURVEY.md §8d "ragged variant": 94x125, 125x84, ... stride-4 grids), not a measurement.
VOC_SIZES = (((375, 500), 0.50), ((500, 375), 0.14), ((333, 500), 0.12), ((500, 333), 0.05), ((334, 500), 0.04),
             ((332, 500), 0.03), ((374, 500), 0.03), ((500, 500), 0.02), ((281, 500), 0.02), ((400, 500), 0.02),
             ((500, 400), 0.01), ((357, 500), 0.01), ((442, 500), 0.01))


def voc_image_size(seed):
    """(H, W) of one synthetic VOC12-shaped image, drawn from `VOC_SIZES`."""
    u = _rng(seed + 77003).rand()
    acc = 0.0
    for size, p in VOC_SIZES:
        acc += p
        if u < acc:
            return size
    return VOC_SIZES[0][0]


def grid_of(size, stride=4):
    """Stride-4 grid of an image: ((H - 1) // 4 + 1, (W - 1) // 4 + 1) (reference net/resnet50_irn.py:224)."""
    return (size[0] - 1) // stride + 1, (size[1] - 1) // stride + 1


def photo(h, w, seed=0):
    """uint8 [h,w,3] photo-like image: smooth colour fields, hard-edged saturated rectangles (the bicubic
    overshoot there exercises the 0/255 clipping) and sensor-like noise."""
    rng = _rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        fy, fx, ph = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0, 6.28)
        img[..., c] = 128 + 90 * np.sin(fy * yy / h * 6.28 + ph) * np.cos(fx * xx / w * 6.28)
    for _ in range(4):
        y0, x0 = int(rng.randint(0, max(h - 2, 1))), int(rng.randint(0, max(w - 2, 1)))
        y1, x1 = y0 + int(rng.randint(1, max(h // 3, 2))), x0 + int(rng.randint(1, max(w // 3, 2)))
        img[y0:y1, x0:x1] = rng.choice([0.0, 255.0], 3)
    img += rng.normal(0, 6, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)



def image_pair(h, w, seed=0, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """float32 [2,3,h,w]: a `photo` normalised like TorchvisionNormalize (reference voc12/dataloader.py:65-78),
    channel-major, followed by its horizontal flip — the network input of one scale of a dataset item
    (voc12/dataloader.py:196-199).  Regenerated from the seed wherever it is needed (6 MB at 512^2: not committed)."""
    img = photo(h, w, seed)
    out = np.empty((3, h, w), np.float32)
    for c in range(3):
        out[c] = (img[..., c] / 255. - mean[c]) / std[c]
    return np.stack([out, out[..., ::-1]]).astype(np.float32)
