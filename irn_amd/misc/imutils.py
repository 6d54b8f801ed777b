"""Image helpers the hot path needs (API mirror of the corresponding reference misc/imutils.py
functions; the augmentation / CRF / colouring helpers of that file are training-side and out of scope).
"""
import numpy as np
from PIL import Image


def pil_resize(img, size, order):
    """misc/imutils.py:8-17 — PIL bicubic (order 3) or nearest (order 0) resize to (h, w)."""
    if size[0] == img.shape[0] and size[1] == img.shape[1]:
        return img
    resample = {3: Image.BICUBIC, 0: Image.NEAREST}[order]
    return np.asarray(Image.fromarray(img).resize(size[::-1], resample))


def pil_rescale(img, scale, order):
    """misc/imutils.py:19-22."""
    h, w = img.shape[:2]
    return pil_resize(img, (int(np.round(h * scale)), int(np.round(w * scale))), order)


def HWC_to_CHW(img):
    return np.transpose(img, (2, 0, 1))


def get_strided_size(orig_size, stride):
    """misc/imutils.py:173-174: ceil(size / stride) per axis."""
    return ((orig_size[0] - 1) // stride + 1, (orig_size[1] - 1) // stride + 1)


def get_strided_up_size(orig_size, stride):
    """misc/imutils.py:177-179."""
    s = get_strided_size(orig_size, stride)
    return s[0] * stride, s[1] * stride


def compress_range(arr):
    """misc/imutils.py:182-190: renumber the distinct values to 0..K-1 in ascending order."""
    uniq = np.unique(arr)
    lut = np.zeros(int(uniq.max()) + 1, np.int32)
    lut[uniq] = np.arange(uniq.shape[0])
    out = lut[arr]
    return out - np.min(out)
