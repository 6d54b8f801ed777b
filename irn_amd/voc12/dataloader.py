"""Multi-scale inference dataset of the hot path — API mirror of the reference
voc12/dataloader.py pieces the label-generation steps use:

    decode_int_filename (:24-26), load_img_name_list (:56-60), TorchvisionNormalize (:65-78),
    VOC12ClassificationDatasetMSF (:175-205)

JPEGs are decoded with PIL (the reference's imageio call decodes through PIL as well).  The
image-level labels come from ``cls_labels.npy`` ({int id -> float32[20]}); pass ``cls_labels=`` or
keep the file next to the image lists as the reference does.  Training datasets / augmentation of
that file are out of scope.
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from ..misc import imutils

IMG_FOLDER_NAME = "JPEGImages"
N_CAT = 20


def decode_int_filename(int_filename):
    s = str(int(int_filename))
    return s[:4] + "_" + s[4:]


def load_img_name_list(dataset_path):
    """Reads '2007_000032'-style ids as integers 2007000032.  (The reference's
    np.loadtxt(dtype=int32) rejects the underscore under numpy >= 2; parse explicitly.)"""
    with open(dataset_path) as f:
        return np.asarray([int(line.strip().replace("_", "")) for line in f if line.strip()], np.int64)


def get_img_path(img_name, voc12_root):
    if not isinstance(img_name, str):
        img_name = decode_int_filename(img_name)
    return os.path.join(voc12_root, IMG_FOLDER_NAME, img_name + ".jpg")


class TorchvisionNormalize:
    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.mean = mean
        self.std = std

    def __call__(self, img):
        arr = np.asarray(img)
        out = np.empty_like(arr, np.float32)
        for c in range(3):
            out[..., c] = (arr[..., c] / 255. - self.mean[c]) / self.std[c]
        return out


class VOC12ClassificationDatasetMSF(Dataset):
    """item -> {'name': str, 'img': [scales x [2,3,Hs,Ws]] (image + h-flip; a bare array when there
    is a single scale), 'size': (H, W), 'label': float32[20]}  (voc12/dataloader.py:175-205).

    ``raw=True`` hands over the decoded image as a uint8 tensor [H,W,3] instead: the steps then build the
    per-scale pairs on the GPU (`irn_amd.ops.msf_pack`, bit-identical to the PIL/numpy loop below), which
    cuts the host-to-device traffic of a 4-scale item from 47 MB of floats to 0.8 MB of bytes and frees the
    loader workers of the bicubic resizes."""

    def __init__(self, img_name_list_path, voc12_root, img_normal=TorchvisionNormalize(), scales=(1.0,),
                 cls_labels=None, raw=False, skip_image=None):
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.voc12_root = voc12_root
        self.img_normal = img_normal
        self.scales = scales
        self.raw = raw
        # raw mode only: `skip_image(name) -> bool` lets a step say that it will not need an image's pixels (the label step that
        # runs second takes the boundary / displacement maps from device memory): the item then carries an EMPTY uint8 tensor
        # and the size read from the file header — no JPEG decode, no upload
        self.skip_image = skip_image
        if cls_labels is None:
            path = os.path.join(os.path.dirname(os.path.abspath(img_name_list_path)), "cls_labels.npy")
            cls_labels = np.load(path, allow_pickle=True).item()
        self.label_list = np.array([cls_labels[int(n)] for n in self.img_name_list], np.float32)

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        name_str = decode_int_filename(self.img_name_list[idx])
        if self.raw and self.skip_image is not None and self.skip_image(name_str):
            with Image.open(get_img_path(name_str, self.voc12_root)) as im:
                w, h = im.size                       # header only
            return {"name": name_str, "img": torch.empty((0, 0, 3), dtype=torch.uint8), "size": (h, w),
                    "label": torch.from_numpy(self.label_list[idx])}
        img = np.asarray(Image.open(get_img_path(name_str, self.voc12_root)).convert("RGB"))
        if self.raw:
            return {"name": name_str, "img": torch.from_numpy(np.array(img)),
                    "size": (img.shape[0], img.shape[1]), "label": torch.from_numpy(self.label_list[idx])}
        ms = []
        for s in self.scales:
            s_img = img if s == 1 else imutils.pil_rescale(img, s, order=3)
            s_img = imutils.HWC_to_CHW(self.img_normal(s_img))
            ms.append(np.stack([s_img, np.flip(s_img, -1)], axis=0))
        if len(self.scales) == 1:
            ms = ms[0]
        return {"name": name_str, "img": ms, "size": (img.shape[0], img.shape[1]),
                "label": torch.from_numpy(self.label_list[idx])}
