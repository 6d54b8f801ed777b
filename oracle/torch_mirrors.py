"""CPU/GPU ORACLE, torch-op form — TEST INFRASTRUCTURE, NOT PRODUCT CODE (only ``tests/`` may import this module).

The reference's own op sequences restated on torch tensors of whatever device they live on, for tests that compare the
HIP kernels with "what the reference's ops give on this very device"."""
import torch
import torch.nn.functional as F


def merge_scales_torch(outputs, size, label):
    """The multi-scale merge of reference step/make_cam.py:32-52: per-scale bilinear resize to the stride-4 size and to
    the stride-16-rounded full size, sums over the scales, crop, present classes, per-channel max-normalisation.
    Pinned on the reference's output in tests/golden/cam_merge.npz (tests/test_host_logic.py)."""
    size = (int(size[0]), int(size[1]))
    strided_size = ((size[0] - 1) // 4 + 1, (size[1] - 1) // 4 + 1)                        # misc/imutils.py get_strided_size
    strided_up_size = (((size[0] - 1) // 16 + 1) * 16, ((size[1] - 1) // 16 + 1) * 16)     # get_strided_up_size
    strided_cam = torch.sum(torch.stack(
        [F.interpolate(o[None], strided_size, mode="bilinear", align_corners=False)[0] for o in outputs]), 0)
    highres = torch.sum(torch.stack(
        [F.interpolate(o[:, None], strided_up_size, mode="bilinear", align_corners=False) for o in outputs], 0), 0)
    highres = highres[:, 0, :size[0], :size[1]]
    valid_cat = torch.nonzero(label)[:, 0]
    # per-channel spatial max: the reference calls F.adaptive_max_pool2d(x, (1, 1)); amax is the same value
    strided_cam = strided_cam[valid_cat]
    strided_cam = strided_cam / (strided_cam.amax(dim=(1, 2), keepdim=True) + 1e-5)
    highres = highres[valid_cat]
    highres = highres / (highres.amax(dim=(1, 2), keepdim=True) + 1e-5)
    return valid_cat, strided_cam, highres
