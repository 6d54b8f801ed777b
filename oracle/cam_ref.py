"""CPU ORACLE for the CAM half of the metric — TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module).

The reference's ResNet-50 CAM forward (net/resnet50.py:17-108, net/resnet50_cam.py:55-70) restated as plain torch
functional ops over a STATE DICT with the reference's own keys (`resnet50.conv1.weight`, `resnet50.layer1.0.bn3.running_var`,
`classifier.weight`, ...): no module classes, nothing of the product's kernels.  Pinned on tests/golden/nets.npz, an output of
the reference itself (tests/golden/make_golden.py) — tests/test_oracle_golden.py."""
import torch
import torch.nn.functional as F

BLOCKS = (3, 4, 6, 3)                      # net/resnet50.py:113 resnet50(): Bottleneck, [3, 4, 6, 3]
CAM_STRIDES = (2, 2, 2, 1)                 # net/resnet50_cam.py:12


def _bn(x, sd, p):
    # FixedBatchNorm: always the running statistics (net/resnet50.py:11-14)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _unit(x, sd, p, stride):
    # Bottleneck.forward (net/resnet50.py:34-54); every dilation of the CAM trunk is 1
    y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    y = F.relu(_bn(F.conv2d(y, sd[p + ".conv2.weight"], None, stride, 1), sd, p + ".bn2"))
    y = _bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if p + ".downsample.0.weight" in sd:                       # _make_layer (net/resnet50.py:76-82)
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return F.relu(y + x)


def trunk(sd, x, strides=CAM_STRIDES, prefix="resnet50."):
    """conv1 / bn1 / relu / maxpool and the four stages (net/resnet50.py:94-103 without the pooling head)."""
    x = F.relu(_bn(F.conv2d(x, sd[prefix + "conv1.weight"], None, strides[0], 3), sd, prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (n, s) in enumerate(zip(BLOCKS, (1,) + tuple(strides[1:]))):
        for b in range(n):
            x = _unit(x, sd, "%slayer%d.%d" % (prefix, li + 1, b), s if b == 0 else 1)
    return x


def cam_forward(sd, x):
    """CAM.forward (net/resnet50_cam.py:55-70): x [2,3,H,W] = (image, h-flipped image) -> [20, ceil(H/16), ceil(W/16)]."""
    a = F.relu(F.conv2d(trunk(sd, x), sd["classifier.weight"]))
    return a[0] + a[1].flip(-1)


def flops_per_pair(h, w):
    """Multiply-add flops (2 per MAC) of `cam_forward` on one [2,3,h,w] pair: convolutions only."""
    def out(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    total = 0
    hh, ww = out(h, 7, 2, 3), out(w, 7, 2, 3)
    total += 2 * hh * ww * 64 * 3 * 49
    hh, ww = out(hh, 3, 2, 1), out(ww, 3, 2, 1)
    c_in = 64
    for planes, n, s in zip((64, 128, 256, 512), BLOCKS, (1,) + CAM_STRIDES[1:]):
        for b in range(n):
            st = s if b == 0 else 1
            total += 2 * hh * ww * c_in * planes                                  # conv1 at the input resolution
            ho, wo = out(hh, 3, st, 1), out(ww, 3, st, 1)
            total += 2 * ho * wo * planes * planes * 9 + 2 * ho * wo * planes * planes * 4
            if b == 0:
                total += 2 * ho * wo * c_in * planes * 4
            hh, ww, c_in = ho, wo, planes * 4
    total += 2 * hh * ww * 2048 * 20
    return 2 * total                                                               # two rows per pair
