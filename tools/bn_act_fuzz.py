#!/usr/bin/env python3
"""Randomised check of irn_bn_act's flat addressing (multiply-shift divisions by H*W and C, pieces straddling planes):
random [N, C, H, W] shapes up to 2^27 elements against the exact expression, plus one tensor just under 2^31 elements.
    python tools/bn_act_fuzz.py [n_cases]          (needs a GPU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from irn_amd import ops

dev = torch.device("cuda", 0)
rng = np.random.RandomState(int(os.environ.get("SEED", "1")))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for it in range(n_cases):
    c = int(rng.choice([1, 2, 3, 5, 7, 20, 32, 63, 64, 65, 256, 257, 1000, 2048]))
    h, w = int(rng.randint(1, 300)), int(rng.randint(1, 300))
    n = int(rng.randint(1, 5))
    while n * c * h * w > 2 ** 27:
        h = max(1, h // 2)
    x = torch.randn(n, c, h, w, device=dev)
    res = torch.randn(n, c, h, w, device=dev) if rng.rand() < 0.5 else None
    aff = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)) if res is not None and rng.rand() < 0.5 else None
    s, b = torch.rand(c, device=dev) * 2 - 0.5, torch.randn(c, device=dev)
    relu = bool(rng.rand() < 0.5)
    v = (1, c, 1, 1)
    want = (x.double() * s.double().view(v) + b.double().view(v)).float()
    if res is not None:
        r = res if aff is None else (res.double() * aff[0].double().view(v) + aff[1].double().view(v)).float()
        want = want + r
    if relu:
        want = want.clamp_min(0)
    got = ops.bn_act_(x, s, b, res, relu, aff)
    if not torch.equal(got, want):
        bad += 1
        print("MISMATCH", (n, c, h, w), res is not None, aff is not None, relu, float((got - want).abs().max()))
print("%d cases, %d mismatches" % (n_cases, bad))
# just under 2^31 elements in one call, and the Python wrapper's split above it
for shape in ((1, 64, 4096, 8191), (3, 64, 4096, 4096)):
    x = torch.ones(shape, device=dev)
    s, b = torch.arange(64, device=dev, dtype=torch.float32), torch.ones(64, device=dev)
    ops.bn_act_(x, s, b, None, False)
    ok = bool((x[:, 5] == 6).all()) and bool((x[-1, 63] == 64).all()) and bool((x[0, 0] == 1).all())
    print(shape, "%.2f G elements:" % (x.numel() / 2 ** 30), "ok" if ok else "WRONG")
    del x
