"""Label parity, measured: how many pixels of a label map differ from the oracle's, and why each of them may.

The product's walk and the reference's (or the fp64 oracle's) differ by fp32 rounding, <= 1e-4 on the normalised
scores by the north star's own bar.  A label (argmax over {background threshold, class scores}, reference
step/make_sem_seg_labels.py:43-49) can therefore differ ONLY where the two best entries of the oracle's score stack are
closer than that bar, and then only by picking the runner-up.  `label_mismatches` counts the differing pixels and
asserts exactly that for every one of them — no fractional allowance."""
import numpy as np

TIE_TOL = 1e-4      # north star: scores within 1e-4 max-abs => only ties at that level can flip an argmax


def label_mismatches(got, want, up, bg_thres, lut=None, what=""):
    """got / want: [H,W] maps (labels through `lut`, or raw argmax indices when lut is None); up: the oracle's
    normalised scores [C,H,W] (oracle.irn_oracle.sem_seg_epilogue()[0]).  Returns (count, largest top-2 gap among them)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    diff = got != want
    n = int(diff.sum())
    if n == 0:
        return 0, 0.0
    stack = np.concatenate([np.full((1,) + up.shape[1:], bg_thres, np.float32), np.asarray(up, np.float32)], 0)[:, diff]
    order = np.argsort(stack, axis=0, kind="stable")
    best, second = order[-1], order[-2]
    cols = np.arange(stack.shape[1])
    gap = stack[best, cols] - stack[second, cols]
    assert float(gap.max()) < TIE_TOL, "%s: %d pixels differ and one is NOT a tie: top-2 gap %.3g" % (what, n, float(gap.max()))
    # ... and the product picked the other one of the tied pair (or another entry inside the tie band)
    pick = got[diff].astype(np.int64)
    if lut is not None:
        lut = np.asarray(lut)
        in_band = stack >= (stack[best, cols] - TIE_TOL)[None]
        ok = np.array([pick[j] in set(lut[np.nonzero(in_band[:, j])[0]].tolist()) for j in range(n)])
    else:
        ok = stack[pick, cols] >= stack[best, cols] - TIE_TOL
    assert bool(ok.all()), "%s: %d of %d differing pixels chose an entry outside the tie band" % (what, int((~ok).sum()), n)
    return n, float(gap.max())
