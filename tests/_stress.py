"""Adversarial inputs for the walk's polynomial schedule: the synthetic fields of irn_amd/synth.py
are smooth, and the series' error model depends on the operator's SPECTRUM (how much of it sits near |lambda| = 1, where
the three-term recurrence amplifies rounding) — so these fields push the spectrum around instead: no edges at all (one
uniform averaging operator), edges everywhere (T ~ I: the whole spectrum at 1), 0/1 edges (exactly zero and exactly
one weights, isolated pixels, disconnected components), a one-pixel wall (two nearly decoupled halves), other beta, and
white-noise CAMs (energy in every eigenvector, not only the smooth ones).  Used by the CPU model test and the GPU tests."""
import numpy as np

from irn_amd import synth


def cases():
    """-> list of (name, beta, make(h, w, c, seed) -> (edge [h,w] f32, cam [c,h,w] f32))."""
    def noise(c, h, w, seed):
        return np.random.RandomState(seed).rand(c, h, w).astype(np.float32)

    def const(v):
        return lambda h, w, c, seed: (np.full((h, w), v, np.float32), noise(c, h, w, seed))

    def bernoulli(p):
        def make(h, w, c, seed):
            rs = np.random.RandomState(seed + 17)
            return (rs.rand(h, w) < p).astype(np.float32), noise(c, h, w, seed)
        return make

    def wall(h, w, c, seed):
        e = np.full((h, w), 0.05, np.float32)
        e[:, w // 2] = 1.0
        return e, noise(c, h, w, seed)

    def smooth_noise(h, w, c, seed):
        return synth.edge_field(h, w, seed=seed), noise(c, h, w, seed)

    def smooth_blobs(h, w, c, seed):
        return synth.edge_field(h, w, seed=seed), synth.cam_blobs(c, h, w, seed=seed)

    return [("edge=0", 10, const(0.0)), ("edge=0.999", 10, const(0.999)), ("bernoulli p=0.1", 10, bernoulli(0.1)),
            ("bernoulli p=0.5", 10, bernoulli(0.5)), ("wall", 10, wall), ("smooth, noise cam", 10, smooth_noise),
            ("beta=1", 1, smooth_noise), ("beta=8", 8, smooth_blobs), ("beta=20", 20, smooth_noise)]


def argmax_mismatch_is_tie(got, exact, tol):
    """Grid argmax over channels: got may differ from exact only where exact's two best channels are within `tol`
    (both inputs [C,h,w], normalised by exact's maximum).  -> number of differing pixels."""
    if got.shape[0] == 1:
        return 0
    a, b = np.argmax(got, 0), np.argmax(exact, 0)
    diff = a != b
    n = int(diff.sum())
    if n:
        srt = np.sort(exact[:, diff] / max(float(exact.max()), 1e-30), axis=0)
        gap = srt[-1] - srt[-2]
        assert float(gap.max()) < tol, "argmax differs at a pixel whose top-2 gap is %.3g" % float(gap.max())
    return n
