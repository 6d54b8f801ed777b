"""Host-side checks of the walk's polynomial schedule (no GPU): the series irn_power_series returns IS lambda^n on
the operator's spectrum to the stated bound, and the three-term recurrence the kernels run — restated here in numpy
on the oracle's stencil sweep with the state rounded to fp32 after every step, as the kernels store it — reproduces
the oracle's x . T^n (reference misc/indexing.py:132-139, :164) as closely as the plain fp32-state iteration does."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from irn_amd import _lib, synth
from oracle import irn_oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _series(n, tol_exp=7):
    k, rec = C.c_int(), C.c_int()
    coef = (C.c_double * (n + 1))()
    _lib.check(_lib.lib.irn_power_series(n, tol_exp, coef, n + 1, C.byref(k), C.byref(rec)))
    return k.value, bool(rec.value), np.array(coef[:k.value + 1])


@pytest.mark.parametrize("n", [0, 1, 4, 7, 8, 16, 32, 64, 128, 256, 512, 1024])
def test_series_is_the_power_on_the_spectrum(n):
    k, rec, c = _series(n)
    lam = np.linspace(-1.0, 1.0, 4001)
    if not rec:
        assert k == n and c[-1] == 1.0 and np.all(c[:-1] == 0.0)
        return
    assert k + 2 < n and abs(c.sum() - 1.0) < 1e-12 and np.all(c >= 0.0)
    assert np.all(c[(n & 1) ^ 1::2] == 0.0)                       # only the powers' own parity
    assert np.abs(np.polynomial.chebyshev.chebval(lam, c) - lam ** n).max() <= 2.5e-7


def test_known_lengths_and_bad_arguments():
    assert _series(256)[0] == 84 and _series(256, 6)[0] == 78 and _series(256, 9)[0] > 84 and _series(16)[1] is False
    k, rec = C.c_int(), C.c_int()
    assert _lib.lib.irn_power_series(-1, 7, None, 0, C.byref(k), C.byref(rec)) == 1
    assert _lib.lib.irn_power_series(256, 3, None, 0, C.byref(k), C.byref(rec)) == 1
    small = (C.c_double * 4)()
    assert _lib.lib.irn_power_series(256, 7, small, 4, C.byref(k), C.byref(rec)) == 1


@pytest.mark.parametrize("r,h,w,c", [(5, 40, 48, 3), (10, 36, 44, 1)])
def test_recurrence_model_matches_the_oracle_power(r, h, w, c):
    edge, cam = synth.edge_field(h, w, seed=11), synth.cam_blobs(c, h, w, seed=11)
    dirs, wts = O.stencil_weights(edge, r, 10)
    deg = O.stencil_degree(dirs, wts)
    w64 = wts.astype(np.float64)
    x0 = (cam * (1 - edge)).astype(np.float32).astype(np.float64)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)       # noqa: E731
    exact = O.propagate_to_edge_stencil(cam, edge, r, 10, 8)[:, 0]
    plain = x0.copy()
    for _ in range(256):
        plain = f32(O.stencil_sweep(plain, dirs, w64, deg))
    k, rec, coef = _series(256)
    assert rec
    prev, y = x0, f32(O.stencil_sweep(x0, dirs, w64, deg))
    s = f32(coef[0] * x0)
    s = f32(s + np.float32(coef[1]) * y)
    for t in range(1, k):
        y, prev = f32(2.0 * O.stencil_sweep(y, dirs, w64, deg) - prev), y
        s = f32(s + float(np.float32(coef[t + 1])) * y)
    e_series, e_plain = np.abs(s - exact).max(), np.abs(plain - exact).max()
    assert e_series <= 2e-6 and e_plain <= 4e-6, (e_series, e_plain)
    assert e_series <= 2.0 * e_plain + 5e-7
    assert np.array_equal(np.argmax(s, 0), np.argmax(exact, 0))


def _model(edge, cam, r, beta, tol_exp=7):
    """(exact fp64 x.T^256, plain iteration with fp32 state, the kernels' recurrence with fp32 state)."""
    dirs, wts = O.stencil_weights(edge, r, beta)
    deg = O.stencil_degree(dirs, wts)
    w64 = wts.astype(np.float64)
    x0 = (cam * (1 - edge)).astype(np.float32).astype(np.float64)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)       # noqa: E731
    exact = O.propagate_to_edge_stencil(cam, edge, r, beta, 8)[:, 0]
    plain = x0.copy()
    for _ in range(256):
        plain = f32(O.stencil_sweep(plain, dirs, w64, deg))
    k, rec, coef = _series(256, tol_exp)
    prev, y = x0, f32(O.stencil_sweep(x0, dirs, w64, deg))
    s = f32(float(np.float32(coef[0])) * x0)
    s = f32(s + float(np.float32(coef[1])) * y)
    for t in range(1, k):
        y, prev = f32(2.0 * O.stencil_sweep(y, dirs, w64, deg) - prev), y
        s = f32(s + float(np.float32(coef[t + 1])) * y)
    return exact, plain, s


@pytest.mark.parametrize("case", range(9))
def test_recurrence_model_on_adversarial_fields(case):
    """The schedule where its maths is stressed (tests/_stress.py): no edges, edges everywhere, 0/1 edges, a wall, other
    beta, white-noise CAMs.  The recurrence with the kernels' fp32 state stays within 2e-6 of the exact product (1e-5 is
    the GPU tests' bar, 1e-4 the north star's) and within 1e-4 after the epilogue's division by the maximum; it is NOT
    always closer than the plain iteration — on fields whose operator has a flat spectrum it is several times further
    (still two orders inside the bar), which is what the printed pairs record."""
    import _stress
    name, beta, make = _stress.cases()[case]
    edge, cam = make(40, 48, 2, 31 + case)
    exact, plain, s = _model(edge, cam, 5, beta)
    e_s, e_p = np.abs(s - exact).max(), np.abs(plain - exact).max()
    scale = max(float(exact.max()), 1e-30)
    print("%-20s radius 5: series %.2e  plain %.2e  (normalised %.2e / %.2e)" % (name, e_s, e_p, e_s / scale, e_p / scale))
    assert e_s <= 2e-6 and e_p <= 5e-6
    assert e_s / scale <= 1e-4
    _stress.argmax_mismatch_is_tie(s, exact, 2e-5)
    # the looser truncation bound offered as an option (tol 1e-6, 78 applications at n = 256)
    _, _, s6 = _model(edge, cam, 5, beta, tol_exp=6) if case in (0, 3, 6) else (None, None, None)
    if s6 is not None:
        e6 = np.abs(s6 - exact).max()
        print("%-20s radius 5: series at 1e-6 %.2e" % (name, e6))
        assert e6 <= 5e-6 and e6 / scale <= 1e-4
