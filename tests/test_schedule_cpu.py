"""Host-side checks of the walk's polynomial schedule (no GPU): the series irn_power_series returns IS lambda^n on
the operator's spectrum to the stated bound, and the three-term recurrence the kernels run — restated here in numpy
on the oracle's stencil sweep with the state rounded to fp32 after every step, as the kernels store it — reproduces
the oracle's x . T^n (reference misc/indexing.py:132-139, :164) as closely as the plain fp32-state iteration does."""
import ctypes as C

import numpy as np
import pytest

from irn_amd import _lib, synth
from oracle import irn_oracle as O


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
    assert _series(256)[0] == 84 and _series(256, 9)[0] > 84 and _series(16)[1] is False
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
