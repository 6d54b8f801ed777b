"""GPU parity of the walk's polynomial schedule (round 3): x . T^n evaluated as a truncated Chebyshev series
(84 operator applications instead of 256 at exp_times = 8 with the default truncation bound 1e-7, 78 at 1e-6; irn_amd/csrc/walk.hip header, include/irn_hip.h).

What has to hold: (a) the accelerated walk is as close to the fp64 oracle of the reference's operator
(misc/indexing.py:141-165) as the plain iteration — both far inside the 1e-4 bar — with identical argmax, on every
kernel variant; (b) "accel" = 0 is the plain iteration; (c) the recurrence's private terms survive every storage
path — LDS for the first channels of a job, the workspace for the rest, and the write-back between the launches of a
walk cut into several; (d) the reference's own 128x128 outputs."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import irn_oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stress  # noqa: E402

pytestmark = pytest.mark.gpu

TOL_REF = 1e-4
TOL_F64 = 1e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


def _walker(r, variant=2, **opts):
    from irn_amd.misc import indexing
    wk = indexing.RandomWalk(r, _dev())
    wk.set_option("variant", variant)
    for k, v in opts.items():
        wk.set_option(k, v)
    return wk


def _inputs(shapes, seed0):
    from irn_amd import synth
    edges = [torch.from_numpy(synth.edge_field(h, w, seed=seed0 + i)).to(_dev()) for i, (h, w, c) in enumerate(shapes)]
    cams = [torch.from_numpy(synth.cam_blobs(c, h, w, seed=seed0 + i)).to(_dev()) for i, (h, w, c) in enumerate(shapes)]
    return edges, cams


def test_schedule_lengths():
    wk = _walker(10)
    assert wk.steps(256) == 84 and wk.steps(16) == 16 and wk.steps(0) == 0
    assert wk.steps(128) < 70 and wk.steps(1024) < 200
    wk.set_option("accel_tol_exp", 6)
    assert wk.steps(256) == 78
    wk.set_option("accel", 0)
    assert wk.steps(256) == 256
    wk.set_option("accel", 1)
    wk.set_option("accel_tol_exp", 9)
    assert 84 < wk.steps(256) <= 100
    wk.close()


@pytest.mark.parametrize("r,shapes", [
    (10, [(128, 128, 1), (128, 128, 2), (128, 128, 3), (94, 125, 5), (40, 52, 9)]),
    (5, [(128, 128, 1), (128, 128, 2), (94, 125, 3), (47, 31, 7), (130, 66, 4)]),
])
@pytest.mark.parametrize("variant", [2, 1, 0])
def test_accelerated_walk_vs_fp64_oracle_and_plain_iteration(r, shapes, variant):
    """Every kernel variant, 2^8 sweeps: series and plain powers against the fp64 oracle of the operator, against each
    other, and the measured errors side by side (the series must not be the worse of the two by more than rounding)."""
    from irn_amd import synth
    edges, cams = _inputs(shapes, 900)
    fast = _walker(r, variant)
    plain = _walker(r, variant, accel=0)
    a = fast(edges, cams, beta=10, exp_times=8)
    fast.check()
    b = plain(edges, cams, beta=10, exp_times=8)
    plain.check()
    worst = (0.0, 0.0)
    for i, (h, w, c) in enumerate(shapes):
        st = O.propagate_to_edge_stencil(synth.cam_blobs(c, h, w, seed=900 + i), synth.edge_field(h, w, seed=900 + i), r, 10, 8)
        ea = np.abs(a[i].cpu().numpy() - st).max()
        eb = np.abs(b[i].cpu().numpy() - st).max()
        worst = (max(worst[0], ea), max(worst[1], eb))
        assert ea <= TOL_F64 and eb <= TOL_F64, (shapes[i], ea, eb)
        assert (a[i] - b[i]).abs().max().item() <= 4e-6, shapes[i]
        assert np.array_equal(np.argmax(a[i].cpu().numpy()[:, 0], 0), np.argmax(st[:, 0], 0)), shapes[i]
    print("variant %d radius %d: max |series - fp64| %.2e, max |plain - fp64| %.2e" % (variant, r, worst[0], worst[1]))
    assert worst[0] <= 2.0 * worst[1] + 5e-7
    # the looser truncation bound (option accel_tol_exp = 6: 78 applications, +7 %): inside every numeric bar, but its grid
    # argmax may differ from the oracle's at exact-tie level (one pixel of the 128x128 two-class image at radius 5 in
    # round 4's session 1) — counted here, and the reason it is an option and not the default
    loose = _walker(r, variant, accel_tol_exp=6)
    c6 = loose(edges, cams, beta=10, exp_times=8)
    loose.check()
    flips = 0
    for i, (h, w, c) in enumerate(shapes):
        st = O.propagate_to_edge_stencil(synth.cam_blobs(c, h, w, seed=900 + i), synth.edge_field(h, w, seed=900 + i), r, 10, 8)
        got = c6[i].cpu().numpy()
        assert np.abs(got - st).max() <= TOL_F64, shapes[i]
        flips += _stress.argmax_mismatch_is_tie(got[:, 0], st[:, 0], 2e-5)
    print("variant %d radius %d, truncation bound 1e-6: %d grid-argmax pixel(s) differ from the oracle's, all ties below 2e-5" % (variant, r, flips))
    fast.close()
    plain.close()
    loose.close()


@pytest.mark.parametrize("r,shapes", [
    (10, [(128, 128, 1), (128, 128, 3), (94, 125, 2), (64, 64, 55)]),       # 55 channels: 7 of them beyond the LDS-held 48
    (5, [(128, 128, 2), (94, 125, 14), (125, 94, 1), (60, 200, 3)]),        # 14 channels: 2 beyond the LDS-held 12
])
def test_series_one_launch_equals_launch_per_step_bitwise(r, shapes):
    """The recurrence carries {y_{t-1}, s_t} per pixel: in LDS inside a launch, through the workspace between launches.
    One launch, a launch per step and launches of 5 steps must agree bit for bit, repeatedly on one workspace."""
    edges, cams = _inputs(shapes, 700)
    per = _walker(r, sweeps_per_launch=1)
    ref = [o.clone() for o in per(edges, cams, beta=10, exp_times=8)]
    per.check()
    one = _walker(r)
    for rep in range(2):
        out = one(edges, cams, beta=10, exp_times=8)
        one.check()
        for i in range(len(shapes)):
            assert torch.equal(out[i], ref[i]), (r, shapes[i], rep)
    five = _walker(r, sweeps_per_launch=5)
    out = five(edges, cams, beta=10, exp_times=8)
    five.check()
    for i in range(len(shapes)):
        assert torch.equal(out[i], ref[i]), (r, shapes[i])
    gen = _walker(r, variant=0)
    g = gen(edges, cams, beta=10, exp_times=8)
    for i in range(len(shapes)):
        assert (out[i] - g[i]).abs().max().item() <= 3e-6, shapes[i]
    for wkr in (per, one, five, gen):
        wkr.close()


def test_series_batch_equals_single_and_instance_split():
    """Ragged multi-round batch == single-image runs bit for bit; instance split channels through the series."""
    from irn_amd import synth
    shapes = [(128, 128, 1 + (i * 7) % 4) for i in range(12)] + [(94, 125, 6), (125, 84, 2)]
    edges, cams = _inputs(shapes, 40)
    wk = _walker(10)
    batch = [o.clone() for o in wk(edges, cams, beta=10, exp_times=8)]
    wk.check()
    for i in (0, 3, 12, 13):
        single = wk([edges[i]], [cams[i]], beta=10, exp_times=8)[0]
        wk.check()
        assert torch.equal(single, batch[i]), i
    h, w, k = 96, 112, 3
    edge = torch.from_numpy(synth.edge_field(h, w, seed=5)).to(_dev())
    cam = torch.from_numpy(synth.cam_blobs(2, h, w, seed=5)).to(_dev())
    inst = torch.from_numpy((np.arange(h * w).reshape(h, w) // 7 % k).astype(np.int32)).to(_dev())
    rw = wk([edge], [cam], beta=10, exp_times=8, inst_maps=[inst], k_inst=[k])[0]
    wk.check()
    split = (cam[:, None] * torch.stack([(inst == j).float() for j in range(k)])[None]).reshape(2 * k, h, w)
    ref = wk([edge], [split], beta=10, exp_times=8)[0]
    wk.check()
    assert torch.equal(rw, ref)
    wk.close()


@pytest.mark.parametrize("accel", [1, 0])
def test_walk_128_vs_reference_golden_both_schedules(golden, accel):
    """The reference's own 128x128 outputs (radius 10 and 5, 2^8 dense squarings): <= 1e-4, identical grid argmax."""
    g = golden("walk128")
    for r in (10, 5):
        key = "r%d_b10_e8_128" % r
        wk = _walker(r, accel=accel)
        cam = torch.from_numpy(g[key + "_cam"]).to(_dev())
        edge = torch.from_numpy(g[key + "_edge"])[None].to(_dev())
        rw = wk([edge], [cam], beta=10, exp_times=8)[0]
        wk.check()
        rw = rw.cpu().numpy()
        ref = g[key + "_rw"]
        err = np.abs(rw - ref).max()
        print("radius %d accel %d: max |gpu - reference| = %.2e" % (r, accel, err))
        assert err <= TOL_REF
        assert np.array_equal(np.argmax(rw[:, 0], 0), np.argmax(ref[:, 0], 0))
        wk.close()


def test_other_exponents_and_generic_radius():
    """exp_times 5..10 at radius 5/10, and a radius without a blocked kernel (generic sweep) at exp_times 8."""
    from irn_amd import synth
    h, w, c = 64, 80, 2
    cam_np, edge_np = synth.cam_blobs(c, h, w, seed=3), synth.edge_field(h, w, seed=3)
    cam, edge = torch.from_numpy(cam_np).to(_dev()), torch.from_numpy(edge_np).to(_dev())
    for r, exps in ((10, (5, 6, 7, 9, 10)), (5, (6, 9)), (7, (8,))):
        wk = _walker(r, variant=2 if r in (5, 10) else 0)
        for e in exps:
            rw = wk([edge], [cam], beta=10, exp_times=e)[0]
            wk.check()
            st = O.propagate_to_edge_stencil(cam_np, edge_np, r, 10, e)
            assert np.abs(rw.cpu().numpy() - st).max() <= TOL_F64, (r, e)
        wk.close()


_STRESS_EXACT = {}


def _stress_exact(r):
    """fp64 oracle of every adversarial case at radius r (computed once per session: ~1 s each at radius 10)."""
    if r not in _STRESS_EXACT:
        out = []
        for ci, (name, beta, make) in enumerate(_stress.cases()):
            h, w, c = (40, 52, 2) if ci % 2 else (56, 72, 3)
            edge, cam = make(h, w, c, 500 + ci)
            out.append((name, beta, edge, cam, O.propagate_to_edge_stencil(cam, edge, r, beta, 8)))
        _STRESS_EXACT[r] = out
    return _STRESS_EXACT[r]


@pytest.mark.parametrize("r", [10, 5])
@pytest.mark.parametrize("variant", [2, 1, 0])
def test_schedule_on_adversarial_fields(r, variant):
    """Where the series' maths is stressed (tests/_stress.py: edge = 0, edge = 0.999, Bernoulli 0/1 edges, a one-pixel wall,
    beta 1 / 8 / 10 / 20, white-noise CAMs), every kernel variant, default truncation bound (1e-7), the looser option (1e-6)
    and the plain iteration: <= 1e-5 from the fp64 oracle, <= 1e-4 after division by the maximum (the epilogue's scale, the
    north star's bar), argmax equal except at ties.  The measured triples are printed: on flat-spectrum operators the
    series is several times FURTHER from the exact product than the plain iteration (3e-7 against 5e-8 with no edges at
    all), on the structured ones closer; both are two orders inside the bar."""
    cases = _stress_exact(r)
    rows = []
    for opts in ({}, {"accel_tol_exp": 6}, {"accel": 0}):
        wk = _walker(r, variant, **opts)
        for beta in sorted({b for _, b, _, _, _ in cases}):
            sel = [c for c in cases if c[1] == beta]
            outs = wk([torch.from_numpy(c[2]).to(_dev()) for c in sel], [torch.from_numpy(c[3]).to(_dev()) for c in sel],
                      beta=beta, exp_times=8)
            wk.check()
            for (name, _, edge, cam, exact), o in zip(sel, outs):
                got = o.cpu().numpy()
                err = float(np.abs(got - exact).max())
                scale = max(float(exact.max()), 1e-30)
                assert err <= TOL_F64, (name, opts, err)
                assert err / scale <= TOL_REF, (name, opts, err / scale)
                n_tie = _stress.argmax_mismatch_is_tie(got[:, 0], exact[:, 0], 2e-5)
                rows.append((name, tuple(opts.items()), err, err / scale, n_tie))
        wk.close()
    for name in [c[0] for c in cases]:
        e = {k: (a, b, t) for n, k, a, b, t in rows if n == name}
        print("variant %d radius %2d %-20s series(1e-7) %.2e  series(1e-6) %.2e  plain %.2e   normalised %.2e / %.2e / %.2e   argmax ties %d/%d/%d" % (
            variant, r, name, e[()][0], e[(("accel_tol_exp", 6),)][0], e[(("accel", 0),)][0],
            e[()][1], e[(("accel_tol_exp", 6),)][1], e[(("accel", 0),)][1], e[()][2], e[(("accel_tol_exp", 6),)][2], e[(("accel", 0),)][2]))


def test_walk_accel_switch_of_the_wrapper(monkeypatch):
    """IRN_WALK_ACCEL / IRN_WALK_ACCEL_TOL_EXP (what run_sample.py --walk_accel / --walk_accel_tol_exp set for the steps)
    reach the context; explicit options still win."""
    from irn_amd.misc import indexing
    monkeypatch.setenv("IRN_WALK_ACCEL", "0")
    wk = indexing.RandomWalk(5, _dev())
    assert wk.steps(256) == 256
    wk.set_option("accel", 1)
    assert wk.steps(256) == 84
    wk.close()
    monkeypatch.setenv("IRN_WALK_ACCEL", "1")
    monkeypatch.setenv("IRN_WALK_ACCEL_TOL_EXP", "6")
    wk = indexing.RandomWalk(5, _dev())
    assert wk.steps(256) == 78
    wk.close()
