"""How the weights-stationary walk packs a batch onto the device (irn_walk_plan_rounds, csrc/walk_resident.hip
pack_rounds): host arithmetic, so it is checked here without a GPU.  The reference has no counterpart — it walks one image
at a time (step/make_sem_seg_labels.py:41, misc/indexing.py:141-165) — so what is pinned is the contract the kernel relies
on (every tile of every image exactly once, the tiles of an image in ONE round on consecutive slots) and the balance the
placement is there for."""
import ctypes as C

import numpy as np
import pytest

from irn_amd import _lib, synth

PI = C.POINTER(C.c_int32)
N_WG = 256


def plan(radius, h, w, c, n_wg=N_WG, placement=1):
    h, w, c = (np.ascontiguousarray(a, dtype=np.int32) for a in (h, w, c))
    nr = C.c_int(-1)
    _lib.check(_lib.lib.irn_walk_plan_rounds(radius, len(h), h.ctypes.data_as(PI), w.ctypes.data_as(PI), c.ctypes.data_as(PI),
                                             n_wg, placement, None, 0, C.byref(nr)))
    jobs = np.full((max(nr.value, 1), n_wg, 4), -7, np.int32)
    if nr.value:
        _lib.check(_lib.lib.irn_walk_plan_rounds(radius, len(h), h.ctypes.data_as(PI), w.ctypes.data_as(PI), c.ctypes.data_as(PI),
                                                 n_wg, placement, jobs.ctypes.data_as(PI), nr.value, C.byref(nr)))
    return nr.value, jobs[:nr.value]


def slot_of(b, n_wg, placement):
    """inverse of the slot -> block mapping: consecutive slots share an XCD when block b runs on XCD b % 8"""
    per = n_wg // 8
    return b if placement == 2 or n_wg % 8 else (b % 8) * per + b // 8


def step_cost(radius, c):
    per = (1.5 if c == 1 else 1.07 if c == 2 else 1.0) if radius == 10 else (1.15 if c == 1 else 1.0)
    return 7.0 + c * per * 84.0


def check_plan(radius, h, w, c, jobs, n_wg, placement):
    """the contract of the kernel; returns (tile height, tile width, modelled finishing time per workgroup)"""
    n = len(h)
    seen = {}
    for r in range(jobs.shape[0]):
        for b in range(n_wg):
            i, ty, tx, nt = (int(v) for v in jobs[r, b])
            if i < 0:
                continue
            assert 0 <= i < n and 0 <= ty < h[i] and 0 <= tx < w[i]
            seen.setdefault(i, []).append((r, slot_of(b, n_wg, placement), ty, tx, nt))
    assert sorted(seen) == list(range(n)), "every image is placed"
    th = min([t[2] for v in seen.values() for t in v if t[2] > 0], default=None)
    tw = min([t[3] for v in seen.values() for t in v if t[3] > 0], default=None)
    for i, v in seen.items():
        assert len({t[0] for t in v}) == 1, "the tiles of an image share a round (they wait for each other)"
        slots = sorted(t[1] for t in v)
        assert slots == list(range(slots[0], slots[0] + len(v))), "on consecutive slots"
        assert all(t[4] == len(v) for t in v)
        origins = sorted((t[2], t[3]) for t in v)
        assert len(set(origins)) == len(origins)
        if th and tw:
            want = sorted((y, x) for y in range(0, h[i], th) for x in range(0, w[i], tw))
            assert origins == want, "the tiles cover the grid once"
    busy = np.zeros(n_wg)
    for r in range(jobs.shape[0]):
        for i in sorted({int(v) for v in jobs[r, :, 0] if v >= 0}):
            mine = np.nonzero(jobs[r, :, 0] == i)[0]
            busy[mine] = busy[mine].max() + step_cost(radius, int(c[i]))
    return th, tw, busy


@pytest.mark.parametrize("radius,placement", [(10, 1), (10, 2), (5, 1)])
def test_bench_batch_is_packed_validly_and_balanced(radius, placement):
    n = 192 if radius == 10 else 256
    c = np.array([synth.voc_num_classes(1000 + i % 96) for i in range(n)], np.int32)
    h = w = np.full(n, 128, np.int32)
    n_rounds, jobs = plan(radius, h, w, c, placement=placement)
    th, tw, busy = check_plan(radius, h, w, c, jobs, N_WG, placement)
    per_image = (128 // th) * (128 // tw)
    assert n_rounds == -(-n * per_image // N_WG)
    ideal = sum(step_cost(radius, int(k)) for k in c) * per_image / N_WG
    print("radius %d: %d rounds, tile %dx%d, modelled finish %.0f .. %.0f (ideal %.0f)" % (radius, n_rounds, th, tw, busy.min(), busy.max(), ideal))
    # what it replaced (rounds 1-3 and sessions 1-15 of round 4): descending order, first range first
    ranges = N_WG // per_image
    naive = np.zeros(ranges)
    for j, k in enumerate(sorted((int(k) for k in c), reverse=True)):
        naive[j % ranges] += step_cost(radius, k)
    print("          first range first: %.0f .. %.0f" % (naive.min(), naive.max()))
    assert busy.max() < naive.max()
    # radius 10: 48 jobs per range, 0.1 % over the ideal (was 1.0 %); radius 5: 16 jobs per range, one job is 4.6 % of a range
    assert busy.max() <= (1.003 if radius == 10 else 1.015) * ideal


def test_ragged_voc_batch_and_the_batches_that_do_not_fit():
    rng = np.random.default_rng(5)
    sizes = [(94, 125), (125, 84), (128, 128), (84, 125), (125, 94), (71, 125), (32, 40)]
    pick = rng.integers(0, len(sizes), 60)
    h = np.array([sizes[k][0] for k in pick], np.int32)
    w = np.array([sizes[k][1] for k in pick], np.int32)
    c = rng.integers(1, 5, 60).astype(np.int32)
    for radius in (5, 10):
        n_rounds, jobs = plan(radius, h, w, c)
        assert n_rounds > 0
        check_plan(radius, h, w, c, jobs, N_WG, 1)
    # one image: one round, everything else idle
    n_rounds, jobs = plan(10, [128], [128], [3])
    assert n_rounds == 1 and (jobs[0, :, 0] >= 0).sum() == (jobs[0, :, 0] == 0).sum() > 0
    # narrower than the radius / more tiles than workgroups: not an error, the batch runs on the streaming sweeps
    assert plan(10, [64, 64], [64, 8], [1, 1])[0] == 0
    assert plan(10, [1024], [1024], [1])[0] == 0
    assert plan(10, [128], [128], [1], n_wg=16)[0] == 0
    # a device whose workgroup count is no multiple of 8 keeps launch order
    n_rounds, jobs = plan(10, [128, 128], [128, 128], [1, 2], n_wg=100, placement=1)
    assert n_rounds == 2 or n_rounds == 1
    check_plan(10, [128, 128], [128, 128], [1, 2], jobs, 100, 1)


def test_bad_arguments_are_refused():
    nr = C.c_int(0)
    one = np.ones(1, np.int32)
    p = one.ctypes.data_as(PI)
    assert _lib.lib.irn_walk_plan_rounds(7, 1, p, p, p, N_WG, 1, None, 0, C.byref(nr)) != 0      # no persistent walk at radius 7
    assert _lib.lib.irn_walk_plan_rounds(10, 0, p, p, p, N_WG, 1, None, 0, C.byref(nr)) != 0
    assert _lib.lib.irn_walk_plan_rounds(10, 1, None, p, p, N_WG, 1, None, 0, C.byref(nr)) != 0
    out = np.zeros((1, N_WG, 4), np.int32)
    c4 = np.full(8, 1, np.int32)
    h8 = np.full(8, 128, np.int32)
    # eight 64-tile images need two rounds: a one-round buffer is refused, the count is still reported
    rc = _lib.lib.irn_walk_plan_rounds(10, 8, h8.ctypes.data_as(PI), h8.ctypes.data_as(PI), c4.ctypes.data_as(PI), N_WG, 1,
                                       out.ctypes.data_as(PI), 1, C.byref(nr))
    assert rc != 0 and nr.value == 2 and b"too small" in _lib.lib.irn_last_error()
