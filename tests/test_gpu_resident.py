"""GPU parity of the weights-stationary persistent walk (irn_walk option variant=2,
irn_amd/csrc/walk_resident.hip) — same operator as misc/indexing.py:141-165, weights held in
registers for all sweeps, tiles of an image exchanging state as tagged granules inside one launch.

Staleness is the failure mode to hunt: late sweeps barely change the state, so a tile that read a
neighbour's x_{t-2} instead of x_t would pass a loose tolerance at 2^8 sweeps.  Hence (a) few-sweep
runs, where consecutive states differ by O(0.1), against the full-fp64 generic kernel, and (b)
bit-equality of the one-launch walk with the same kernel relaunched every sweep (kernel boundaries
instead of in-launch hand-offs)."""
import numpy as np
import pytest
import torch

from oracle import irn_oracle as O

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


def test_resident_vs_reference_golden(golden):
    wk = golden("walk")
    names = sorted(k[:-3] for k in wk.files if k.endswith("_rw"))
    done = 0
    for n in names:
        h, w, c, r, b, e = (int(v) for v in wk[n + "_params"])
        if r not in (5, 10):
            continue
        walker = _walker(r)
        cam = torch.from_numpy(wk[n + "_cam"]).to(_dev())
        if n.endswith("_ck"):
            cam = cam.view(2, c // 2, h, w)
        edge = torch.from_numpy(wk[n + "_edge"])[None].to(_dev())
        rw = walker([edge], [cam], beta=b, exp_times=e)[0]
        walker.check()
        rw = rw.cpu().numpy()
        ref = wk[n + "_rw"]
        assert rw.shape == ref.shape, n
        assert np.abs(rw - ref).max() <= TOL_REF, (n, np.abs(rw - ref).max())
        assert np.array_equal(np.argmax(rw[:, 0], 0), np.argmax(ref[:, 0], 0)), n
        st = O.propagate_to_edge_stencil(wk[n + "_cam"], wk[n + "_edge"], r, b, e)
        assert np.abs(rw - st).max() <= TOL_F64, (n, np.abs(rw - st).max())
        walker.close()
        done += 1
    assert done >= 8


@pytest.mark.parametrize("r,shapes", [
    (10, [(128, 128, 1), (128, 128, 2), (128, 128, 3), (128, 128, 1), (128, 128, 4), (128, 128, 1)]),   # 2 rounds
    (10, [(94, 125, 2), (125, 84, 1), (40, 52, 9), (33, 70, 20), (128, 128, 1), (30, 30, 3), (130, 250, 2)]),
    (5, [(128, 128, 3), (94, 125, 1), (47, 31, 7), (12, 9, 2), (130, 66, 17), (64, 64, 4)]),
    (10, [(256, 256, 5)]),                                                                          # all 256 workgroups
    (5, [(256, 256, 11), (128, 128, 2)]),
])
def test_few_sweeps_match_generic_fp64_kernel(r, shapes):
    edges, cams = _inputs(shapes, 300)
    res = _walker(r)
    gen = _walker(r, variant=0)
    for n_sw in (1, 2, 3, 7):
        a = res(edges, cams, beta=10, n_sweeps=n_sw)
        res.check()
        b = gen(edges, cams, beta=10, n_sweeps=n_sw)
        for i in range(len(shapes)):
            d = (a[i] - b[i]).abs().max().item()
            assert d <= 1e-6, (r, shapes[i], n_sw, d)
    res.close()
    gen.close()


@pytest.mark.parametrize("r,shapes", [
    (10, [(128, 128, 1), (128, 128, 3), (94, 125, 2), (125, 94, 12), (128, 128, 2)]),
    (5, [(128, 128, 2), (94, 125, 9), (125, 94, 1), (60, 200, 3)]),
])
def test_one_launch_equals_launch_per_sweep_bitwise(r, shapes):
    """In-launch hand-offs between tiles vs kernel boundaries: identical arithmetic, so any stale or
    torn hand-off shows as a bit difference.  Runs twice on the same workspace (stale tags of the
    previous run must never match)."""
    edges, cams = _inputs(shapes, 500)
    per = _walker(r, sweeps_per_launch=1)
    ref = [o.clone() for o in per(edges, cams, beta=10, n_sweeps=24)]
    per.check()
    one = _walker(r)
    for rep in range(3):
        out = one(edges, cams, beta=10, n_sweeps=24)
        one.check()
        for i in range(len(shapes)):
            assert torch.equal(out[i], ref[i]), (r, shapes[i], rep)
    # chunks of 5 sweeps per launch (odd chunk: both buffer parities at launch boundaries)
    five = _walker(r, sweeps_per_launch=5)
    out = five(edges, cams, beta=10, n_sweeps=24)
    five.check()
    for i in range(len(shapes)):
        assert torch.equal(out[i], ref[i])
    for wkr in (per, one, five):
        wkr.close()


def test_uneven_load_many_rounds_batch_equals_single():
    """40 images with 1..6 channels (10 rounds at radius 10): tiles of light images run ahead of heavy
    ones inside a round and rounds pipeline; every image must equal its single-image run bit for bit."""
    shapes = [(128, 128, 1 + (i * 7) % 6) for i in range(40)]
    edges, cams = _inputs(shapes, 900)
    wk = _walker(10)
    batch = [o.clone() for o in wk(edges, cams, beta=10, n_sweeps=40)]
    wk.check()
    for i in (0, 5, 17, 39):
        single = wk([edges[i]], [cams[i]], beta=10, n_sweeps=40)[0]
        wk.check()
        assert torch.equal(single, batch[i]), i
    gen = _walker(10, variant=0)
    for i in (3, 22):
        g = gen([edges[i]], [cams[i]], beta=10, n_sweeps=40)[0]
        assert (g - batch[i]).abs().max().item() <= 2e-6
    wk.close()
    gen.close()


def test_instance_split_and_full_walk_vs_oracle(golden):
    ins = golden("instance")
    for name in "abc":
        edge, cam = ins[name + "_edge"], ins[name + "_cam"]
        shape = tuple(ins[name + "_instance_map_shape"])
        inst = np.unpackbits(ins[name + "_instance_map"])[:int(np.prod(shape))].reshape(shape)
        cmap = np.argmax(inst, 0).astype(np.int32)
        wk = _walker(5)
        rw = wk([torch.from_numpy(edge).to(_dev())], [torch.from_numpy(cam).to(_dev())], beta=10, exp_times=8,
                inst_maps=[torch.from_numpy(cmap).to(_dev())], k_inst=[shape[0]])[0]
        wk.check()
        assert np.abs(rw.cpu().numpy() - ins[name + "_rw"]).max() <= TOL_REF
        wk.close()


def test_full_size_256_sweeps_vs_fp64_oracle_and_streaming_kernel():
    from irn_amd import synth
    h = w = 128
    edge = synth.edge_field(h, w, seed=11)
    cam = synth.cam_blobs(2, h, w, seed=11)
    e, c = torch.from_numpy(edge).to(_dev()), torch.from_numpy(cam).to(_dev())
    wk = _walker(10)
    rw = wk([e], [c], beta=10, exp_times=8)[0]
    wk.check()
    st = O.propagate_to_edge_stencil(cam, edge, 10, 10, 8)
    assert np.abs(rw.cpu().numpy() - st).max() <= TOL_F64
    assert np.array_equal(np.argmax(rw.cpu().numpy()[:, 0], 0), np.argmax(st[:, 0], 0))
    blocked = _walker(10, variant=1)
    rb = blocked([e], [c], beta=10, exp_times=8)[0]
    assert (rw - rb).abs().max().item() <= 2e-6
    wk.close()
    blocked.close()


def test_falls_back_when_an_image_does_not_fit_a_round():
    """A 264x264 radius-10 image has 33x9 = 297 tiles > 256 workgroups: the run must still be right
    (streaming sweeps take over) — and so must a narrow image (w < radius)."""
    for r, shp in ((10, (264, 264, 2)), (5, (20, 4, 2))):
        edges, cams = _inputs([shp], 40)
        wk = _walker(r)
        a = wk(edges, cams, beta=10, n_sweeps=8)[0]
        wk.check()
        gen = _walker(r, variant=0)
        b = gen(edges, cams, beta=10, n_sweeps=8)[0]
        assert (a - b).abs().max().item() <= 2e-6
        wk.close()
        gen.close()


def test_radius5_plain_store_vote_is_bitwise_equal_and_radius10_refuses():
    """Option plain_store (radius 5): tiles vote their XCC id per image and store the state without sc1 only when
    the whole image was seen on one XCD; the arithmetic is the same, so results equal the default bit for bit —
    36 images of 1..5 channels (3 rounds), launched whole and in 5-sweep chunks (votes are cleared per launch)."""
    from irn_amd._lib import IrnHipError
    shapes = [(128, 128, 1 + (i * 3) % 5) for i in range(34)] + [(94, 125, 2), (60, 200, 1)]
    edges, cams = _inputs(shapes, 1300)
    ref_w = _walker(5)
    ref = [o.clone() for o in ref_w(edges, cams, beta=10, n_sweeps=40)]
    ref_w.check()
    for extra in ({}, {"sweeps_per_launch": 5}, {"poll_delay_plain": 0}):
        wk = _walker(5, plain_store=1, **extra)
        for rep in range(2):
            out = wk(edges, cams, beta=10, n_sweeps=40)
            wk.check()
            for i in range(len(shapes)):
                assert torch.equal(out[i], ref[i]), (extra, rep, i)
        wk.close()
    ref_w.close()
    with pytest.raises(IrnHipError, match="radius 5"):
        _walker(10, plain_store=1)


def test_timeout_is_repaired_by_sync_and_reported_by_check():
    """A persistent launch that gives up its bounded wait (test hook `inject_timeout`: the first poll that misses twice
    gives up) must never hand garbage to the caller: `sync()` re-runs the batch on the streaming sweeps and says so;
    `check()` — the benchmark's call — raises instead; the next run is healthy again."""
    from irn_amd._lib import IrnHipError
    shapes = [(128, 128, 1), (128, 128, 2), (94, 125, 3)]
    edges, cams = _inputs(shapes, 2100)
    wk = _walker(10)
    ref = [o.clone() for o in wk(edges, cams, beta=10, n_sweeps=48)]
    assert wk.sync() is False and wk.fallback_runs == 0
    wk.set_option("poll_delay", 0)                 # early polls: plenty of misses
    wk.set_option("inject_timeout", 1)
    out = wk(edges, cams, beta=10, n_sweeps=48)
    assert wk.sync() is True and wk.fallback_runs == 1
    for i in range(len(shapes)):
        assert (out[i] - ref[i]).abs().max().item() <= 2e-6, i      # streaming kernel: same operator, fp32 rounding apart
    wk.check()                                     # handled: nothing left to report
    wk.set_option("inject_timeout", 1)
    wk(edges, cams, beta=10, n_sweeps=48)
    with pytest.raises(IrnHipError, match="timed out"):
        wk.check()
    wk.set_option("poll_delay", 10)
    out = wk(edges, cams, beta=10, n_sweeps=48)
    assert wk.sync() is False
    for i in range(len(shapes)):
        assert torch.equal(out[i], ref[i]), i
    wk.close()


def test_propagate_to_edge_never_returns_a_timed_out_result():
    """The drop-in entry point waits and repairs (the reference's call is synchronous too)."""
    from irn_amd.misc import indexing
    edges, cams = _inputs([(128, 128, 2)], 2200)
    ref = indexing.propagate_to_edge(cams[0], edges[0][None], radius=10, beta=10, exp_times=5).clone()
    wk = indexing._walker(_dev(), 10)
    wk.set_option("poll_delay", 0)
    wk.set_option("inject_timeout", 1)
    n0 = wk.fallback_runs
    out = indexing.propagate_to_edge(cams[0], edges[0][None], radius=10, beta=10, exp_times=5)
    assert wk.fallback_runs == n0 + 1
    assert (out - ref).abs().max().item() <= 2e-6
    wk.set_option("poll_delay", 10)


@pytest.mark.parametrize("cooperative", [1, 0])
def test_walk_while_another_stream_holds_the_compute_units(cooperative):
    """Co-residency of the persistent grid is requested from the runtime (cooperative launch), not assumed: with a
    GEMM chain running on a second stream the walk's workgroups become resident as compute units drain, the tiles wait
    for each other a little longer, and the result is the same — bit for bit unless the bounded wait expired, in which
    case `sync()` has re-run the batch on the streaming sweeps."""
    shapes = [(128, 128, 1 + i % 3) for i in range(8)]
    edges, cams = _inputs(shapes, 2300)
    wk = _walker(10, cooperative=cooperative)
    ref = [o.clone() for o in wk(edges, cams, beta=10, n_sweeps=64)]
    assert wk.sync() is False
    a = torch.randn(8192, 8192, device=_dev())
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(30):
            a = (a @ a) * 1e-4
    out = wk(edges, cams, beta=10, n_sweeps=64)
    fell = wk.sync()
    torch.cuda.synchronize()
    for i in range(len(shapes)):
        if fell:
            assert (out[i] - ref[i]).abs().max().item() <= 2e-6, i
        else:
            assert torch.equal(out[i], ref[i]), i
    wk.close()


def test_start_up_self_checks_never_change_results():
    """Round 4: the walk's two measured assumptions are checked, not trusted (irn_walk_tuning).  The block -> XCD placement
    has been probed once a batch is configured; the poll delay of single-channel jobs is probed on the first
    representative batch of the process (8 / 10 / 12 / 14, cached per device) unless it is pinned; and whatever the delay,
    the walk writes the same bits."""
    shapes = [(128, 128, 1)] * 12 + [(128, 128, 2)] * 4            # 16 images x 64 tiles = 4 rounds, 3/4 single-channel
    edges, cams = _inputs(shapes, 4100)
    auto = _walker(10)
    ref = [o.clone() for o in auto(edges, cams, beta=10, exp_times=8)]
    auto.check()
    t = auto.tuning()
    print("walk self-checks: %s" % (t,))
    assert t["placement"] in (1, 2) and t["poll_delay"] in (8, 10, 12, 14)
    if t["probe_ms"] is not None:                                  # this context was the one that probed
        assert len(t["probe_ms"]) == 4 and all(v > 0 for v in t["probe_ms"])
    for delay in (6, 16):
        pinned = _walker(10, poll_delay=delay)
        out = pinned(edges, cams, beta=10, exp_times=8)
        pinned.check()
        assert pinned.tuning()["poll_delay"] == delay and pinned.tuning()["probe_ms"] is None
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        pinned.close()
    small = _walker(10)                                            # a batch too small to be representative still runs
    one = small([edges[0]], [cams[0]], beta=10, exp_times=8)[0]
    small.check()
    assert torch.equal(one, ref[0]) and small.tuning()["poll_delay"] == t["poll_delay"]
    small.close()
    auto.close()
