"""The C restatement (oracle/walk_oracle.c, the cpu_baseline 'port') against the reference's own
outputs and the numpy oracle."""
import numpy as np

from oracle import build_oracle
from oracle import irn_oracle as O


def test_c_oracle_matches_reference_golden(golden):
    lib = build_oracle.load()
    wk = golden("walk")
    names = sorted(k[:-3] for k in wk.files if k.endswith("_rw"))
    for n in names:
        h, w, c, r, b, e = (int(v) for v in wk[n + "_params"])
        rw = build_oracle.walk(lib, wk[n + "_cam"], wk[n + "_edge"], r, b, 2 ** e)
        assert rw.shape == wk[n + "_rw"].shape
        assert np.abs(rw - wk[n + "_rw"]).max() <= 1e-4, n
        st = O.propagate_to_edge_stencil(wk[n + "_cam"], wk[n + "_edge"], r, b, e)
        assert np.abs(rw - st).max() <= 1e-6, n


def test_c_batch_port_matches_reference_golden_and_the_per_pixel_form(golden):
    """irn_oracle_walk_batch (image-parallel, row-vectorised: the timed CPU baseline of bench.py) on ALL golden cases
    in one call: <= 1e-4 from the reference's outputs, <= 1e-6 from the per-pixel fp64 form."""
    lib = build_oracle.load()
    wk = golden("walk")
    names = sorted(k[:-3] for k in wk.files if k.endswith("_rw"))
    by_cfg = {}
    for n in names:
        h, w, c, r, b, e = (int(v) for v in wk[n + "_params"])
        by_cfg.setdefault((r, b, e), []).append(n)
    assert len(by_cfg) >= 4
    for (r, b, e), group in by_cfg.items():
        outs = build_oracle.walk_batch(lib, [wk[n + "_cam"] for n in group], [wk[n + "_edge"] for n in group], r, b, 2 ** e)
        for n, rw in zip(group, outs):
            assert rw.shape == wk[n + "_rw"].shape
            assert np.abs(rw - wk[n + "_rw"]).max() <= 1e-4, n
            one = build_oracle.walk(lib, wk[n + "_cam"], wk[n + "_edge"], r, b, 2 ** e)
            assert np.abs(rw - one).max() <= 1e-6, n


def test_c_oracle_matches_reference_at_the_headline_grid(golden):
    """128x128, radius 10 and 5, 2^8 sweeps: the reference's own dense run (tests/golden/walk128.npz)."""
    lib = build_oracle.load()
    wk = golden("walk128")
    for n in sorted(k[:-3] for k in wk.files if k.endswith("_rw")):
        h, w, c, r, b, e = (int(v) for v in wk[n + "_params"])
        rw = build_oracle.walk_batch(lib, [wk[n + "_cam"]], [wk[n + "_edge"]], r, b, 2 ** e)[0]
        ref = wk[n + "_rw"]
        assert np.abs(rw - ref).max() <= 1e-4, (n, np.abs(rw - ref).max())
        assert np.array_equal(np.argmax(rw[:, 0], 0), np.argmax(ref[:, 0], 0)), n


def test_c_oracle_matches_reference_at_voc_grids(golden):
    """Ragged grids of real VOC images (94x125 at the reference's call-site radius 5, 84x125 at radius 10), 2^8 sweeps: the
    reference's own dense run (tests/golden/walk_voc.npz, round 3), and the label map through the reference's epilogue."""
    from oracle import irn_oracle as O
    lib = build_oracle.load()
    wk = golden("walk_voc")
    names = sorted(k[:-3] for k in wk.files if k.endswith("_rw"))
    assert len(names) == 2
    for n in names:
        h, w, c, r, b, e = (int(v) for v in wk[n + "_params"])
        rw = build_oracle.walk_batch(lib, [wk[n + "_cam"]], [wk[n + "_edge"]], r, b, 2 ** e)[0]
        ref = wk[n + "_rw"]
        assert np.abs(rw - ref).max() <= 1e-4, (n, np.abs(rw - ref).max())
        assert np.array_equal(np.argmax(rw[:, 0], 0), np.argmax(ref[:, 0], 0)), n
        keys = np.arange(c) * 3 + 1
        size = (4 * h - 1, 4 * w - 3)                 # a crop like a 375x497 photo gives
        _, lab, _ = O.sem_seg_epilogue(rw, size, keys, 0.25)
        _, want, _ = O.sem_seg_epilogue(ref, size, keys, 0.25)
        assert (lab != want).sum() <= 4, (n, int((lab != want).sum()))
