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
