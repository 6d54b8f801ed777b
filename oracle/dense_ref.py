"""CPU ORACLE, dense form on PyTorch — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

The reference's own algorithm for the random walk, restated op for op on torch CPU tensors so that
it can be TIMED on the GPU box's host cores (the reference tree does not travel there):

    misc/indexing.py:91-109   edge_to_affinity      index_select over the path index tensors + max over the path axis
    misc/indexing.py:112-129  affinity_sparse2dense symmetric COO (i,j),(j,i) + unit diagonal -> dense
    misc/indexing.py:132-139  to_transition_matrix  pow(beta); divide by column sums; `times` squarings (dense sgemm)
    misc/indexing.py:141-165  propagate_to_edge     pad with 1.0, crop the matrix to the image, x*(1-edge) @ T

The index tables come from ``oracle.irn_oracle.PathIndexOracle`` (the numpy restatement of PathIndex,
pinned on the reference's tables in tests/golden/path_tables.npz).  Cost: 2*N^3 flops per squaring with
N = h*w — 70 TFLOP for a 128x128 grid at exp_times = 8, independent of the radius.

Pinned by tests/test_oracle_golden.py::test_dense_torch_restatement_vs_reference_golden (the
reference's own outputs in tests/golden/walk.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import irn_oracle as O


def edge_to_affinity(edge, path_indices):
    """edge [B, Hp*Wp] -> [B, |S|, Ns]  (misc/indexing.py:91-109)."""
    out = []
    for ind in path_indices:                                   # int64 [n_paths, L, Ns]
        ind = torch.as_tensor(ind)
        dist = torch.index_select(edge, -1, ind.reshape(-1)).view(edge.size(0), *ind.shape)
        out.append(1 - dist.max(dim=2).values)                 # max_pool2d over the whole path axis
    return torch.cat(out, dim=1)


def affinity_sparse2dense(aff, ind_from, ind_to, n_vertices):
    """misc/indexing.py:112-129 (COO with both orientations and a unit diagonal, densified)."""
    ind_from = torch.as_tensor(ind_from)
    ind_to = torch.as_tensor(ind_to)
    a = aff.reshape(-1)
    f = ind_from.repeat(ind_to.size(0)).view(-1)
    t = ind_to.reshape(-1)
    diag = torch.arange(n_vertices)
    idx = torch.cat([torch.stack([f, t]), torch.stack([diag, diag]), torch.stack([t, f])], dim=1)
    val = torch.cat([a, torch.ones(n_vertices), a])
    return torch.sparse_coo_tensor(idx, val, (n_vertices, n_vertices)).to_dense()


def to_transition_matrix(affinity_dense, beta, times):
    """misc/indexing.py:132-139."""
    s = torch.pow(affinity_dense, beta)
    t = s / torch.sum(s, dim=0, keepdim=True)
    for _ in range(times):
        t = torch.matmul(t, t)
    return t


def propagate_to_edge(x, edge, radius=5, beta=10, exp_times=8, timings=None):
    """misc/indexing.py:141-165.  x: float32 [..., h, w]; edge [1, h, w]; returns [C', 1, h, w].
    `timings` (a dict) receives the seconds of the phases, in the reference's own order: "path_index" (:148, the index
    tables), "affinity" (:150-151, pad + path-max gather), "dense" (:154-157, COO -> dense (hp*wp)^2 -> crop to (h*w)^2),
    "pow_normalise" (:133-135), "squarings" (:136-137, `exp_times` dense sgemm: 2*N^3 flops each), "final" (:162-164), and
    the two sums "setup" = the first three (~N^2) and "transition" = the rest (~N^3)."""
    import time
    tm = {}
    t0 = time.perf_counter()
    x = torch.as_tensor(np.asarray(x, np.float32))
    edge = torch.as_tensor(np.asarray(edge, np.float32)).reshape(1, *x.shape[-2:])
    h, w = x.shape[-2:]
    hp, wp = h + radius, w + 2 * radius
    pi = O.PathIndexOracle(radius, (hp, wp))
    t1 = time.perf_counter()
    tm["path_index"] = t1 - t0
    edge_padded = F.pad(edge, (radius, radius, 0, radius), mode="constant", value=1.0)
    aff = edge_to_affinity(edge_padded.reshape(1, -1), pi.path_indices)
    t2 = time.perf_counter()
    tm["affinity"] = t2 - t1
    dense = affinity_sparse2dense(aff, pi.src_indices, pi.dst_indices, hp * wp)
    dense = dense.view(hp, wp, hp, wp)[:-radius, radius:-radius, :-radius, radius:-radius].reshape(h * w, h * w)
    t3 = time.perf_counter()
    tm["dense"] = t3 - t2
    s = torch.pow(dense, beta)                                  # to_transition_matrix, misc/indexing.py:132-139
    del dense
    t = s / torch.sum(s, dim=0, keepdim=True)
    del s
    t4 = time.perf_counter()
    tm["pow_normalise"] = t4 - t3
    for _ in range(exp_times):
        t = torch.matmul(t, t)
    t5 = time.perf_counter()
    tm["squarings"] = t5 - t4
    xe = x.reshape(-1, h, w) * (1 - edge)
    out = torch.matmul(xe.view(-1, h * w), t).view(-1, 1, h, w)
    tm["final"] = time.perf_counter() - t5
    tm["setup"] = t3 - t0
    tm["transition"] = time.perf_counter() - t3
    if timings is not None:
        timings.update(tm)
    return out
