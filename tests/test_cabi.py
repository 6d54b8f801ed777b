"""The C-ABI library loads without a GPU, exports every symbol include/irn_hip.h declares, and its
host-only entry points (path tables, argument validation) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from irn_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "irn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(irn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(_lib.lib, n), "header declares %s but the library does not export it" % n
    assert sorted(_lib.EXPORTS) == names, "irn_amd/_lib.py binds a different set than the header declares"


def test_version_and_error_string():
    assert _lib.lib.irn_version() >= 100
    nd, nc = C.c_int(), C.c_int()
    assert _lib.lib.irn_path_count(1, C.byref(nd), C.byref(nc)) == 1          # IRN_ERR_ARG
    assert b"radius" in _lib.lib.irn_last_error()
    with pytest.raises(_lib.IrnHipError):
        _lib.check(_lib.lib.irn_path_count(99, C.byref(nd), C.byref(nc)))


@pytest.mark.parametrize("r", [2, 3, 5, 7, 10])
def test_path_table_reference_order_vs_golden(golden, r):
    from irn_amd.misc import indexing
    pt = golden("path_tables")
    dst, start, cells = indexing._path_table(r, 0)
    assert np.array_equal(dst, pt["r%d_dst" % r])
    assert np.array_equal(cells, pt["r%d_paths_flat" % r])
    lens = np.diff(start)
    assert np.array_equal(lens, np.repeat(pt["r%d_group_lens" % r], pt["r%d_group_counts" % r]))


@pytest.mark.parametrize("r", [5, 10])
def test_path_table_raster_order_is_a_permutation(r):
    from irn_amd.misc import indexing
    d0, s0, c0 = indexing._path_table(r, 0)
    d1, s1, c1 = indexing._path_table(r, 1)
    assert [tuple(x) for x in d1] == sorted(tuple(x) for x in d0)
    paths0 = {tuple(d0[i]): c0[s0[i]:s0[i + 1]].tolist() for i in range(len(d0))}
    for i in range(len(d1)):
        assert c1[s1[i]:s1[i + 1]].tolist() == paths0[tuple(d1[i])]


def test_pathindex_mirror_matches_golden(golden):
    from irn_amd.misc import indexing
    pt = golden("path_tables")
    for r in (3, 5, 10):
        pi = indexing.PathIndex(r, tuple(pt["r%d_size" % r]))
        assert pi.radius_floor == r - 1
        assert np.array_equal(pi.search_dst, pt["r%d_dst" % r])
        assert np.array_equal(pi.src_indices, pt["r%d_src_indices" % r])
        assert np.array_equal(pi.dst_indices, pt["r%d_dst_indices" % r])
        assert np.array_equal(np.concatenate([p.reshape(-1) for p in pi.path_indices]),
                              pt["r%d_path_indices_flat" % r])
        assert pi.path_indices.radius == r


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = _lib.lib.irn_walk_create(5, C.byref(ctx))
    assert rc == 2 and _lib.lib.irn_last_error()                              # IRN_ERR_HIP, no crash
    from irn_amd.misc import indexing
    with pytest.raises(ValueError):
        indexing.propagate_to_edge(torch.zeros(1, 4, 4), torch.zeros(1, 4, 4))  # CPU tensors are refused


def test_argument_validation_of_the_newer_entry_points():
    """Null pointers, non-positive sizes and unsupported channel counts are refused with IRN_ERR_ARG before any
    device work (no GPU needed); the message names the entry point."""
    L = _lib.lib
    assert L.irn_msf_pack(None, 4, 4, 1, None, None, None, None, None, None) == 1
    assert b"irn_msf_pack" in L.irn_last_error()
    assert L.irn_bicubic_resize_u8(None, 4, 4, 3, 2, 2, None, None, None) == 1
    one = C.c_void_p(64)                                                       # never dereferenced on these paths
    assert L.irn_bicubic_resize_u8(one, 4, 4, 2, 2, 2, one, one, None) == 1   # 2 channels
    assert b"channels" in L.irn_last_error()
    assert L.irn_bicubic_resize_u8(one, 0, 4, 3, 2, 2, one, one, None) == 1   # empty image
    assert L.irn_bicubic_resize_u8(one, 4, 4, 3, 2, 3, one, None, None) == 1  # width changes: scratch required
    assert L.irn_bicubic_scratch_bytes(375, 500, 188, 250, 3) == 375 * 250 * 3
    assert L.irn_bicubic_scratch_bytes(375, 500, 188, 500, 3) == 0            # vertical pass only
    ks = C.c_int32()
    assert L.irn_bicubic_plan(0, 4, C.byref(ks), None, None, None, 0) == 1
    assert L.irn_bicubic_plan(8, 4, C.byref(ks), None, None, None, 0) == 0 and ks.value == 9   # support 4 -> 2*4+1
    lo = (C.c_int32 * 4)()
    assert L.irn_bicubic_plan(8, 4, C.byref(ks), lo, lo, lo, 3) == 1          # weights array too small
    assert L.irn_bn_act(None, None, one, one, None, None, 1, 4, 16, 1, None) == 1 and b"irn_bn_act" in L.irn_last_error()
    assert L.irn_bn_act(one, None, one, one, None, None, 1, 0, 16, 1, None) == 1               # no channels
    assert L.irn_bn_act(C.c_void_p(68), None, one, one, None, None, 1, 4, 16, 1, None) == 1    # not 16-byte aligned
    assert L.irn_bn_act(one, None, one, one, None, None, 1 << 20, 2048, 4096, 1, None) == 1 and b"2^31" in L.irn_last_error()
    assert L.irn_bn_act(one, None, one, one, None, None, 0, 4, 16, 1, None) == 0   # empty batch: nothing to do
    assert L.irn_bn_act(one, None, one, one, one, one, 1, 4, 16, 1, None) == 1      # residual constants without a residual
    assert L.irn_bn_act(one, one, one, one, one, None, 1, 4, 16, 1, None) == 1      # only one of the two
    assert L.irn_stem_pool(None, one, one, 1, 64, 8, 8, one, None) == 1 and b"irn_stem_pool" in L.irn_last_error()
    assert L.irn_stem_pool(one, one, one, 1, 64, 0, 8, one, None) == 1
    assert L.irn_stem_pool(one, one, one, 0, 64, 8, 8, one, None) == 0             # empty batch
    assert L.irn_upsample_bilinear(None, 1, 4, 4, 2, 1, one, None) == 1
    assert L.irn_upsample_bilinear(one, 1, 4, 4, 0, 1, one, None) == 1 and b"factor" in L.irn_last_error()
    assert L.irn_upsample_bilinear(one, 1, 4, 4, 2, 1, C.c_void_p(72), None) == 1  # output not 16-byte aligned
    assert L.irn_upsample_bilinear(one, 0, 4, 4, 2, 1, one, None) == 0
    assert L.irn_pair_displacement(None, 1, 2, 20, 27, 5, None, None) == 1
    assert L.irn_pair_displacement(one, 1, 2, 4, 27, 5, one, None) == 1       # grid smaller than the radius
    assert b"too small" in L.irn_last_error()
    assert L.irn_pair_displacement_backward(one, 40000, 2, 20, 27, 5, one, None) == 1    # batch * channels > 65535
