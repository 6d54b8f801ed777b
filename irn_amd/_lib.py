"""ctypes binding of libirn_hip.so (C ABI declared in include/irn_hip.h).

There is no CPU fallback: if the library is missing, importing this module raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C irn_amd/csrc``.
"""
import ctypes as C
import os

# PyTorch-ROCm owns the process's HIP runtime (it ships its own libamdhip64.so.7).  It must be
# loaded BEFORE libirn_hip.so so that the library binds to the same runtime instance: loading ours
# first pulls in /opt/rocm's copy, the process ends up with two runtimes, and ours then reports
# "no ROCm-capable device" while torch works (seen on the GPU box under pytest's import order).
import torch  # noqa: F401  (side effect: HIP runtime of the process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRN_HIP_LIB") or os.path.join(_HERE, "lib", "libirn_hip.so")   # override: A/B runs of two builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "irn_amd: %s not found — the HIP extension is the product path and has no fallback; "
        "build it with `make -C irn_amd/csrc` (needs hipcc, targets gfx950)" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

vp, i32, f32, sz, i64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
pi32 = C.POINTER(C.c_int32)
ppv = C.POINTER(C.c_void_p)

_SIGS = {
    "irn_version": (C.c_int, []),
    "irn_last_error": (C.c_char_p, []),
    "irn_path_count": (i32, [i32, C.POINTER(i32), C.POINTER(i32)]),
    "irn_path_table": (i32, [i32, i32, pi32, pi32, pi32]),
    "irn_edge_to_affinity": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "irn_edge_to_affinity_backward": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "irn_pair_displacement": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "irn_pair_displacement_backward": (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "irn_walk_create": (i32, [i32, C.POINTER(vp)]),
    "irn_walk_destroy": (i32, [vp]),
    "irn_walk_configure": (i32, [vp, i32, pi32, pi32, pi32, C.POINTER(sz)]),
    "irn_walk_run": (i32, [vp, ppv, ppv, ppv, pi32, ppv, f32, i32, vp, sz, vp]),
    "irn_walk_set_option": (i32, [vp, C.c_char_p, i32]),
    "irn_walk_steps": (i32, [vp, i32, C.POINTER(i32)]),
    "irn_power_series": (i32, [i32, i32, C.POINTER(C.c_double), i32, C.POINTER(i32), C.POINTER(i32)]),
    "irn_walk_sync": (i32, [vp, C.POINTER(i32)]),
    "irn_walk_check": (i32, [vp]),
    "irn_walk_fallback_runs": (i32, [vp]),
    "irn_walk_tuning": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32)]),
    "irn_walk_plan_rounds": (i32, [i32, i32, pi32, pi32, pi32, i32, i32, pi32, i32, C.POINTER(i32)]),
    "irn_walk_read_profile": (i32, [vp, vp]),
    "irn_walk_enable_timing": (i32, [vp, i32]),
    "irn_walk_last_sweep_ms": (i32, [vp, C.POINTER(f32), C.POINTER(i32)]),
    "irn_walk_export_weights": (i32, [vp, i32, vp, vp, vp, vp]),
    "irn_label_epilogue": (i32, [i32, ppv, pi32, pi32, pi32, pi32, pi32, f32, ppv, ppv, ppv, ppv, vp, vp]),
    "irn_cam_merge": (i32, [i32, ppv, pi32, pi32, i32, vp, i32, i32, i32, vp, vp, vp, vp]),
    "irn_bicubic_plan": (i32, [i32, i32, pi32, pi32, pi32, pi32, sz]),
    "irn_bicubic_scratch_bytes": (sz, [i32, i32, i32, i32, i32]),
    "irn_bicubic_resize_u8": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "irn_msf_pack": (i32, [vp, i32, i32, i32, pi32, pi32, vp, ppv, vp, vp]),
    "irn_bn_act": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, vp]),
    "irn_bn_act_nhwc": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "irn_conv1x1_workspace_bytes": (sz, []),
    "irn_conv1x1_algo_count": (i32, [i64, i32, i32, i32, i32, i32, sz, C.POINTER(i32)]),
    "irn_conv1x1_nhwc": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, sz, vp]),
    "irn_split16": (i32, [vp, vp, vp, i32, vp, i64, i32, vp, vp]),
    "irn_split16_pad": (i32, [vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    "irn_conv3x3_split_gemm": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, i32, vp, sz, vp]),
    "irn_gemm16_algo_count": (i32, [i64, i32, i32, i32, i32, i32, sz, C.POINTER(i32)]),
    "irn_gemm16_nhwc": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp, sz, vp]),
    "irn_stem_pool": (i32, [vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    "irn_upsample_bilinear": (i32, [vp, i64, i32, i32, i32, i32, vp, vp]),
    "irn_find_centroids": (i32, [vp, i32, i32, i32, vp, vp]),
    "irn_cluster_scratch_bytes": (sz, [i32, i32]),
    "irn_cluster_centroids": (i32, [vp, vp, i32, i32, f32, vp, C.POINTER(i32), vp, vp]),
    "irn_find_centroids_batch": (i32, [i32, ppv, pi32, pi32, i32, ppv, vp]),
    "irn_cluster_batch_scratch_bytes": (sz, [i32, pi32, pi32]),
    "irn_cluster_centroids_batch": (i32, [i32, ppv, ppv, pi32, pi32, f32, ppv, vp, vp, vp]),
    "irn_ccl_scratch_bytes": (sz, [i32, i32, i32]),
    "irn_label4": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "irn_detect_scratch_bytes": (sz, [i32, i32, i32]),
    "irn_detect_instance_count": (i32, [vp, vp, i32, i32, i32, C.POINTER(i32), vp, vp]),
    "irn_detect_instance_emit": (i32, [vp, vp, i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp]),
    "irn_detect_batch_scratch_bytes": (sz, [i32, pi32, pi32, pi32]),
    "irn_detect_instance_batch_count": (i32, [i32, ppv, ppv, pi32, pi32, pi32, vp, vp, vp]),
    "irn_detect_instance_batch_emit": (i32, [i32, ppv, ppv, pi32, pi32, pi32, pi32, C.POINTER(C.c_double), ppv, ppv, ppv, vp, vp]),
}

EXPORTS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args


class IrnHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib.irn_last_error()
        raise IrnHipError("libirn_hip status %d: %s" % (rc, msg.decode() if msg else "?"))


def i32_array(values):
    return (C.c_int32 * len(values))(*[int(v) for v in values])


def ptr_array(ptrs):
    """Host array of device pointers; None -> NULL."""
    return (C.c_void_p * len(ptrs))(*[None if p is None else int(p) for p in ptrs])
