"""ctypes binding of libgdbnerf_hip.so (C ABI: include/gdb_nerf_hip.h).

There is no CPU fallback: if the shared library is missing or does not load, importing the
operators raises.  Build it with `python gdb-nerf_amd/build.py` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GDB_NERF_LIB selects another build of the same ABI (tools/: the -DGDB_DIAG diagnostic library and A/B flag variants are
# built beside the product library as libgdbnerf_hip.<tag>.so and never overwrite it).
LIB_PATH = os.environ.get("GDB_NERF_LIB") or os.path.join(HERE, "libgdbnerf_hip.so")
ABI_VERSION = 7
PREC_F16, PREC_F32, PREC_F32X = 0, 1, 2
SCHED_AUTO, SCHED_SLOT_WAVES, SCHED_SEGMENT_WAVE, SCHED_DENSE, SCHED_FLAT = 0, 1, 2, 3, 4
SCHED_PLAN_READY = 0x100
SCHED_PYR16_READY = 0x200
PREP_PYR16 = 1
PREP_PYR16_ONLY = 2
PREP_SOURCES_READY = 4
PREP_STRIP_REACH = 8
PREP_STRIP_WHOLE = 16

GDB_OK, GDB_E_BADARG, GDB_E_SHAPE, GDB_E_HIP, GDB_E_WORKSPACE = 0, -1, -2, -3, -4
GDB_MAX_SAMPLES, GDB_MAX_MIP, GDB_MAX_VIEWS = 16, 3, 8


class GdbConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "bundle_size", "max_num_samples", "is_adaptive", "inv_depth", "global_num_depth",
        "max_mipmap_level", "feat_dim", "voxel_dim", "hid_dim", "viewdir_agg")]


class GdbFrame(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "V", "Ho", "Wo", "H", "W", "D")] + \
               [(n, C.c_void_p) for n in (
                   "d_src_images", "d_img_feat", "d_feat_volume", "d_depth_range", "d_vol_range",
                   "d_src_exts", "d_src_ints", "d_tar_exts", "d_tar_ints", "d_near_far")]


class GdbError(RuntimeError):
    pass


_P = C.c_void_p
_CFG, _FRM = C.POINTER(GdbConfig), C.POINTER(GdbFrame)
_SIGNATURES = {
    "gdb_abi_version": (C.c_int, []),
    "gdb_last_error": (C.c_char_p, []),
    "gdb_workspace_bytes": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_pyramid_layout": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_dense_plan_layout": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_dense_map_layout": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_packed_weight_floats": (C.c_int, [_CFG, C.POINTER(C.c_size_t)]),
    "gdb_pack_weights": (C.c_int, [_CFG, C.POINTER(_P), _P]),
    "gdb_prepare": (C.c_int, [_CFG, _FRM, _P, C.c_size_t, _P]),
    "gdb_prepare_fpn": (C.c_int, [_CFG, _FRM, _P, _P, C.c_size_t, _P]),
    "gdb_prepare_ex": (C.c_int, [_CFG, _FRM, _P, C.c_int32, _P, C.c_size_t, _P]),
    "gdb_prepare_rows": (C.c_int, [_CFG, _FRM, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    "gdb_pyramid16_layout": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_build_rays": (C.c_int, [_CFG, _FRM, _P, _P, _P, _P, _P, _P, _P]),
    "gdb_sample": (C.c_int, [_CFG, _FRM, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdb_encode": (C.c_int, [_CFG, _FRM, _P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "gdb_mlp": (C.c_int, [_CFG, _P, C.c_int32, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "gdb_composite": (C.c_int, [_CFG, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "gdb_render_weights": (C.c_int, [_CFG, _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "gdb_accumulate": (C.c_int, [_CFG, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "gdb_build_feature_volume": (C.c_int, [_P] * 6 + [C.c_int32] * 9 + [_P, _P, _P]),
    "gdb_build_feature_volume_ws": (C.c_int, [_P] * 6 + [C.c_int32] * 9 + [_P, _P, _P, _P]),
    "gdb_depth_regression": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, _P, _P]),
    "gdb_render_bundles_fused": (C.c_int, [_CFG, _FRM, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "gdb_render_bundles_packed": (C.c_int, [_CFG, _FRM, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "gdb_render_info": (C.c_int, [_CFG, _FRM, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "gdb_merge": (C.c_int, [_CFG, _FRM, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "gdb_merge_packed": (C.c_int, [_CFG, _FRM, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "gdb_decoder_packed_floats": (C.c_int, [_CFG, C.c_int32, C.POINTER(C.c_size_t)]),
    "gdb_pack_decoder_weights": (C.c_int, [_CFG, C.c_int32, C.POINTER(_P), _P]),
    "gdb_decoder_workspace_bytes": (C.c_int, [_CFG, _FRM, C.POINTER(C.c_size_t)]),
    "gdb_decode": (C.c_int, [_CFG, _FRM, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load the library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdbError(f"{LIB_PATH} not found: the HIP hot path is not built "
                           "(run `python gdb-nerf_amd/build.py`); there is no CPU fallback")
        # torch bundles its own HIP runtime (SONAME-compatible with the one hipcc links against).
        # It must be in the process first so that this library binds to the SAME runtime as the
        # tensors and streams it is handed; loading in the other order maps two runtimes.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.gdb_abi_version() != ABI_VERSION:
            raise GdbError(f"ABI version {lib.gdb_abi_version()} != {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int) -> None:
    """Status -> exception.  Bad arguments / shapes raise ValueError, as the reference's
    Python operators do (bundle_sampler.py:220-221, network.py:33-34)."""
    if rc == GDB_OK:
        return
    msg = load().gdb_last_error().decode("utf-8", "replace")
    if rc in (GDB_E_BADARG, GDB_E_SHAPE):
        raise ValueError(msg)
    raise GdbError(f"[{rc}] {msg}")
