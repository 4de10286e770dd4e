"""The C-ABI library loads, exports every symbol include/gdb_nerf_hip.h declares, and its
host-only entry points behave (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gdb_nerf_amd import _lib, synthetic


@pytest.fixture(scope="module")
def lib():
    from gdb_nerf_amd import build
    build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gdb_nerf_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(gdb_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.GdbConfig) == 40
    assert C.sizeof(_lib.GdbFrame) == 7 * 4 + 4 + 10 * 8  # 7 ints, pad, 10 pointers


def _cfg(**kw):
    d = dict(bundle_size=2, max_num_samples=3, is_adaptive=1, inv_depth=0, global_num_depth=64, max_mipmap_level=3,
             feat_dim=16, voxel_dim=8, hid_dim=64, viewdir_agg=1)
    d.update(kw)
    return _lib.GdbConfig(*d.values())


def _shape(B=1, V=3, Ho=512, Wo=640, b=2, D=8):
    return _lib.GdbFrame(B, V, Ho, Wo, Ho // b, Wo // b, D, *([None] * 10))


def test_workspace_bytes_and_errors(lib):
    n = C.c_size_t()
    assert lib.gdb_workspace_bytes(C.byref(_cfg()), C.byref(_shape()), C.byref(n)) == 0
    pyr = 3 * 20 * 4 * (256 * 320 + 128 * 160 + 64 * 80 + 32 * 40)
    assert pyr < n.value < pyr + 2 * 256 * 320 * 4 + 64 * 1024
    # reference error: network.py:33-34 ValueError('`Bundle size` must be a power of 2.')
    with pytest.raises(ValueError, match="power of 2"):
        _lib.check(lib.gdb_workspace_bytes(C.byref(_cfg(bundle_size=3)), C.byref(_shape()), C.byref(n)))
    with pytest.raises(ValueError):
        _lib.check(lib.gdb_workspace_bytes(C.byref(_cfg(max_num_samples=0)), C.byref(_shape()), C.byref(n)))
    with pytest.raises(ValueError, match="divisible"):
        _lib.check(lib.gdb_workspace_bytes(C.byref(_cfg()), C.byref(_shape(Ho=511)), C.byref(n)))
    with pytest.raises(ValueError, match="views"):
        _lib.check(lib.gdb_workspace_bytes(C.byref(_cfg()), C.byref(_shape(V=9)), C.byref(n)))
    with pytest.raises(ValueError, match="built for"):
        _lib.check(lib.gdb_workspace_bytes(C.byref(_cfg(hid_dim=128)), C.byref(_shape()), C.byref(n)))
    # a NULL device pointer is rejected before any launch
    with pytest.raises(ValueError, match="NULL"):
        _lib.check(lib.gdb_prepare(C.byref(_cfg()), C.byref(_shape()), None, 0, None))


def test_pack_weights_roundtrip(lib):
    w = synthetic.make_nerf_weights()
    n = C.c_size_t()
    cfg = _cfg()
    assert lib.gdb_packed_weight_floats(C.byref(cfg), C.byref(n)) == 0
    host = np.full(n.value, np.nan, np.float32)
    from gdb_nerf_amd.engine import NERF_KEYS
    arrs = [np.ascontiguousarray(w[k + s]) for k in NERF_KEYS for s in (".weight", ".bias")]
    ptrs = (C.c_void_p * 18)(*[a.ctypes.data for a in arrs])
    assert lib.gdb_pack_weights(C.byref(cfg), ptrs, host.ctypes.data) == 0
    assert np.isfinite(host).all()
    # the fp32 section keeps every tensor verbatim, in state-dict order
    pos = 0
    for a in arrs:
        flat = a.ravel()
        idx = None
        for start in range(pos, pos + 8):
            if np.array_equal(host[start:start + flat.size], flat):
                idx = start
                break
        assert idx is not None
        pos = idx + flat.size
    ptrs[0] = None
    with pytest.raises(ValueError, match="NULL"):
        _lib.check(lib.gdb_pack_weights(C.byref(cfg), ptrs, host.ctypes.data))
    assert lib.gdb_pack_weights(C.byref(_cfg(viewdir_agg=0)), ptrs, host.ctypes.data) == 0


def test_product_path_never_imports_oracle():
    """The package may not reach into oracle/ (the oracle is a checker, not a fallback)."""
    pkg = os.path.join(ROOT, "gdb-nerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "gdb_oracle" not in src, fn
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
