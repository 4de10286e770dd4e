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
    # camera block + pyramid + its half-precision copy (GDB_PREC_F16's taps) + per-bundle counts / offsets (mirror) + dense plan +
    # sample list (4 B per sample offset of a row)
    # ... and the flat schedule's straddling-bundle records: (81920 x 3 / 32 + 3) boundaries x 3 samples x 2 lane halves x 32 floats
    # (one 128-byte line per record since round 6), + headers
    flat = (256 * 320 * 3 // 32 + 3) * (3 * 2 * 32 * 4 + 8)
    # ... and the half-precision RGBA copy of the source images (GDB_PREC_F16's colour taps): 8 bytes per pixel and view
    img16 = 3 * 512 * 640 * 8
    assert pyr + pyr // 2 + img16 < n.value < pyr + pyr // 2 + img16 + 2 * 256 * 320 * 4 + 256 * (32 * 30 + 32) * 4 + flat + 96 * 1024
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


def test_header_is_plain_c_and_links_from_c(lib, tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (gcc, no C++), every declared entry point must be
    addressable from C, the structs must have the layout ctypes assumes, and the library must link from a C program
    (host-only calls: version, argument checking, packed-weight size — no GPU)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    hdr = open(os.path.join(ROOT, "include", "gdb_nerf_hip.h")).read()
    names = sorted(set(re.findall(r"^(?:int|const char\*)\s+(gdb_\w+)\s*\(", hdr, flags=re.M)))
    src = tmp_path / "abi.c"
    src.write_text("""
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include "gdb_nerf_hip.h"
int main(void) {
    void (*fns[])(void) = {%s};
    size_t i, n = 0;
    GdbConfig c; GdbFrame f;
    if (sizeof(GdbConfig) != 40 || sizeof(GdbFrame) != 112 || offsetof(GdbFrame, d_src_images) != 32) return 10;
    for (i = 0; i < sizeof fns / sizeof fns[0]; ++i) if (!fns[i]) return 11;
    if (gdb_abi_version() != GDB_ABI_VERSION || GDB_ABI_VERSION != 7) return 12;
    memset(&c, 0, sizeof c); memset(&f, 0, sizeof f);
    c.bundle_size = 3; c.max_num_samples = 3; c.global_num_depth = 64; c.feat_dim = 16; c.voxel_dim = 8; c.hid_dim = 64;
    if (gdb_packed_weight_floats(&c, &n) != GDB_E_BADARG) return 13;            /* bundle size must be a power of 2 */
    if (!strstr(gdb_last_error(), "power of 2")) return 14;
    c.bundle_size = 2;
    if (gdb_packed_weight_floats(&c, &n) != GDB_OK || n < 11930) return 15;
    /* argument checking happens on the host, before any launch: an unknown precision / schedule is refused */
    c.is_adaptive = 1; c.max_mipmap_level = 3; c.viewdir_agg = 1;
    f.B = 1; f.V = 3; f.Ho = 64; f.Wo = 80; f.H = 32; f.W = 40; f.D = 8;
    if (gdb_render_bundles_packed(&c, &f, NULL, NULL, 0, 32, GDB_PREC_F32, GDB_SCHED_AUTO, NULL, NULL) != GDB_E_BADARG) return 16;
    printf("%%zu entry points, %%zu packed floats\\n", sizeof fns / sizeof fns[0], n);
    return 0;
}
""" % ", ".join(f"(void (*)(void)){n}" for n in names))
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-l:libgdbnerf_hip.so", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    import torch  # the library's HIP runtime dependency resolves against the one torch ships
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert f"{len(names)} entry points" in r.stdout


def test_product_library_reads_no_environment():
    """no global mutable state and no hidden knobs in the product entry points — every getenv in the HIP
    sources sits inside an `#ifdef GDB_DIAG` block (the diagnostic build of tools/)."""
    csrc = os.path.join(ROOT, "gdb-nerf_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h")):
            continue
        depth = []  # stack of booleans: is this #if level a GDB_DIAG block
        for ln, line in enumerate(open(os.path.join(csrc, fn), encoding="utf-8"), 1):
            st = line.strip()
            if st.startswith(("#ifdef", "#ifndef", "#if ")):
                depth.append(st.startswith("#ifdef GDB_DIAG"))
            elif st.startswith("#endif") and depth:
                depth.pop()
            elif "getenv(" in st and not st.startswith("//"):
                assert any(depth), f"{fn}:{ln}: getenv outside the diagnostic build"
        assert not re.search(r"^static\s+(int|unsigned|float|bool)\s+g_\w+", open(os.path.join(csrc, fn)).read(), flags=re.M), fn
