"""Build libgdbnerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgdbnerf_hip.so")
SOURCES = ("gdb_ops.hip", "gdb_mlp.hip", "gdb_fused.hip", "gdb_costvol.hip", "gdb_merge.hip")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent f32 mul/add into v_pk_*_f32 with op_sel
# modifiers; in the fused kernel that produced stale values in lanes 48..63 of one packed result
# whenever two or more workgroups shared a CU (run-to-run different bundles, found with
# tools/dbg_fused.py; gone with one workgroup per CU or without packing).  Packed f32 math is no
# faster beside MFMA (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), so it is off.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# The operator mirrors keep a*b+c as two roundings (as separate torch ops are) unless written fmaf();
# the fused fast path lets the compiler contract.
CONTRACT = {"gdb_ops.hip": "off", "gdb_mlp.hip": "off", "gdb_fused.hip": "fast-honor-pragmas", "gdb_costvol.hip": "off", "gdb_merge.hip": "off"}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "gdb_nerf_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *os.environ.get("GDB_HIPCC_EXTRA", "").split(), f"-ffp-contract={CONTRACT[src]}", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
