"""Build libgdbnerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python gdb-nerf_amd/build.py [--force] [--tag NAME --extra "FLAGS"]

The product library is `libgdbnerf_hip.so`.  Diagnostic / experiment variants (extra compiler flags, e.g. -DGDB_DIAG for
the stamp and ablation build) are written beside it as `libgdbnerf_hip.<tag>.so` with their own object directory and
never replace the product library; select one with the environment variable GDB_NERF_LIB (see _lib.py).
`libgdbpeaks.so` holds the attainable-peak micro-benchmarks bench.py runs (tools/ubench/peaks.hip) — measurement only.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgdbnerf_hip.so")
PEAKS_SRC = os.path.join(HERE, "..", "tools", "ubench", "peaks.hip")
PEAKS_LIB = os.path.join(HERE, "libgdbpeaks.so")
SOURCES = ("gdb_ops.hip", "gdb_mlp.hip", "gdb_fused.hip", "gdb_costvol.hip", "gdb_merge.hip", "gdb_decoder.hip")
# -fno-slp-vectorize: packed f32 VALU (v_pk_*_f32) beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md,
# "price of one filler beside MFMAs": +22..26 cycles per packed op) and hipcc's SLP pass packs adjacent f32 mul/add
# under plain -O3; DESIGN.md §4.1 has the history of the stale-lane corruption first seen with it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# The operator mirrors keep a*b+c as two roundings (as separate torch ops are) unless written fmaf();
# the fused fast path lets the compiler contract.
CONTRACT = {"gdb_ops.hip": "off", "gdb_mlp.hip": "off", "gdb_fused.hip": "fast-honor-pragmas", "gdb_costvol.hip": "off", "gdb_merge.hip": "off", "gdb_decoder.hip": "off"}


# Kernels that must not touch private memory at all (no vector spill, no struct parked in scratch).  The round-1 "packed f32"
# corruption was a private-memory round trip of a weight struct; every fused kernel keeps its data in registers.
NO_SCRATCH = ("k_render_",)


def parse_resource_usage(text: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel: {vgprs, sgprs, scratch, occupancy, ...}}."""
    res, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"),
                         ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"), ("waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return res


def lib_path(tag: str = "") -> str:
    return LIB if not tag else os.path.join(HERE, f"libgdbnerf_hip.{tag}.so")


def _flag_stamp(extra) -> str:
    return hashlib.sha1(" ".join([*FLAGS, *extra, *sorted(f"{k}={v}" for k, v in CONTRACT.items())]).encode()).hexdigest()


def _stale(lib: str, stamp_file: str, stamp: str) -> bool:
    if not os.path.exists(lib) or not os.path.exists(stamp_file) or open(stamp_file).read().strip() != stamp:
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(HERE, "..", "include", "gdb_nerf_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, tag: str = "", extra=()) -> str:
    """Build the library (tag "" = product flags only; the product build ignores the environment)."""
    extra = list(extra)
    if tag == "" and extra:
        raise ValueError("extra flags need a tag: the product library is always built with the default flags")
    lib = lib_path(tag)
    objdir = os.path.join(CSRC, "obj" + ("." + tag if tag else ""))
    stamp_file, stamp = os.path.join(objdir, "flags.sha1"), _flag_stamp(extra)
    if not force and not _stale(lib, stamp_file, stamp):
        return lib
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra, f"-ffp-contract={CONTRACT[src]}", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    usage = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        usage[src] = parse_resource_usage(out)
        rest = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and "remark:" not in l)
        if verbose and rest.strip():
            print(rest)
    try:  # the compiler the numbers (and the scratch finding of DESIGN.md §4.1) belong to
        ver = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout.strip().splitlines()
        usage["_toolchain"] = {"hipcc": " | ".join(l.strip() for l in ver if l.strip())[:400], "flags": " ".join([*FLAGS, *extra])}
    except OSError as e:
        usage["_toolchain"] = {"hipcc": f"unknown ({e})"}
    with open(os.path.join(objdir, "resource_usage.json"), "w") as f:
        json.dump(usage, f, indent=1)
    if not tag:  # the product build only: diagnostic builds (in-kernel stamps, ablation switches, the in-tree reproducer of the
        # private-memory corruption -DGDB_XP_PK=1) spend registers on their instrumentation and are never shipped
        bad = {k: v for src, u in usage.items() if not src.startswith("_") for k, v in u.items() if any(n in k for n in NO_SCRATCH) and v.get("scratch_bytes_per_lane", 0) > 0}
        if bad:
            raise RuntimeError(f"kernels that must keep their data in registers use private memory: {bad}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(stamp_file, "w") as f:
        f.write(stamp + "\n")
    return lib


def build_peaks(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(PEAKS_LIB) and os.path.getmtime(PEAKS_LIB) >= os.path.getmtime(PEAKS_SRC):
        return PEAKS_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-shared", PEAKS_SRC, "-o", PEAKS_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on peaks.hip:\n{r.stdout}")
    return PEAKS_LIB


if __name__ == "__main__":
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    extra = argv[argv.index("--extra") + 1].split() if "--extra" in argv else []
    print(build(force="--force" in argv, verbose=True, tag=tag, extra=extra))
    if not tag:
        print(build_peaks(force="--force" in argv, verbose=True))
