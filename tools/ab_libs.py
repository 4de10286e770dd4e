#!/usr/bin/env python3
"""A/B of library builds on the GPU box: per (workload, precision, schedule) the fused kernel's mean launch duration (HIP events
around the render call alone, prepare outside; pre-warmed) and its maximum deviation from the exact-fp32 operator chain, for each
build of libgdbnerf_hip (the product library and `libgdbnerf_hip.<tag>.so` variants; one child process per build, selected
through GDB_NERF_LIB).  The builds are interleaved over REPS rounds so that clock drift of the box hits all of them alike.

    python tools/ab_libs.py --libs base,product --cases c2:f32:1,c2:f32:3,c4:f32:0 [--steps 300] [--reps 3]

A workload name may carry an S_max / sampling override: `c2@6f` = the c2 frame at S_max 6 with fixed counts, `c2@8a` = S_max 8 adaptive.
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cases, steps):
    sys.path.insert(0, ROOT)
    import ctypes, numpy as np, torch
    if os.environ.get("GDB_NERF_LIB"):  # an older build of the ABI (a baseline from an earlier commit): bind what it exports
        from gdb_nerf_amd import _lib
        probe = ctypes.CDLL(os.environ["GDB_NERF_LIB"])
        for name in list(_lib._SIGNATURES):
            if not hasattr(probe, name):
                del _lib._SIGNATURES[name]
        _lib.ABI_VERSION = probe.gdb_abi_version()
        if not hasattr(probe, "gdb_render_info"):   # ABI < 6: the rule the engine then restated in Python (round 4's gdb_fixed_counts_dense)
            from gdb_nerf_amd import engine as _eng0
            _eng0.HotPathEngine.render_info = lambda self, *a, **k: {"fused": 1, "schedule": 0, "launches": 1, "kernel": None, "plan_built_by_prepare": int(
                bool(self.cfg.is_adaptive) or ((self.cfg.max_num_samples > 3 or self.cfg.max_num_samples == 2) and self._frame.V <= 3))}
        if not hasattr(probe, "gdb_prepare_rows"):  # ABI < 6: no strip-only plan
            L0 = _lib.load()
            L0.gdb_prepare_rows = lambda cfg, f, fpn, flags, r0, r1, ws, n, st: L0.gdb_prepare_ex(cfg, f, fpn, flags, ws, n, st)
        if not hasattr(probe, "gdb_prepare_ex"):   # ABI < 5: the engine's prepare call expressed in the older entry points
            L = _lib.load()
            L.gdb_prepare_ex = lambda cfg, f, fpn, flags, ws, n, st: (L.gdb_prepare_fpn(cfg, f, fpn, ws, n, st) if fpn else L.gdb_prepare(cfg, f, ws, n, st))
            from gdb_nerf_amd import engine as _eng
            _sched = _eng.HotPathEngine._sched
            _eng.HotPathEngine._sched = lambda self: _sched(self) & ~_lib.SCHED_PYR16_READY   # (a flag those builds reject)
    from bench import WORKLOADS, PREC, to_dev
    from gdb_nerf_amd import synthetic
    from gdb_nerf_amd.engine import HotPathEngine
    dev = torch.device("cuda", 0)
    w = synthetic.make_nerf_weights(seed=0)
    out = {}
    frames = {}
    for case in cases:
        wl_name, prec, sched = case.split(":")
        base_name, _, ovr = wl_name.partition("@")
        wl = dict(WORKLOADS[base_name])
        if ovr:
            wl["S"], wl["adaptive"] = int(ovr[:-1]), ovr[-1] == "a"
        if base_name not in frames:
            frames[base_name] = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], scene=wl["scene"], seed=0), dev)
        frames[wl_name] = frames[base_name]
        eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
        eng.set_schedule(int(sched)); eng.precision = PREC[prec]; eng.load_weights(w)
        eng.prepare(frames[wl_name])
        nb = eng.n_bundles
        o = (torch.zeros((nb, eng.Q), device=dev), torch.zeros((nb,), device=dev), torch.zeros((nb,), device=dev))
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(20):
                eng.render(out=o)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("AB_WITH_PREPARE"):   # the bench step: prepare + render, frames cycled through an HBM ring
            ring = [frames[wl_name]] + [{k: v.clone() for k, v in frames[wl_name].items()} for _ in range(6)]
            for i in range(50):
                eng.prepare(ring[i % 7]); eng.render(out=o)
            torch.cuda.synchronize()
            e0.record()
            for i in range(steps):
                eng.prepare(ring[i % 7]); eng.render(out=o)
            e1.record(); torch.cuda.synchronize()
        else:
            e0.record()
            for _ in range(steps):
                eng.render(out=o)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / steps * 1e3
        ref = eng.render_unfused()[0]
        err = float((o[0] - ref).abs().max())
        out[case] = (us, err)
        del eng, o, ref
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="product")
    ap.add_argument("--cases", default="c2:f32:1")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    cases = a.cases.split(",")
    if a.child:
        return child(cases, a.steps)
    libs = a.libs.split(",")
    res = {l: {c: [] for c in cases} for l in libs}
    for rep in range(a.reps):
        for l in libs:
            env = dict(os.environ)
            if l != "product":
                env["GDB_NERF_LIB"] = os.path.join(ROOT, "gdb-nerf_amd", f"libgdbnerf_hip.{l}.so")
            else:
                env.pop("GDB_NERF_LIB", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--cases", a.cases, "--steps", str(a.steps)],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
            if not line:
                print(f"[{l}] failed:\n{r.stdout[-2000:]}", flush=True)
                continue
            for c, v in json.loads(line[0][7:]).items():
                res[l][c].append(v)
        print(f"rep {rep} done", flush=True)
    print(f"{'case':14s} " + " ".join(f"{l:>28s}" for l in libs))
    for c in cases:
        cells = []
        for l in libs:
            v = res[l][c]
            cells.append("failed" if not v else f"{min(x[0] for x in v):7.1f} us (med {sorted(x[0] for x in v)[len(v)//2]:6.1f}) err {max(x[1] for x in v):.1e}")
        print(f"{c:14s} " + " ".join(f"{x:>28s}" for x in cells))


if __name__ == "__main__":
    main()
