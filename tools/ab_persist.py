#!/usr/bin/env python3
"""Size of the persistent grid of k_render_dense against its launch duration, on one box (diagnostic build: the environment
variable GDB_DENSE_WGS_PER_CU overrides the workgroups per CU the launcher asks the runtime for).  With a grid at least as large
as the tile count every wave renders ONE tile - round 3's one-workgroup-per-tile schedule - so the same run is the A/B of the
persistent tile walk against it.  One child process per setting (the library reads the variable once), settings interleaved over
REPS rounds so that clock drift of the box hits all of them alike.

    python tools/ab_persist.py --case c2:f32 --wgs 0,2,4,5,6,7,64 [--steps 400] [--reps 3]      (0 = what the launcher picks)
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(case, steps, schedule=3):
    sys.path.insert(0, ROOT)
    import time, torch
    from bench import WORKLOADS, PREC, to_dev
    from gdb_nerf_amd import synthetic
    from gdb_nerf_amd.engine import HotPathEngine
    dev = torch.device("cuda", 0)
    wl_name, prec = case.split(":")
    wl = WORKLOADS[wl_name]
    fr = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], scene=wl["scene"], seed=0), dev)
    eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
    eng.set_schedule(schedule); eng.precision = PREC[prec]; eng.load_weights(synthetic.make_nerf_weights(seed=0))
    eng.prepare(fr)
    nb = eng.n_bundles
    o = (torch.zeros((nb, eng.Q), device=dev), torch.zeros((nb,), device=dev), torch.zeros((nb,), device=dev))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            eng.render(out=o)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.render(out=o)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    err = float((o[0] - eng.render_unfused()[0]).abs().max())
    print("RESULT " + json.dumps({"us": us, "err": err}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="c2:f32")
    ap.add_argument("--wgs", default="0,2,4,5,6,7,64")
    ap.add_argument("--persist", default="", help="comma list of GDB_DENSE_PERSIST values (1 = tile walk, 0 = one tile per wave) to run instead of --wgs: "
                                                  "the launcher's two forms against each other, whatever its policy picks")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--schedule", type=int, default=3, help="3 dense (windows of whole bundles), 4 flat (32 consecutive samples)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.case, a.steps, a.schedule)
    sys.path.insert(0, ROOT)
    import gdb_nerf_amd  # noqa: F401
    from gdb_nerf_amd import build as _b
    lib = _b.build(tag="diag", extra=["-DGDB_DIAG"])
    res = {}
    settings = [("persist " + v, {"GDB_DENSE_PERSIST": v}) for v in a.persist.split(",")] if a.persist else \
               [(w, {"GDB_DENSE_WGS_PER_CU": w} if int(w) > 0 else {}) for w in a.wgs.split(",")]
    for rep in range(a.reps):
        for w, extra in settings:
            env = dict(os.environ, GDB_NERF_LIB=lib)
            env.pop("GDB_DENSE_WGS_PER_CU", None); env.pop("GDB_DENSE_PERSIST", None)
            env.update(extra)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--case", a.case, "--steps", str(a.steps), "--schedule", str(a.schedule)],
                               env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(f"wgs {w}: child failed\n{p.stderr[-800:]}", flush=True)
                continue
            r = json.loads(line[-1][7:])
            res.setdefault(w, []).append(r)
    print(f"case {a.case}: k_render_dense launch duration (us per render call, plan ready; events around {a.steps} calls) by workgroups per CU")
    for w, rs in res.items():
        us = [r["us"] for r in rs]
        print(f"  {'wgs/CU ' if not a.persist else ''}{w:>3s}: " + " ".join(f"{u:7.2f}" for u in us) + f"   min {min(us):7.2f}   max|err| vs fp32 chain {max(r['err'] for r in rs):.2e}")


if __name__ == "__main__":
    main()
