#!/usr/bin/env python3
"""Experiment (round 6): a start-of-launch stagger of the list kernels' waves by their dispatch rank on the CU (GDB_XP_STAGGER, diagnostic
build `libgdbnerf_hip.diag.so` = -DGDB_DIAG): n > 0 = the OLDER workgroups of a CU sleep (R - 1 - rank) x n x 4,096 cycles before their
first tile, n < 0 = the YOUNGER ones rank x |n| x 4,096.  Per (workload, precision, schedule) the kernel's mean duration (HIP events over
STEPS back-to-back renders, pre-warmed) for every n, interleaved over REPS rounds; results checked bit-identical to n = 0.
usage: GDB_NERF_LIB=.../libgdbnerf_hip.diag.so xp_stagger.py [STEPS=300] [REPS=3]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS, PREC, to_dev
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NS = (0, 1, 2, 3, 5, -1, -2, -3)
dev = torch.device("cuda", 0)
w = synthetic.make_nerf_weights(seed=0)
cases = (("c2", "f32", 4), ("c2", "f32", 3), ("c3", "f32", 0), ("c4", "f32", 0), ("c2", "f16", 0), ("c4", "f16", 0))
print("us per render; columns: GDB_XP_STAGGER =", NS, flush=True)
for wl_name, pname, sched in cases:
    wl = WORKLOADS[wl_name]
    frame = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], scene=wl["scene"], seed=0), dev)
    eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
    eng.set_schedule(sched); eng.precision = PREC[pname]; eng.load_weights(w); eng.prepare(frame)
    out = torch.zeros((eng.n_bundles, eng.Q + 2), device=dev)
    os.environ["GDB_XP_STAGGER"] = "0"
    ref = eng.render_packed().clone()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            eng.render_packed(out=out)
        torch.cuda.synchronize()
    best = {n: 1e9 for n in NS}
    for rep in range(REPS):
        for n in NS:
            os.environ["GDB_XP_STAGGER"] = str(n)
            for _ in range(20):
                eng.render_packed(out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(STEPS):
                eng.render_packed(out=out)
            e1.record(); torch.cuda.synchronize()
            best[n] = min(best[n], e0.elapsed_time(e1) / STEPS * 1e3)
            assert torch.equal(out, ref), (wl_name, pname, sched, n)
    print(f"{wl_name}:{pname}:{sched} ({eng.render_info()['kernel']}) " + " ".join(f"{best[n]:7.1f}" for n in NS), flush=True)
    del eng, frame, out, ref
