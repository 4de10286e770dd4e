#!/usr/bin/env python3
"""How much of a c2 render is launch ramp + tail: the same frame shape rendered B = 1, 2, 4 frames per call (GdbFrame.B; dense = ONE launch over
all batch items, flat = one launch per item), us per FRAME.  What a finer work unit / any tail fix could return at most is the gap to B = 4.
usage: xp_batch.py [STEPS=200]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS, PREC, to_dev
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
w = synthetic.make_nerf_weights(seed=0)
for wl_name, pname in (("c2", "f32"), ("c2", "f16"), ("c4", "f32")):
    wl = WORKLOADS[wl_name]
    for sched in (3, 4) if pname == "f32" and wl_name == "c2" else (3,):
        row = []
        for B in (1, 2, 4):
            frame = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], B=B, scene=wl["scene"], seed=0), dev)
            eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
            eng.set_schedule(sched); eng.precision = PREC[pname]; eng.load_weights(w); eng.prepare(frame)
            out = torch.zeros((eng.n_bundles, eng.Q + 2), device=dev)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                for _ in range(10):
                    eng.render_packed(out=out)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(STEPS):
                eng.render_packed(out=out)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / STEPS * 1e3 / B)
            del eng, frame, out
        print(f"{wl_name} {pname} schedule {sched}: us per frame at B = 1, 2, 4: " + " ".join(f"{x:7.1f}" for x in row), flush=True)
