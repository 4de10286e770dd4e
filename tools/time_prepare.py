"""k_prepare alone (HIP events around N back-to-back gdb_prepare calls, frames cycled through an HBM ring): with the dense plan
(adaptive config) and without (fixed counts).  usage: python tools/time_prepare.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
fr = synthetic.make_frame(512, 640, V=3, seed=0)
ring = [{k: torch.from_numpy(v).cuda() for k, v in fr.items()} for _ in range(7)]
for adaptive in (True, False, True, False):
    eng = HotPathEngine(max_num_samples=3, is_adaptive=adaptive)
    for i in range(50): eng.prepare(ring[i % 7])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(500): eng.prepare(ring[i % 7])
    e1.record(); torch.cuda.synchronize()
    print(f"adaptive={adaptive}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per prepare (call to call)")
