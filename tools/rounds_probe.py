"""Fused-kernel time vs number of workgroup 'rounds' (workgroups / resident slots) at fixed per-workgroup work."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
w = synthetic.make_nerf_weights(seed=0)
for Ho, Wo in ((512, 512), (512, 576), (512, 640), (512, 704), (512, 768), (512, 1024)):
    frame = synthetic.make_frame(Ho, Wo, V=3, seed=0)
    eng = HotPathEngine(max_num_samples=3, is_adaptive=True); eng.load_weights(w)
    eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
    out = eng.render()
    for _ in range(300): eng.render(0, None, 0, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): eng.render(0, None, 0, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 500 * 1e3
    wgs = (Ho // 2) * ((Wo // 2 + 31) // 32)
    print(f"{Ho}x{Wo}: {wgs} workgroups = {wgs / 1024:.2f} rounds of 1024; kernel {us:.1f} us; {us / wgs * 1024:.1f} us per 1024 workgroups; {Ho * Wo / us / 1e3:.2f} G rays/s (kernel only)", flush=True)
