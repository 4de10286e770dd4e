"""k_plan alone (the dense plan + sample list of one frame, one wave per bundle-map row): run under rocprofv3 --kernel-trace --stats.
The engine is made to forget that gdb_prepare built the plan, so every render call rebuilds it in a launch of its own."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
for (Ho, Wo, S, scene) in ((512, 640, 3, "dtu"), (800, 800, 6, "nerf")):
    fr = synthetic.make_frame(Ho, Wo, V=3, scene=scene, seed=0)
    eng = HotPathEngine(max_num_samples=S, is_adaptive=True); eng.load_weights(synthetic.make_nerf_weights(seed=0))
    eng.prepare({k: torch.from_numpy(v).cuda() for k, v in fr.items()})
    for _ in range(200):
        eng._plan_key = None
        eng.render()
    torch.cuda.synchronize()
