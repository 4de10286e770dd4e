"""Where do two schedules of one build differ?  (debug aid)  usage: dbg_sched_diff.py [lib.so]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from gdb_nerf_amd import _lib, synthetic
if len(sys.argv) > 1:
    import ctypes
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    probe = ctypes.CDLL(_lib.LIB_PATH)
    for name in list(_lib._SIGNATURES):
        if not hasattr(probe, name):
            del _lib._SIGNATURES[name]
    _lib.ABI_VERSION = probe.gdb_abi_version()
from gdb_nerf_amd.engine import HotPathEngine
if len(sys.argv) > 1 and not hasattr(probe, "gdb_render_info"):
    HotPathEngine.render_info = lambda self, *a, **k: {"fused": 1, "schedule": 0, "launches": 1, "kernel": None, "plan_built_by_prepare": 1}
    L0 = _lib.load()
    L0.gdb_prepare_rows = lambda cfg, f, fpn, flags, r0, r1, ws, n, st: L0.gdb_prepare_ex(cfg, f, fpn, flags, ws, n, st)
frame = synthetic.make_frame(96, 144, V=3, B=1, seed=9)
eng = HotPathEngine(max_num_samples=5, is_adaptive=True)
eng.load_weights(synthetic.make_nerf_weights(seed=4))
eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
eng.precision = 0
outs = {}
for s in (1, 2, 3, 4):
    eng.set_schedule(s)
    outs[s] = [t.clone().cpu().numpy() for t in eng.render()]
np.savez(os.path.join(ROOT, "gpurun_out", "dbg_sched_" + (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product") + ".npz"), **{f"s{s}": outs[s][0] for s in outs})
cnt = eng.sample()["samples_per_bundle"].cpu().numpy()
for s in (2, 3, 4):
    d = np.abs(outs[1][0] - outs[s][0])
    rows = np.nonzero(d.max(axis=1) > 2e-6)[0]
    print(f"schedule 1 vs {s}: max {d.max():.3e}; bundles above 2e-6: {len(rows)}", rows[:12], "counts", cnt[rows[:12]], "x", rows[:12] % 72, "channels", [int(np.argmax(d[r])) for r in rows[:12]])
