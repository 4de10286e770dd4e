"""Phase-share profile of k_render_fused from in-kernel s_memtime stamps (diagnostic build:
-DGDB_DEBUG_STAMPS).  Shares, not absolute times (the stamps' waits forbid overlaps)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa: F401  (registers the package alias)
from gdb_nerf_amd import build as _b
# the diagnostic library is built beside the product one and selected for this process only
os.environ["GDB_NERF_LIB"] = _b.build(tag="diag", extra=["-DGDB_DIAG", "-DGDB_DEBUG_STAMPS"])
from gdb_nerf_amd import synthetic, _lib
from gdb_nerf_amd.engine import HotPathEngine
frame = synthetic.make_frame(512, 640, V=3, seed=0); w = synthetic.make_nerf_weights(seed=0)
eng = HotPathEngine(max_num_samples=3, is_adaptive=True); eng.load_weights(w)
eng.precision = 0 if "f16" in sys.argv else 1
sched = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--schedule=")), 1)
eng.set_schedule(sched)
print("precision:", "f16" if eng.precision == 0 else "f32", " schedule:", sched)
eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
for _ in range(3): eng.render()
lib = _lib.load(); lib.gdb_debug_set_buffer.argtypes = [ctypes.c_void_p]; lib.gdb_debug_set_buffer.restype = None
nblk = 16384
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device="cuda")
lib.gdb_debug_set_buffer(dbg.data_ptr())
eng.render(); torch.cuda.synchronize()
lib.gdb_debug_set_buffer(None)
t = dbg.cpu().numpy().reshape(nblk, 16, 16)[:, :3, :10].reshape(-1, 10).astype(np.int64)
if sched != 3:
    t[:, 1] = t[:, 0]
t = t[(t[:, 0] > 0) & (t[:, 9] > 0)]
names = ["(dense: plan + sample map)" if sched == 3 else "(unused)", "bundle+vox+gather 3 views", "MLP mean/var+base", "MLP agg+fc", "MLP lr0+fh+shared", "MLP blend pass", "hand-off", "wait barrier", "composite+store"]
full = t[t[:, 6] > 0]  # waves that ran a slot
d = np.diff(full, axis=1).astype(np.float64)
tot = (full[:, 9] - full[:, 0]).mean()
print(f"{len(full)} active waves; mean lifetime {tot:.0f} cycles (s_memtime ticks)")
for n, m in zip(names, d.mean(0)):
    print(f"  {n:22s} {m:9.0f}  {100 * m / tot:5.1f} %")
print("kernel span (first start -> last end):", (t[:, 9].max() - t[:, 0].min()), "ticks")
