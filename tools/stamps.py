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
eng = HotPathEngine(max_num_samples=3, is_adaptive="--fixed" not in sys.argv); eng.load_weights(w)
eng.precision = 0 if "f16" in sys.argv else 2 if "f32x" in sys.argv else 1
sched = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--schedule=")), 1)
eng.set_schedule(sched)
print("precision:", {0: "f16", 1: "f32", 2: "f32x"}[eng.precision], " schedule:", sched)
eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
for _ in range(3): eng.render()
lib = _lib.load(); lib.gdb_debug_set_buffer.argtypes = [ctypes.c_void_p]; lib.gdb_debug_set_buffer.restype = None
nblk = 16384
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device="cuda")
lib.gdb_debug_set_buffer(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.render(); e1.record(); torch.cuda.synchronize()
launch_us = e0.elapsed_time(e1) * 1e3
lib.gdb_debug_set_buffer(None)
raw_all = dbg.cpu().numpy().reshape(nblk, 16, 16)[:, :3, :12].astype(np.int64)
if "--by-xcd" in sys.argv:   # wave slots of a persistent launch: block index b & 7 = XCD (observed round-robin placement)
    for x in range(8):
        r = raw_all[x::8].reshape(-1, 12); r = r[(r[:, 0] > 0) & (r[:, 9] > 0)]
        if len(r):
            print(f"  XCD {x}: {len(r):5d} tiles, mean tile {np.mean(r[:, 9] - r[:, 0]):8.0f} cycles, last tile ends at {(r[:, 11].max() - raw_all[..., 10][raw_all[..., 10] > 0].min()) / 100.0:6.1f} us")
raw12 = raw_all.reshape(-1, 12)
raw12 = raw12[(raw12[:, 0] > 0) & (raw12[:, 9] > 0)]
raw = raw12[:, :10]
t = raw.reshape(-1, 10)
if sched not in (3, 4):
    t[:, 1] = t[:, 0]
t = t[(t[:, 0] > 0) & (t[:, 9] > 0)]
names = ["(dense / flat: plan + sample map)" if sched in (3, 4) else "(unused)", "bundle+vox+gather 3 views", "MLP mean/var+base", "MLP agg+fc", "MLP lr0+fh+shared", "MLP blend pass", "hand-off", "wait barrier", "composite+store"]
full = t[t[:, 6] > 0]  # waves that ran a slot
d = np.diff(full, axis=1).astype(np.float64)
tot = (full[:, 9] - full[:, 0]).mean()
print(f"{len(full)} active waves; mean lifetime {tot:.0f} cycles (s_memtime ticks)")
for n, m in zip(names, d.mean(0)):
    print(f"  {n:22s} {m:9.0f}  {100 * m / tot:5.1f} %")
rt0, rt1 = raw12[:, 10], raw12[:, 11]                      # s_memrealtime at wave start / end, 100 MHz, chip-wide
span_us = (rt1.max() - rt0.min()) / 100.0
clk = np.median((raw12[:, 9] - raw12[:, 0]) / np.maximum(rt1 - rt0, 1) * 0.1)   # GHz
busy = ((rt1 - rt0).sum() / 100.0) / span_us / 1024        # average waves resident per SIMD
print(f"launch: first wave start -> last wave end {span_us:.1f} us ({launch_us:.1f} us between events); shader clock while the waves ran "
      f"{clk:.2f} GHz (median, stamped build); {busy:.2f} waves resident per SIMD on average")

# concurrency over the launch: waves resident per SIMD at 16 instants (s_memrealtime, 100 MHz); shows fill, steady state and tail
ts = np.linspace(rt0.min(), rt1.max(), 18)[1:-1]
print("resident waves per SIMD over the launch:", " ".join(f"{((rt0 <= t) & (rt1 > t)).sum() / 1024:.2f}" for t in ts))
