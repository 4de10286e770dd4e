"""Where the tail of the persistent flat / dense launch comes from: per wave slot the durations of its tiles and its finish time
(s_memrealtime, 100 MHz, chip-wide), from the stamped diagnostic build.  usage: stamps_tail.py [--schedule=4]"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa: F401
from gdb_nerf_amd import build as _b
os.environ["GDB_NERF_LIB"] = _b.build(tag="diag", extra=["-DGDB_DIAG", "-DGDB_DEBUG_STAMPS"])
from gdb_nerf_amd import synthetic, _lib
from gdb_nerf_amd.engine import HotPathEngine
frame = synthetic.make_frame(512, 640, V=3, seed=0); w = synthetic.make_nerf_weights(seed=0)
eng = HotPathEngine(max_num_samples=3, is_adaptive=True); eng.load_weights(w); eng.precision = 1
sched = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--schedule=")), 4)
eng.set_schedule(sched)
eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
for _ in range(3): eng.render()
lib = _lib.load(); lib.gdb_debug_set_buffer.argtypes = [ctypes.c_void_p]; lib.gdb_debug_set_buffer.restype = None
nblk = 16384
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device="cuda")
lib.gdb_debug_set_buffer(dbg.data_ptr()); eng.render(); torch.cuda.synchronize(); lib.gdb_debug_set_buffer(None)
raw = dbg.cpu().numpy().reshape(nblk, 16, 16)[:, :4, :12].astype(np.int64)
# record layout (gdb_fused.hip STAMP): row = iteration * gridDim.x + block, column = wave of the block, 16 stamps each
rows = np.nonzero((raw[:, :, 0] > 0).any(1))[0]
G = int(rows.max() + 1 + 1) // 2 if rows.max() >= 1024 else int(rows.max() + 1)
nit = int(np.ceil((rows.max() + 1) / G))
nwv = int((raw[:G, :, 0] > 0).any(0).sum())
tl = np.stack([raw[i * G:(i + 1) * G, :nwv] for i in range(nit)], 0)       # [iteration][block][wave][stamp]
ok = (tl[..., 0] > 0) & (tl[..., 9] > 0)
t0 = tl[..., 10][ok].min()
dur = (tl[..., 11] - tl[..., 10]) / 100.0
beg = (tl[..., 10] - t0) / 100.0
end = (tl[..., 11] - t0) / 100.0
print(f"schedule {sched}: grid {G} workgroups x {nwv} waves, {nit} tile iterations, {int(ok.sum())} tiles")
P = lambda x, q: np.percentile(x, q)
for it in range(nit):
    m = ok[it]
    d, b, e = dur[it][m], beg[it][m], end[it][m]
    print(f"  iteration {it}: n {m.sum():5d}  starts p5 {P(b,5):5.1f} p50 {P(b,50):5.1f} p95 {P(b,95):5.1f} | duration us p5 {P(d,5):5.1f} p25 {P(d,25):5.1f} p50 {P(d,50):5.1f} p75 {P(d,75):5.1f} p95 {P(d,95):5.1f} max {d.max():5.1f} |"
          f" ends p5 {P(e,5):5.1f} p50 {P(e,50):5.1f} p95 {P(e,95):5.1f} max {e.max():5.1f}")
last = np.where(ok[nit - 1], end[nit - 1], end[0])
fin = last[ok[0]]
print(f"  wave finish time us: p5 {P(fin,5):.1f} p25 {P(fin,25):.1f} p50 {P(fin,50):.1f} p75 {P(fin,75):.1f} p95 {P(fin,95):.1f} max {fin.max():.1f}")
if nit >= 2:
    both = ok[0] & ok[1]
    print(f"  correlation of a wave's first and second tile durations: {np.corrcoef(dur[0][both], dur[1][both])[0, 1]:.2f}; first tile of wave 0 vs wave 1 of a workgroup: {np.corrcoef(dur[0][:, 0][ok[0][:, 0] & ok[0][:, 1]], dur[0][:, 1][ok[0][:, 0] & ok[0][:, 1]])[0, 1]:.2f}")
    # by dispatch round of the workgroup on its CU: block index within the XCD (b >> 3) / 32
    rnd = (np.arange(G) >> 3) // 32
    for r in sorted(set(rnd)):
        m = (rnd == r)[:, None] & ok[0]
        m1 = (rnd == r)[:, None] & ok[1]
        print(f"    dispatch round {r}: first tile {dur[0][m].mean():5.1f} us, second {dur[1][m1].mean():5.1f} us, wave ends at {end[1][m1].mean():5.1f} us (max {end[1][m1].max():5.1f})")
phase = np.diff(tl[..., :10], axis=-1)[ok]
names = ["plan", "gather", "mean/var", "agg+fc", "lr0+fh", "blend", "hand-off", "barrier", "composite"]
tot = (tl[..., 9] - tl[..., 0])[ok]
fast, slow = tot <= np.percentile(tot, 10), tot >= np.percentile(tot, 90)
print("  phase cycles, fastest 10 % of tiles vs slowest 10 %:")
for i, n in enumerate(names):
    print(f"    {n:10s} {phase[fast][:, i].mean():8.0f} {phase[slow][:, i].mean():8.0f}")
