"""Debug aid: flat schedule against the dense one, bundle by bundle (which bundles differ: straddling ones, whole ones, row ends)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
Ho, Wo, S = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 96, 3)))
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fr = synthetic.make_frame(Ho, Wo, V=3, seed=3)
eng = HotPathEngine(max_num_samples=S, is_adaptive=True); eng.load_weights(synthetic.make_nerf_weights(seed=0)); eng.precision = prec
eng.prepare({k: torch.from_numpy(v).cuda() for k, v in fr.items()})
eng.set_schedule(3); a = [t.clone() for t in eng.render()]
eng.set_schedule(4); b = [t.clone() for t in eng.render()]
torch.cuda.synchronize()
cnt = eng.sample()["samples_per_bundle"].cpu().numpy().astype(np.int64)
start = np.concatenate(([0], np.cumsum(cnt)[:-1]))
strad = (start // 32) != ((start + cnt - 1) // 32)
d = (a[0] - b[0]).abs().max(dim=1).values.cpu().numpy()
dd = (a[1] - b[1]).abs().cpu().numpy(); do = (a[2] - b[2]).abs().cpu().numpy()
bad = d > 1e-5
print(f"bundles {len(cnt)}, samples {cnt.sum()}, windows {-(-cnt.sum() // 32)}, straddling {strad.sum()}")
print(f"differing bundles: {bad.sum()} (straddling among them {np.logical_and(bad, strad).sum()}; whole {np.logical_and(bad, ~strad).sum()}); max diff feat {d.max():.3e} depth {dd.max():.3e} opac {do.max():.3e}")
idx = np.nonzero(bad)[0][:12]
H, W = Ho // 2, Wo // 2
for i in idx: print(f"  bundle {i} (row {i // W}, x {i % W}) cnt {cnt[i]} start {start[i]} (window {start[i] // 32} lane {start[i] % 32}) strad {bool(strad[i])} diff {d[i]:.3e}  nan_flat {bool(torch.isnan(b[0][i]).any())}")
