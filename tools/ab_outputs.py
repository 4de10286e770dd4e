"""Bitwise A/B of two builds of the library on pyramid-dependent outputs.
usage: ab_outputs.py dump <lib.so> <out.npz>   |   ab_outputs.py cmp <a.npz> <b.npz>"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        ok &= same
        print(k, a[k].shape, "bit-identical" if same else f"DIFFERENT (max abs {np.abs(a[k] - b[k]).max():.3e})")
    sys.exit(0 if ok else 1)
import torch
from gdb_nerf_amd import _lib, synthetic
_lib.LIB_PATH = os.path.abspath(sys.argv[2])
from gdb_nerf_amd.engine import HotPathEngine
out = {}
cases = [(64, 80, 3, 1, 3, "dtu", (1.0, 2.3, 4.1)), (96, 72, 4, 2, 3, "nerf", (0.6, 1.9, 3.3)), (512, 640, 3, 1, 3, "dtu", (1.0, 2.0, 5.0)),
         (40, 104, 2, 1, 2, "llff", (3.0, 7.0))]   # W=52: ragged tiles; level sizes 26, 13 -> mip chain stops early
for i, (Ho, Wo, V, B, lv, scene, fs) in enumerate(cases):
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, scene=scene, seed=40 + i, src_focal_scale=fs)
    eng = HotPathEngine(max_num_samples=3, is_adaptive=True, max_mipmap_level=lv)
    eng.load_weights(synthetic.make_nerf_weights(seed=1))
    eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
    s = eng.sample()
    rfd, vox = eng.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"])
    n = int(s["total"].item())
    out[f"case{i}_rfd"] = rfd[:, :n].cpu().numpy()
    out[f"case{i}_fused"] = eng.render()[0].cpu().numpy()
np.savez(sys.argv[3], **out)
print("dumped", sys.argv[3])
