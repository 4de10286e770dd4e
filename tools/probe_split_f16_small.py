"""Probe (GPU): error of the split-f16 MLP (GDB_PREC_F32X) against the exact fp32 MLP when weights / features are SMALL - low halves
of values below 2^-4 are f16 subnormals, below 6e-5 the high halves too.  Prints |f32x - f32| / output scale."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
for wscale, fscale in [(1.0, 1.0), (0.2, 1.0), (0.05, 1.0), (0.01, 1.0), (0.001, 1.0), (1.0, 0.01), (0.05, 0.01), (0.01, 0.001)]:
    frame = synthetic.make_frame(96, 128, V=3, B=1, seed=5)
    frame["img_feat"] = (frame["img_feat"] * np.float32(fscale)).astype(np.float32)
    frame["feat_volume"] = (frame["feat_volume"] * np.float32(fscale)).astype(np.float32)
    w = {k: (v * np.float32(wscale)).astype(np.float32) for k, v in synthetic.make_nerf_weights(seed=8).items()}
    eng = HotPathEngine(max_num_samples=4, is_adaptive=True); eng.load_weights(w)
    eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
    ref = eng.render(precision=1)[0].clone(); x = eng.render(precision=2)[0].clone(); h = eng.render(precision=0)[0].clone()
    # the 8 feat_head channels are pure MLP outputs (the 31 blended ones are convex combinations of the inputs)
    fh = slice(31, 39)
    sc = float(ref[:, fh].abs().max())
    print(f"weights x{wscale:<6} features x{fscale:<6}: feat_head scale {sc:.3e}  |f32x-f32|/scale {float((x[:, fh]-ref[:, fh]).abs().max())/sc:.2e}  "
          f"|f16-f32|/scale {float((h[:, fh]-ref[:, fh]).abs().max())/sc:.2e}   blended: |f32x-f32| {float((x[:, :31]-ref[:, :31]).abs().max()):.2e}")
