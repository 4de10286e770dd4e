"""One-off stress of the bundle_size 1 / 4 fused path (round 6): N seeded random shapes x bundle_size in {1, 4} x the three precisions, the fused
render (dense list kernel on centre rays + k_bundle_colours) against the fp32 operator chain (gdb_sample -> gdb_encode -> gdb_mlp ->
gdb_composite), packed rows = the three tensors, two row strips cut at a random row = the full frame bit for bit.
usage: stress_bundle_sizes.py [N] [seed]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst, worst16, bad = 0.0, 0.0, 0
for i in range(N):
    b = int(rng.choice([1, 4]))
    H, W = int(rng.integers(2, 70)), int(rng.integers(2, 90))
    Ho, Wo = H * b, W * b
    c = dict(V=int(rng.integers(2, 9)), B=int(rng.integers(1, 4)), S=int(rng.integers(1, 17)), adaptive=bool(rng.integers(0, 2)),
             inv=bool(rng.integers(0, 2)), levels=int(rng.integers(0, 4)), scene=["dtu", "llff", "nerf"][int(rng.integers(0, 3))],
             fs=tuple(float(x) for x in rng.uniform(0.3, 8.0, size=3)))
    frame = synthetic.make_frame(Ho, Wo, V=c["V"], B=c["B"], bundle_size=b, scene=c["scene"], seed=2000 + i, src_focal_scale=c["fs"])
    eng = HotPathEngine(bundle_size=b, max_num_samples=c["S"], is_adaptive=c["adaptive"], inv_depth=c["inv"], max_mipmap_level=c["levels"])
    eng.load_weights(synthetic.make_nerf_weights(seed=i))
    eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
    ubf, ud, uo = [t.cpu().numpy() for t in eng.render_unfused()]
    for prec, tol in ((1, 1e-3), (2, 1e-3), (0, 3e-3)):
        bf, d, o = eng.render(precision=prec)
        e = float((bf.cpu().numpy() - ubf).__abs__().max()) if torch.isfinite(bf).all() else float("inf")
        eo = float(np.abs(o.cpu().numpy() - uo).max())
        if prec: worst = max(worst, e)
        else: worst16 = max(worst16, e)
        packed = eng.render_packed(precision=prec).clone()
        ok = torch.equal(packed[:, :eng.Q], bf) and torch.equal(packed[:, eng.Q], d) and torch.equal(packed[:, eng.Q + 1], o)
        cut = int(rng.integers(0, H + 1))
        part = torch.full_like(packed, float("nan"))
        eng.render_packed(0, cut, prec, part); eng.render_packed(cut, H, prec, part)
        ok = ok and torch.equal(part, packed)
        if not (e <= tol and eo <= 1e-5 and ok):
            bad += 1
            print(f"FAIL case {i} b {b} {Ho}x{Wo} {c} precision {prec}: bundle_feat err {e:.3e}, opacity err {eo:.3e}, packed / strips equal {ok}", flush=True)
print(f"stress (bundle_size 1 / 4): {N} shapes x 3 precisions, {bad} failures; worst fp32 / split-f16 error vs the fp32 chain {worst:.3e}, f16 {worst16:.3e}")
sys.exit(1 if bad else 0)
