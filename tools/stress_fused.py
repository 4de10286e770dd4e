"""One-off stress: N seeded random shapes x the four schedules x the three precisions, fused kernel vs the fp32 operator chain (same checks as
tests/test_hip_parity.py::test_fused_random_shapes_vs_fp32_chain).  usage: stress_fused.py [N] [seed]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst, bad = 0.0, 0
for i in range(N):
    Ho, Wo = 2 * int(rng.integers(4, 130)), 2 * int(rng.integers(4, 200))
    c = dict(V=int(rng.integers(2, 9)), B=int(rng.integers(1, 4)), S=int(rng.integers(1, 17)), adaptive=bool(rng.integers(0, 2)),
             inv=bool(rng.integers(0, 2)), levels=int(rng.integers(0, 4)), scene=["dtu", "llff", "nerf"][int(rng.integers(0, 3))],
             fs=tuple(float(x) for x in rng.uniform(0.3, 8.0, size=3)))
    frame = synthetic.make_frame(Ho, Wo, V=c["V"], B=c["B"], scene=c["scene"], seed=1000 + i, src_focal_scale=c["fs"])
    eng = HotPathEngine(max_num_samples=c["S"], is_adaptive=c["adaptive"], inv_depth=c["inv"], max_mipmap_level=c["levels"])
    eng.load_weights(synthetic.make_nerf_weights(seed=i))
    eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
    ubf, ud, uo = [t.cpu().numpy() for t in eng.render_unfused()]
    for sched in (1, 2, 3, 4):
        eng.set_schedule(sched)
        for prec, tol in ((1, 1e-3), (2, 1e-3), (0, 3e-3)):
            bf, d, o = [t.cpu().numpy() for t in eng.render(precision=prec)]
            e = float(np.abs(bf - ubf).max()) if np.isfinite(bf).all() else float("inf")
            eo = float(np.abs(o - uo).max())
            if prec: worst = max(worst, e)
            if not (e <= tol and eo <= 1e-5):
                bad += 1
                print(f"FAIL case {i} {Ho}x{Wo} {c} schedule {sched} precision {prec}: bundle_feat err {e:.3e}, opacity err {eo:.3e}", flush=True)
    # GDB_SCHED_AUTO and the flat schedule (whose windows fall differently in every strip), packed rows, two row strips cut at a
    # random row against the whole frame: bit for bit (the multi-GPU unit); and flat against dense: bit for bit
    H = Ho // 2
    cut = int(rng.integers(0, H + 1))
    eng.set_schedule(3); dense = eng.render_packed(precision=1).clone()
    for sch in (4, 0):
        eng.set_schedule(sch)
        full = eng.render_packed(precision=1).clone()
        part = torch.full_like(full, float("nan"))
        eng.render_packed(0, cut, 1, part); eng.render_packed(cut, H, 1, part)
        if not torch.equal(full, part):
            bad += 1
            print(f"FAIL case {i} {Ho}x{Wo} {c} schedule {sch}: row strips [0,{cut}) + [{cut},{H}) differ from the full frame", flush=True)
        if sch == 4 and not torch.equal(full, dense):
            bad += 1
            print(f"FAIL case {i} {Ho}x{Wo} {c}: flat differs from dense (max {float((full - dense).abs().max()):.3e})", flush=True)
    if not (np.abs(full[:, :39].cpu().numpy() - ubf).max() <= 1e-3):
        bad += 1
        print(f"FAIL case {i} {Ho}x{Wo} {c}: packed AUTO render off the fp32 chain", flush=True)
    if i % 50 == 49: print(f"{i + 1} cases, worst err so far {worst:.3e}, failures {bad}", flush=True)
print(f"done: {N} cases x 4 schedules x 3 precisions, worst fp32-grade bundle_feat err {worst:.3e}, failures {bad}")
sys.exit(1 if bad else 0)
