#!/usr/bin/env python3
"""What a rank's prepare costs when ONE frame is split by rows over WORLD ranks (VERDICT r05 task 4): gdb_prepare_rows
builds only the pyramid tiles / image rows the strip's samples can reach (k_strip_bounds + k_prepare), against the whole-frame prepare
every rank ran until round 5.  One GPU emulates one rank after the other: per workload / precision the whole-frame prepare and the
prepare of each of the WORLD strips (HIP events over N back-to-back calls, the frame resident), the strip's render beside it, and the
share of the pyramid bytes the strip's prepare wrote.  usage: time_prepare_rows.py [WORLD=8] [N=200]"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS, PREC, to_dev
from gdb_nerf_amd import synthetic, _lib
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.parallel import row_strip

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
w = synthetic.make_nerf_weights(seed=0)


def timed(fn, n=N):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


res = {"world": WORLD, "calls": N, "note": "us per call, HIP events over back-to-back calls on one resident frame; strip k = rank k's rows"}
for wl_name, pname in (("c2", "f32"), ("c2", "f16"), ("c4", "f32"), ("c5", "f16")):
    wl = WORKLOADS[wl_name]
    H = wl["Ho"] // 2
    frame = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], scene=wl["scene"], seed=0), dev)
    eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
    eng.precision = PREC[pname]; eng.load_weights(w); eng.prepare(frame)
    eng.strip_reach = True   # (forced: by size the library builds the strip's reach only from ~128 MB of whole-frame tile work up)
    lay = (C.c_size_t * 7)()
    if pname == "f16":
        _lib.check(eng.lib.gdb_pyramid16_layout(C.byref(eng.cfg), C.byref(eng._frame), lay)); nbytes = int(lay[1]) * wl["V"]
    else:
        _lib.check(eng.lib.gdb_pyramid_layout(C.byref(eng.cfg), C.byref(eng._frame), lay)); nbytes = 4 * int(lay[1]) * wl["V"]
    region = eng._ws[int(lay[0]):int(lay[0]) + nbytes]
    out = torch.zeros((eng.n_bundles, eng.Q + 2), device=dev)
    rec = {"whole_frame_prepare_us": timed(lambda: eng.prepare(frame)), "strips": []}
    eng.prepare(frame)
    rec["whole_frame_render_us"] = timed(lambda: eng.render_packed(0, H, None, out), max(20, N // 4))
    for k in range(WORLD):
        r0, r1 = row_strip(H, k, WORLD)
        region.fill_(0x7F)
        eng.prepare(frame, rows=(r0, r1))
        torch.cuda.synchronize()
        share = float((region != 0x7F).float().mean())
        p_us = timed(lambda: eng.prepare(frame, rows=(r0, r1)))
        k_us = timed(lambda: eng.render_packed(r0, r1, None, out), max(20, N // 4))
        rec["strips"].append({"rows": [r0, r1], "prepare_us": p_us, "render_us": k_us, "pyramid_share_written": share})
    ps = [s["prepare_us"] for s in rec["strips"]]
    rec["strip_prepare_us_mean"] = sum(ps) / len(ps); rec["strip_prepare_us_max"] = max(ps)
    rec["strip_over_whole"] = rec["strip_prepare_us_mean"] / rec["whole_frame_prepare_us"]
    res[f"{wl_name}:{pname}"] = rec
    del eng, frame, out
print(json.dumps(res, indent=1))
