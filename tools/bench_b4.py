#!/usr/bin/env python3
"""What `bundle_size: 4` costs today (VERDICT r05 "missing 2" / task 5): the reference's 4x4 configuration (configs/dtu_pretrain.yaml:33
"bundle_size: 2  # 4 for 4*4", networks/gdb_nerf/network.py:31-34) goes through the HIP operator mirrors
(gdb_sample -> gdb_encode -> gdb_mlp -> gdb_composite, `HotPathEngine.render_unfused_packed`), the fused kernels are built for b = 2.
Same 512 x 640 frame shape, 3 source views, S_max 3 adaptive, same weights:

    b = 2 fused      prepare + gdb_render_bundles_packed                      (the bench headline's step)
    b = 2 mirrors    prepare + the operator chain                             (what b = 4 would cost if it were b = 2)
    b = 4 mirrors    prepare + the operator chain at bundle_size 4            (H x W = 128 x 160 bundles of 16 rays)

per step (HIP events over STEPS steps after a re-warm) and per RAY (both sizes render 512 x 640 = 327,680 rays), plus the chain's own
per-operator split at b = 4.  usage: bench_b4.py [STEPS=200]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import to_dev
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
Ho, Wo, V, S = 512, 640, 3, 3
w = synthetic.make_nerf_weights(seed=0)


def timed(fn, steps=STEPS, warm_s=0.3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


res = {"frame": f"{Ho}x{Wo}, V {V}, S_max {S} adaptive, synthetic (SURVEY 8(d) generator)", "rays": Ho * Wo, "steps": STEPS, "precision": "f32"}
for b in (2, 4):
    frame = to_dev(synthetic.make_frame(Ho, Wo, V=V, bundle_size=b, scene="dtu", seed=0), dev)
    eng = HotPathEngine(bundle_size=b, max_num_samples=S, is_adaptive=True, device=dev)
    eng.load_weights(w); eng.reuse_outputs = True
    eng.prepare(frame)
    info = eng.render_info()
    rec = {"bundles": eng.n_bundles, "fused_supported": bool(info["fused"]), "n_samples": int(eng.sample()["total"].item())}
    if info["fused"]:
        ms = timed(lambda: (eng.prepare(frame), eng.render_packed()))
        rec["fused_ms_per_step"] = ms; rec["fused_ns_per_ray"] = ms * 1e6 / (Ho * Wo)
    ms = timed(lambda: (eng.prepare(frame), eng.render_unfused_packed()))
    rec["mirrors_ms_per_step"] = ms; rec["mirrors_ns_per_ray"] = ms * 1e6 / (Ho * Wo)
    # the chain's operators one by one (each timed alone, inputs of the step before kept)
    s = eng.sample()
    rec["op_sample_ms"] = timed(lambda: eng.sample())
    rfd, vox = eng.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"])
    rec["op_encode_ms"] = timed(lambda: eng.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"]))
    sigma, feat = eng.mlp(vox, rfd, s["total"])
    rec["op_mlp_ms"] = timed(lambda: eng.mlp(vox, rfd, s["total"]))
    rec["op_composite_ms"] = timed(lambda: eng.composite(sigma, feat, s["z_vals"], s["indices"], eng.n_bundles, s["total"]))
    rec["op_prepare_ms"] = timed(lambda: eng.prepare(frame))
    res[f"b{b}"] = rec
    del eng, frame, s, rfd, vox, sigma, feat
# the b = 4 network's decoder (Decoder(upscale_factor=4): two up stages, decoder_rdn.py:59-62) on the HIP library (round 6) against the
# PyTorch-ROCm module it kept until then, on the 128 x 160 bundle map of a 512 x 640 frame
try:
    from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
    torch.manual_seed(0)
    dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=4).eval().to(dev)
    frame = to_dev(synthetic.make_frame(Ho, Wo, V=V, bundle_size=4, scene="dtu", seed=0), dev)
    eng = HotPathEngine(bundle_size=4, max_num_samples=S, is_adaptive=True, device=dev)
    eng.load_weights(w); eng.reuse_outputs = True; eng.prepare(frame)
    eng.load_decoder_weights({k: v.detach() for k, v in dec.state_dict().items()}, 3)
    packed = eng.render_packed().clone()
    x = packed[:, 48:75].view(1, Ho // 4, Wo // 4, 27).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        want = dec(x)
        res["b4_decoder"] = {"hip_fp32_ms": timed(lambda: eng.decode(packed, precision=1)), "hip_split_f16_ms": timed(lambda: eng.decode(packed, precision=2)),
                             "torch_miopen_ms": timed(lambda: dec(x)),
                             "max_abs_err_vs_torch": float((eng.decode(packed, precision=1) - want).abs().max()), "output_scale": float(want.abs().max())}
except Exception as ex:
    res["b4_decoder"] = {"error": repr(ex)}
f2 = res["b2"].get("fused_ns_per_ray")
if f2:
    res["b4_mirrors_over_b2_fused_per_ray"] = res["b4"]["mirrors_ns_per_ray"] / f2
print(json.dumps(res, indent=1))
