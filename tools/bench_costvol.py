"""Time the cost-volume build at the DTU-eval stage shapes: HIP kernel vs the PyTorch
formulation (grid_sample + var, what the reference runs) on the same GPU, torch events, 50 iterations."""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import costvol, synthetic
from gdb_nerf_amd.networks.gdb_nerf import depth_net as dn

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

fr = synthetic.make_frame(512, 640, V=3, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
res = {}
for tag, (C, D, sf, sv, inv) in {"stage0 (C32, D64, 64x80 from 128x160 maps, disparity)": (32, 64, 0.25, 0.125, True),
                                 "stage1 (C16, D8, 256x320 from 256x320 maps)": (16, 8, 0.5, 0.5, False)}.items():
    Hs, Ws, Ht, Wt = int(512 * sf), int(640 * sf), int(512 * sv), int(640 * sv)
    feat = torch.randn(1, 3, C, Hs, Ws, device="cuda")
    Ks, Kt = t(fr["src_ints"]).clone(), t(fr["tar_int"]).clone()
    Ks[..., :2, :] *= sf; Kt[:, :2, :] *= sv
    nf = t(fr["near_far"])[..., None, None]
    if inv:
        dv = dn.get_depth_values(nf, D, True).expand(-1, -1, Ht, Wt).contiguous()
    else:
        # the last stage sweeps around the previous stage's (upsampled, hence smooth) depth: a low-pass depth prior, as
        # synthetic.make_frame builds it — white noise here would measure an incoherent gather no frame produces
        mid = t(fr["depth_range"]).mean(1, keepdim=True)
        dv = dn.get_depth_values(torch.cat((mid - 20, mid + 25), 1), D, False)
    E, Et = t(fr["src_exts"]), t(fr["tar_ext"])
    hip = timeit(lambda: costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv))
    ref = timeit(lambda: dn.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv))
    err = (costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv) - dn.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv)).abs().max().item()
    alg = 4 * (feat.numel() + dv.numel() + C * D * Ht * Wt)
    res[tag] = {"hip_us": round(hip, 1), "torch_us": round(ref, 1), "speedup": round(ref / hip, 2), "alg_MB": round(alg / 1e6, 2),
                "alg_GBps": round(alg / hip / 1e3, 1), "max_abs_diff": err}
print(json.dumps(res, indent=1))
