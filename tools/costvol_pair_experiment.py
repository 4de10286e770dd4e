"""N2 cost volume (depth_net.py:424-476): planar source maps (two 8-byte x-pair loads per channel, row and view) against the channel-pair
re-layout [c/2][y][x][2] (one 16-byte load per channel PAIR, row and view; gdb_build_feature_volume_ws), same process, same inputs,
including the re-layout launch.      python tools/costvol_pair_experiment.py     (on the GPU box)"""
import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa
from gdb_nerf_amd import costvol, synthetic
from gdb_nerf_amd.networks.gdb_nerf import depth_net as dn


def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


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
        mid = t(fr["depth_range"]).mean(1, keepdim=True)
        dv = dn.get_depth_values(torch.cat((mid - 20, mid + 25), 1), D, False)
    E, Et = t(fr["src_exts"]), t(fr["tar_ext"])
    a = costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv, pair_layout=False)
    b = costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv, pair_layout=True)
    res[tag] = {"planar_us": round(timeit(lambda: costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv, pair_layout=False)), 1),
                "pair_layout_us (incl. the re-layout launch)": round(timeit(lambda: costvol.build_feature_volume(feat, E, Ks, Et, Kt, dv, inv, pair_layout=True)), 1),
                "bit_identical": bool(torch.equal(a, b)), "source_maps_MB": round(feat.numel() * 4 / 1e6, 2)}
print(json.dumps(res, indent=1))
