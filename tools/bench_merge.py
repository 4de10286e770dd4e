"""gdb_merge (N1) against the torch ops it replaces (network.py:170-182) at DTU eval size; prints one JSON line."""
import json, os, sys, numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine
B, Ho, Wo, b = 1, 512, 640, 2
H, W = Ho // b, Wo // b
frame = synthetic.make_frame(Ho, Wo, V=3, seed=0)
eng = HotPathEngine(); eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
bf = torch.randn(B * H * W, eng.Q, device="cuda"); rgb_c = torch.randn(B, 3, Ho, Wo, device="cuda")
dep = torch.rand(B * H * W, device="cuda") * 500 + 400; opa = torch.rand(B * H * W, device="cuda")
def hip(): return eng.merge(bf, rgb_c, dep, opa, False)
def ref():
    nerf_feat = bf.view(B, H, W, -1).permute(0, 3, 1, 2)
    rgb_f = F.pixel_shuffle(nerf_feat[:, :12], b)
    up = lambda t: F.interpolate(t.view(B, 1, H, W), scale_factor=b, mode="bilinear", align_corners=False).squeeze(1)
    return rgb_c + rgb_f, up(dep), up(opa)
def tm(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_hip, t_ref = tm(hip), tm(ref)
alg = 4 * (B * H * W * 12 + 2 * B * 3 * Ho * Wo + 2 * B * H * W + 2 * B * Ho * Wo)  # 12 colours + rgb_c in, img out, 2 maps in/out
print(json.dumps({"op": "gdb_merge vs torch (pixel_shuffle + add + 2 x interpolate)", "shape": [B, Ho, Wo], "hip_us": t_hip, "torch_us": t_ref,
                  "speedup": t_ref / t_hip, "alg_bytes": alg, "achieved_GBs": alg / t_hip / 1e3}))
