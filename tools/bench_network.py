"""t_frame: whole Network.forward (CNNs on PyTorch-ROCm + hot path on the HIP library) at DTU eval 512x640, 3 source
views, random-init weights, the reference's timing protocol (run.py:56-73: synchronise, wall clock, drop the first
iteration, FPS = 1 / mean).  Prints a JSON object; hot_path in {fused, mirrors}, precision, hip_cost_volume and hip_decoder on/off."""
import json, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.configs import make_cfg
from gdb_nerf_amd.networks import make_network

fr = synthetic.make_frame(512, 640, V=3, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = {"src_views": {"rgb": t(fr["src_images"]), "extrinsics": t(fr["src_exts"]), "intrinsics": t(fr["src_ints"])},
         "tar_views": {"extrinsics": t(fr["tar_ext"]), "intrinsics": t(fr["tar_int"])}, "near_far": t(fr["near_far"])}
res = {}
for name, opts in {"fused (fp32 MFMA) + hip cost volume + hip decoder [default]": [],
                   "split-f16 pairs (fp32-grade) in the fused MLP and the decoder + hip cost volume": ["nerf.precision", "f32x"],
                   "fused + hip cost volume, torch decoder": ["nerf.hip_decoder", "False"],
                   "fused f16 operands + hip cost volume + hip decoder": ["nerf.precision", "f16"],
                   "fused, torch cost volume, torch decoder": ["mvs.hip_cost_volume", "False", "nerf.hip_decoder", "False"],
                   "operator mirrors + hip cost volume, torch decoder": ["nerf.hot_path", "mirrors", "nerf.hip_decoder", "False"]}.items():
    torch.manual_seed(0)
    net = make_network(make_cfg("configs/dtu_eval.yaml", opts + ["nerf.reuse_outputs", "True"])).eval().cuda()
    times = []
    with torch.no_grad():
        for i in range(16):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ret, _, _ = net(batch)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.mean(times[4:]))   # the first configuration also pays MIOpen's per-shape set-up in its first iterations
    res[name] = {"ms_per_frame": round(ms, 3), "fps": round(1e3 / ms, 1), "rays_per_s": round(512 * 640 / ms * 1e3)}
print(json.dumps(res, indent=1))
