"""k_render_dense inside Network.forward against the bench's launch of it (135 us in situ vs 106 us in the bench).
Run under `rocprofv3 --kernel-trace --stats` (program after `--`, no torch profiler) for the per-kernel averages of the frame; the
script itself prints what the in-situ launch renders - the ACTUAL sample count and window (tile) count of the frame the random-init
CNNs produce, which are not the bench frame's - and times, on the same box with HIP events, (a) the in-situ frame's render alone
and (b) the bench's synthetic c2 frame's render alone.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft -- python3 $ROOT/tools/frame_trace.py
"""
import json, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gdb_nerf_amd  # noqa: F401
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.configs import make_cfg
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks import make_network

N = int(os.environ.get("FRAMES", "60"))
fr = synthetic.make_frame(512, 640, V=3, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = {"src_views": {"rgb": t(fr["src_images"]), "extrinsics": t(fr["src_exts"]), "intrinsics": t(fr["src_ints"])},
         "tar_views": {"extrinsics": t(fr["tar_ext"]), "intrinsics": t(fr["tar_int"])}, "near_far": t(fr["near_far"])}
torch.manual_seed(0)
net = make_network(make_cfg("configs/dtu_eval.yaml", ["nerf.reuse_outputs", "True"])).eval().cuda()
times = []
with torch.no_grad():
    for i in range(N):
        torch.cuda.synchronize(); t0 = time.perf_counter(); net(batch); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
eng = net._engine


def tiles(e):
    f = e._need_frame()
    return int(e.dense_plan()[:, 0].sum().item()), f.B * f.H


def render_us(e, steps=300):
    for _ in range(50):
        e.render_packed()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        e.render_packed()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


res = {"t_frame_ms": 1e3 * float(np.mean(times[5:]))}
ns = int(eng.sample()["total"].item())
tl, rows = tiles(eng)
res["in_situ"] = {"n_samples": ns, "samples_per_bundle": ns / eng.n_bundles, "windows": tl, "windows_per_row": tl / rows,
                  "render_alone_us": render_us(eng)}
# the bench's frame on an engine of its own, same box, same process
e2 = HotPathEngine(max_num_samples=3, is_adaptive=True)
e2.load_weights(synthetic.make_nerf_weights(seed=0))
e2.prepare({k: t(v) for k, v in fr.items()})
ns2 = int(e2.sample()["total"].item())
tl2, rows2 = tiles(e2)
res["bench_frame"] = {"n_samples": ns2, "samples_per_bundle": ns2 / e2.n_bundles, "windows": tl2, "windows_per_row": tl2 / rows2,
                      "render_alone_us": render_us(e2)}
res["note"] = ("the frame Network.forward renders comes from the random-init depth net's confidence interval, not from the bench's synthetic "
               "depth prior: its sample and window counts are its own; us per window is the comparable figure")
for k in ("in_situ", "bench_frame"):
    res[k]["us_per_1000_windows"] = res[k]["render_alone_us"] / res[k]["windows"] * 1e3
print(json.dumps(res, indent=1))
