import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gdb_nerf_amd
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.configs import make_cfg
from gdb_nerf_amd.networks import make_network
fr = synthetic.make_frame(512, 640, V=3, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = {"src_views": {"rgb": t(fr["src_images"]), "extrinsics": t(fr["src_exts"]), "intrinsics": t(fr["src_ints"])},
         "tar_views": {"extrinsics": t(fr["tar_ext"]), "intrinsics": t(fr["tar_int"])}, "near_far": t(fr["near_far"])}
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    net = make_network(make_cfg("configs/dtu_eval.yaml", [])).eval().cuda()
    times = []
    with torch.no_grad():
        for i in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter(); net(batch); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print("cudnn.benchmark", bench, "ms/frame", 1e3 * float(np.mean(times[5:])), "first", times[0])
# where does the frame go?
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): net(batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
