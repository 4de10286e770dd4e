#!/usr/bin/env python3
"""Benchmark of the GDB-NeRF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2] [--path fused|unfused] [--shard frames|rows]

One *step* = one pass of the hot path (per-frame preparation: camera block + feature mip
pyramid; then build_rays → sample → encode → MLP → composite) over one synthetic frame of
the workload, inputs already resident in HBM.  Metric: rendered rays per second, whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU):
  --shard frames (default): every rank renders its own frame (independent target views, as an
      evaluation sweep does); no data-path collective; weak scaling.
  --shard rows: ONE frame, bundle-map row strips over the ranks, then an RCCL all-gather of
      the rendered strips (north_star's single-frame latency mode); strong scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at
N = 1, `cpu_baseline` (the numpy oracle timed on this host's cores — a reported baseline,
never part of the measured path).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gdb_nerf_amd import synthetic  # noqa: E402
from gdb_nerf_amd.engine import HotPathEngine  # noqa: E402
from gdb_nerf_amd.parallel import row_strip, gather_strips  # noqa: E402

# BASELINE.json configs -> (Ho, Wo, V, S_max, adaptive, scene)
WORKLOADS = {
    "c1": dict(Ho=64, Wo=80, V=3, S=3, adaptive=True, scene="dtu", desc="DTU-like 64x80 crop, 3 src views (CPU plumbing case)"),
    "c2": dict(Ho=512, Wo=640, V=3, S=3, adaptive=True, scene="dtu", desc="DTU eval 512x640, 3 src views, S_max 3 adaptive (configs/dtu_eval.yaml)"),
    "c3": dict(Ho=640, Wo=960, V=3, S=3, adaptive=True, scene="llff", desc="LLFF eval 640x960 (configs/llff_eval.yaml input_h_w)"),
    "c4": dict(Ho=800, Wo=800, V=3, S=6, adaptive=True, scene="nerf", desc="NeRF-synthetic eval 800x800, S_max 6 adaptive"),
    "c5": dict(Ho=1200, Wo=1600, V=5, S=6, adaptive=False, scene="dtu", desc="DTU full-res 1200x1600, 5 src views, S 6"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def alg_bytes(Ho, Wo, V, b=2, Cf=16, Cv=8, D=8, levels=3):
    """Algorithmic HBM bytes of one frame: every input read once (source
    images, feature pyramid incl. mips, cost volume, depth/vol ranges), outputs written once
    (Q+2 floats per bundle), MLP weights."""
    H, W = Ho // b, Wo // b
    pyr = sum((H >> l) * (W >> l) for l in range(levels + 1))
    Q = 3 * b * b + Cf + 3 + Cv
    return 4 * (V * 3 * Ho * Wo + V * (Cf + 3) * pyr + Cv * D * H * W + 4 * H * W + H * W * (Q + 2)) + 4 * 11930


def to_dev(frame, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in frame.items()}


def cpu_baseline(wl, frame, weights):
    """Time the oracle (CPU restatement of the reference path) on this host: whole frames of the
    workload, repeated until about 10 s of CPU work has been done (never more than ~30 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gdb_oracle  # the checker, used here only as the reported CPU baseline

    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = 1
    Ho, Wo = wl["Ho"], wl["Wo"]
    frames, t0 = 0, time.perf_counter()
    while True:  # whole frames of the same workload until ~10 s of CPU work (bounded at 30 s)
        gdb_oracle.hot_path(frame, weights, max_num_samples=wl["S"], is_adaptive=wl["adaptive"])
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or dt * (frames + 1) / frames > 30.0:
            break
    return {"value": frames * Ho * Wo / dt, "unit": "rays/s", "cores": int(cores), "kind": "port",
            "sample": f"{frames} full frame(s) of {Ho}x{Wo} through the numpy float32 oracle in {dt:.1f} s "
                      f"(BLAS matmuls on up to {cores} threads, element-wise parts on one)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)   # one step is ~0.1 ms: 2000 steps = a 0.2 s timed region
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps run for this long before the W warm-up steps, so clocks have ramped (0 = off)")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--path", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--shard", default="frames", choices=["frames", "rows"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="fused path only: render consecutive (independent) frames on this many HIP streams with their own workspaces, "
                         "so one frame's fill/drain overlaps the next; per-launch durations then overlap too (default 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    # GDB_BENCH_REHEARSE=1: several ranks share the visible GPU(s) over gloo — a logic rehearsal of the
    # N > 1 path on a one-GPU box, never a measurement.
    rehearse = os.environ.get("GDB_BENCH_REHEARSE") == "1"
    local = local % torch.cuda.device_count() if rehearse else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    wl = WORKLOADS[args.workload]
    Ho, Wo, V = wl["Ho"], wl["Wo"], wl["V"]
    seed = rank if (world > 1 and args.shard == "frames") else 0  # rows mode: every rank holds the same frame
    frame_np = synthetic.make_frame(Ho, Wo, V=V, scene=wl["scene"], seed=seed)
    weights_np = synthetic.make_nerf_weights(seed=0)
    frame = to_dev(frame_np, dev)
    if args.streams > 1 and (args.path != "fused" or (world > 1 and args.shard == "rows")):
        raise SystemExit("--streams > 1 is for the fused path on independent frames")
    H = Ho // 2
    r0, r1 = row_strip(H, rank, world) if args.shard == "rows" else (0, H)
    lanes = []  # one (engine, output buffers, stream) per stream; stream 0 is the current stream
    for i in range(max(1, args.streams)):
        e = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
        e.load_weights(weights_np)
        e.prepare(frame)
        nb = e.n_bundles
        o = (torch.zeros((nb, e.Q), device=dev), torch.zeros((nb,), device=dev), torch.zeros((nb,), device=dev))
        lanes.append((e, o, torch.cuda.current_stream(dev) if i == 0 else torch.cuda.Stream(dev)))
    torch.cuda.synchronize()
    eng, out, _ = lanes[0]
    nb = eng.n_bundles

    ev_pairs = []
    counter = [0]

    def step(timed):
        if args.streams > 1:  # frame i on stream i % n: prepare + render back to back on that stream
            e, o, st = lanes[counter[0] % len(lanes)]
            counter[0] += 1
            with torch.cuda.stream(st):
                e.prepare(frame)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                e.render(r0, r1, 0, o)
                if timed:
                    e1.record()
                    ev_pairs.append((e0, e1))
            return
        eng.prepare(frame)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if args.path == "fused":
            eng.render(r0, r1, 0, out)
        else:
            s = eng.sample()
            rfd, vox = eng.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"])
            if timed:  # dominant kernel of the unfused chain is the fp32 MLP
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            sigma, feat = eng.mlp(vox, rfd, s["total"])
            if timed:
                e1.record()
            eng.composite(sigma, feat, s["z_vals"], s["indices"], nb, s["total"])
        if timed and args.path == "fused":
            e1.record()
        if timed:
            ev_pairs.append((e0, e1))
        if world > 1 and args.shard == "rows":
            gather_strips(out[0], H, world, dist)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: a fresh process starts at idle clocks and a step is ~0.1 ms, so W warm-up steps alone can end before
    # the GPU reaches its sustained clock; run untimed steps for a fixed wall time first (not part of W or K)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(20):
            step(False)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    sync()
    ev_stride = max(1, args.steps // 100)  # kernel-duration events on ~100 evenly spaced steps of the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i % ev_stride == 0)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))
    frames_per_step = world if (world > 1 and args.shard == "frames") else 1
    rays_per_step = frames_per_step * Ho * Wo
    ms_per_step = dt / args.steps * 1e3
    value = rays_per_step * args.steps / dt

    # roofline of the dominant kernel (per launch, this rank)
    share = (r1 - r0) / H
    ab = alg_bytes(Ho, Wo, V) * share
    if args.path == "fused":
        kname = "k_render_fused"
    else:
        kname = "k_mlp"
        ns = int(eng.sample()["total"].item())
        ab = 4.0 * ns * (V * eng.P + 8 + 1 + eng.Q) + 4 * 11930  # what that kernel must read + write
    achieved = ab / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.workload}:{kname}")
        except Exception:
            traffic = None
    res = {
        "metric": "rendered rays/sec, GDB-NeRF hot path (sample+fetch+MLP+composite)", "value": value, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if (world > 1 and args.shard == "rows") else "weak",
        "vs_baseline": None, "dtype": "f32 fetch/composite, f16 MFMA MLP (f32 accumulate)" if args.path == "fused" else "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "bundle_size": 2, "rays_per_step": rays_per_step,
                   "path": args.path, "shard": args.shard if world > 1 else "none", "prewarm_ms": args.prewarm_ms, "streams": args.streams},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": kern_ms, "alg_bytes": ab},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(wl, frame_np, weights_np)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
