#!/usr/bin/env python3
"""Benchmark of the GDB-NeRF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2] [--precision f32|f16|f32x] [--path fused|unfused]

One *step* = one pass of the hot path (per-frame preparation: camera block + feature mip pyramid; then
build_rays → sample → encode → MLP → composite) over one synthetic frame of the workload, inputs already
resident in HBM.  Metric: rendered rays per second, whole job.

Headline (N = 1): workload c2 = BASELINE.json configs[1] at `--precision f32` — the NeRF MLP on fp32 MFMA, the
reference's own precision (nerf.py:84-115 never leaves fp32).  The f16-operand path is timed in the same run and
reported as the `secondary` record.  The JSON line also carries `psnr_delta_db` (fused render vs the exact-fp32
operator chain on the benched frame), `hbm_frac` / `mfma_frac` against vendor peaks, `peaks_measured` (stream
triad and bare MFMA loops run on this node), `t_frame_ms` (whole Network.forward, the reference's run.py:56-73
protocol), `roofline` and `cpu_baseline`.

N > 1 (launched by torch.distributed.run, one rank per GPU), both modes timed in one run:
  rows   (the headline, north_star's partitioning: "rays shard across the GPUs of one node with an RCCL all-gather of rendered tiles";
         strong scaling): ONE frame, contiguous bundle-map row strips over the ranks, every rank prepares (strip-only plan,
         gdb_prepare_rows) and renders its strip straight into the packed (n_bundles, 41) buffer, a single RCCL
         all_gather_into_tensor of the strips; per rank `prepare_ms`, `kernel_ms`, `allgather_ms` and bus GB/s are reported and the
         gathered strips are checked bit for bit against a full render.  The same protocol on c4 and c5 (the
         meaningful multi-GPU points; a 512x640 frame is ~0.1 ms of work and its split is bound by the gather's latency) rides in
         the line as `rows_c4` / `rows_c5`.
  frames (the `independent_frames` record, weak scaling): every rank renders its own frame (independent target views, as an
         evaluation sweep does), no data-path collective; `--shard frames` makes it the headline.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
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
from gdb_nerf_amd.parallel import StripGather  # noqa: E402

# BASELINE.json configs -> (Ho, Wo, V, S_max, adaptive, scene)
WORKLOADS = {
    "c1": dict(Ho=64, Wo=80, V=3, S=3, adaptive=True, scene="dtu", desc="DTU-like 64x80 crop, 3 src views (CPU plumbing case)"),
    "c2": dict(Ho=512, Wo=640, V=3, S=3, adaptive=True, scene="dtu", desc="DTU eval 512x640, 3 src views, S_max 3 adaptive (configs/dtu_eval.yaml)"),
    "c3": dict(Ho=640, Wo=960, V=3, S=3, adaptive=True, scene="llff", desc="LLFF eval 640x960 (configs/llff_eval.yaml input_h_w)"),
    "c3p": dict(Ho=756, Wo=1008, V=3, S=3, adaptive=True, scene="llff", desc="LLFF 756x1008 (BASELINE.json configs[2] literal size; synthetic upstream tensors)"),
    "c4": dict(Ho=800, Wo=800, V=3, S=6, adaptive=True, scene="nerf", desc="NeRF-synthetic eval 800x800, S_max 6 adaptive"),
    "c5": dict(Ho=1200, Wo=1600, V=5, S=6, adaptive=False, scene="dtu", desc="DTU full-res 1200x1600, 5 src views, S 6"),
}
# vendor peaks (MI355X_MICROARCH.md): HBM3E 8 TB/s; dense matrix 2.5 PFLOP/s f16/bf16, 157.3 TFLOP/s f32-input MFMA
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "f32x": 2500.0 / 3}  # f32x: three f16 MFMAs per algorithmic product
DTYPE = {"f32": "f32", "f16": "f16 MFMA operands, f32 accumulate (fetch / geometry / composite f32)",
         "f32x": "split-f16 MFMA operands (hi + lo pairs, 3 MFMAs per product, ~22-bit), f32 accumulate (fetch / geometry / composite f32)"}
PREC = {"f16": 0, "f32": 1, "f32x": 2}


def alg_bytes(Ho, Wo, V, b=2, Cf=16, Cv=8, D=8, levels=3):
    """Algorithmic HBM bytes of one frame: every input read once (source
    images, feature pyramid incl. mips, cost volume, depth/vol ranges), outputs written once
    (Q+2 floats per bundle), MLP weights."""
    H, W = Ho // b, Wo // b
    pyr = sum((H >> l) * (W >> l) for l in range(levels + 1))
    Q = 3 * b * b + Cf + 3 + Cv
    return 4 * (V * 3 * Ho * Wo + V * (Cf + 3) * pyr + Cv * D * H * W + 4 * H * W + H * W * (Q + 2)) + 4 * 11930


def alg_flops(n_samples, V):
    """Algorithmic MLP flops: N_s (V * 18,200 + 5,248), the reference formulation (nerf.py:20-56),
    N_s = the ACTUAL adaptive sample count of the frame."""
    return float(n_samples) * (V * 18200 + 5248)


def kernel_source_sha16():
    """sha256 (16 hex digits) of the sources the fused kernels' code and memory traffic come from — gdb_fused.hip, gdb_internal.h
    (load_bundle, ws_layout, the sample-list stride) and gdb_ops.hip (the plan) — comments and blank lines removed.
    tools/profile_round.sh stamps profiles/traffic.json with the same function."""
    import hashlib
    import re
    code = []
    for name in ("gdb_fused.hip", "gdb_internal.h", "gdb_ops.hip"):
        src = open(os.path.join(ROOT, "gdb-nerf_amd", "csrc", name), encoding="utf-8").read()
        code.append("\n".join(l.rstrip() for l in re.sub(r"//[^\n]*", "", src).splitlines() if l.strip()))
    return hashlib.sha256("\n".join(code).encode()).hexdigest()[:16]


def to_dev(frame, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in frame.items()}


def fine_rgb(bf, H, W):
    """The pixel-shuffled fine RGB image of a render (network.py:170-175): (n_bundles, Q) -> (2H, 2W, 3)."""
    x = bf[:, :12].reshape(H, W, 3, 2, 2)
    return x.permute(0, 3, 1, 4, 2).reshape(2 * H, 2 * W, 3)


def psnr_delta_db(bf, bf_ref, H, W):
    """|PSNR(render, GT*) - PSNR(reference render, GT*)| with GT* = reference render + fixed noise (SURVEY.md §8(c):
    no dataset offline); PSNR as the evaluator computes it (evaluators/gdb_nerf.py:78-82): clamp to [0,1], data range 1."""
    a, b = fine_rgb(bf, H, W).double().cpu(), fine_rgb(bf_ref, H, W).double().cpu()
    g = torch.Generator().manual_seed(0)
    gt = (b + 0.03 * torch.randn(b.shape, generator=g, dtype=torch.float64)).clamp(0, 1)
    ps = lambda x: 10.0 * torch.log10(1.0 / ((gt - x.clamp(0, 1)) ** 2).mean()).item()
    return abs(ps(a) - ps(b))


def measure_peaks(dev):
    """Attainable peaks on this node: HBM stream triad over 3 x 256 MiB (beyond the 256 MiB Infinity
    Cache) and bare MFMA loops (4 independent accumulators per wave, 2 waves per SIMD, operands in registers)."""
    from gdb_nerf_amd import build as _b
    lib = C.CDLL(_b.build_peaks())
    lib.gdb_peak_triad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gdb_peak_mfma.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream(dev).cuda_stream
    n4 = (256 << 20) // 16
    a, b, c = (torch.empty(n4 * 4, device=dev).normal_() for _ in range(3))
    sink = torch.zeros(4096, device=dev)

    def best(fn, reps):
        fn(); torch.cuda.synchronize()
        t = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 1e-3)
        return min(t)
    t_triad = best(lambda: lib.gdb_peak_triad(a.data_ptr(), b.data_ptr(), c.data_ptr(), n4, st), 5)
    blocks, iters = 2048, 2000
    t32 = best(lambda: lib.gdb_peak_mfma(0, sink.data_ptr(), blocks, iters, st), 3)
    t16 = best(lambda: lib.gdb_peak_mfma(1, sink.data_ptr(), blocks, 8 * iters, st), 3)
    del a, b, c
    return {"hbm_triad_GBps": 48.0 * n4 / t_triad / 1e9,
            "mfma_f32_TFLOPs": blocks * 16 * iters * 4096 / t32 / 1e12,
            "mfma_f16_TFLOPs": blocks * 16 * 8 * iters * 32768 / t16 / 1e12,
            "how": "triad a=b+s*c on 3 x 256 MiB float4 arrays (best of 5); v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16 loops, "
                   "2048 workgroups x 4 waves x 4 accumulators (best of 3)"}


def cpu_baseline(wl, frame, weights, quick=False):
    """the pure-PyTorch CPU restatement of the hot path (oracle/gdb_oracle_torch.py: the torch CPU kernels the
    reference itself would run, pinned to the numpy oracle and through it to the reference's fixtures) on this host's cores —
    rows at 8 / 32 / 64 threads (3 frames each after one dropped; `value` = the fastest row, `cores` = its thread count) and the
    os.cpu_count() row on c1 only.  The numpy oracle (single-threaded element-wise numpy, the parity checker) is timed beside it.
    quick: the 8-thread row only (rank 0 of an N > 1 run)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gdb_oracle  # the checker, used here only as the reported CPU baseline
    import gdb_oracle_torch

    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ncpu = os.cpu_count() or 1
    Ho, Wo = wl["Ho"], wl["Wo"]
    kw = dict(max_num_samples=wl["S"], is_adaptive=wl["adaptive"])

    def timed(fn, reps=3):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return float(np.mean(t[1:]))
    before = torch.get_num_threads()
    rows = []
    # thread counts: 8, 32, 64 (what fits the host); the all-hardware-threads row is taken on c1 only — on a 256-thread host these
    # small torch CPU kernels oversubscribe and ONE c2 frame at os.cpu_count() threads took 38 s (round 3), which says nothing
    # about the host and cost the default run most of its time
    counts = [8] if quick else sorted({t for t in (8, 32, 64) if t <= max(8, ncpu)})
    for threads in counts:
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        gdb_oracle_torch.hot_path(frame, weights, **kw)          # first frame, dropped when more follow
        first = time.perf_counter() - t0
        if first <= 8.0:
            s_, note = timed(lambda: gdb_oracle_torch.hot_path(frame, weights, **kw), reps=3), "mean of 2 frames after 2 dropped"
        else:
            s_, note = first, "ONE frame (it took more than 8 s: not repeated)"
        rows.append({"impl": "torch", "threads": threads, "value": Ho * Wo / s_, "s_per_frame": s_, "frames": note})
    best = max(rows, key=lambda r: r["value"])
    # c1 (64x80, the reference's own CPU-runnable case) is always reported beside the benched workload
    c1 = WORKLOADS["c1"]
    f1 = synthetic.make_frame(c1["Ho"], c1["Wo"], V=c1["V"], scene=c1["scene"], seed=0)
    c1_rows = []
    for threads in ([8] if quick else sorted({8, ncpu})):
        torch.set_num_threads(threads)
        fn1 = lambda: gdb_oracle_torch.hot_path(f1, weights, max_num_samples=c1["S"], is_adaptive=c1["adaptive"])
        t0 = time.perf_counter(); fn1(); first = time.perf_counter() - t0
        # (all hardware threads of a 256-thread host: 13 s for ONE 64x80 frame, oversubscribed - taken once, not averaged)
        s1, note = (timed(fn1), "mean of 2 frames after 2 dropped") if first <= 2.0 else (first, "ONE frame (it took more than 2 s: not repeated)")
        c1_rows.append({"workload": "c1 64x80", "impl": "torch", "threads": threads, "value": c1["Ho"] * c1["Wo"] / s1, "s_per_frame": s1, "frames": note})
    torch.set_num_threads(before)
    if not quick:
        sn = timed(lambda: gdb_oracle.hot_path(frame, weights, **kw), reps=2)
        rows.append({"impl": "numpy oracle (the parity checker)", "threads": 1, "value": Ho * Wo / sn, "s_per_frame": sn})
    return {"value": best["value"], "unit": "rays/s", "cores": best["threads"], "kind": "port", "cpu_model": model, "host_cores": ncpu,
            "rows": rows, "c1": c1_rows[0], "c1_rows": c1_rows,
            "sample": f"full frames of {Ho}x{Wo} through the pure-PyTorch fp32 restatement of the hot path at torch.set_num_threads("
                      f"{', '.join(str(c) for c in counts)}) (per row: see `frames`); value = the fastest row, cores = its thread count; "
                      f"the os.cpu_count() = {ncpu} row on c1 only (c1_rows); numpy oracle: 2 frames, first dropped"}


def b4_record(dev, weights_np, fused_ns_per_ray, steps=100):
    """What `nerf.bundle_size: 4` costs (configs/dtu_pretrain.yaml:33 "4 for 4*4", networks/gdb_nerf/network.py:31-34): the same 512x640 /
    3-view / S_max 3 adaptive frame shape as 128 x 160 bundles of 16 rays, through whatever path HotPathEngine takes for b = 4
    (gdb_render_info: the fused entries when they accept the config, else the operator-mirror chain), per step and per RAY against
    the b = 2 fused step of this run.  tools/bench_b4.py has the per-operator split."""
    Ho, Wo = 512, 640
    fr = to_dev(synthetic.make_frame(Ho, Wo, V=3, bundle_size=4, scene="dtu", seed=0), dev)
    e = HotPathEngine(bundle_size=4, max_num_samples=3, is_adaptive=True, device=dev)
    e.load_weights(weights_np); e.reuse_outputs = True
    e.prepare(fr)
    fused = bool(e.render_info()["fused"])
    fn = (lambda: (e.prepare(fr), e.render_packed())) if fused else (lambda: (e.prepare(fr), e.render_unfused_packed()))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"path": "fused (gdb_render_bundles_packed)" if fused else "operator mirrors (gdb_sample -> gdb_encode -> gdb_mlp -> gdb_composite)",
            "bundles": e.n_bundles, "rays": Ho * Wo, "ms_per_step": ms, "ns_per_ray": ms * 1e6 / (Ho * Wo), "steps": steps, "precision": "f32",
            "per_ray_vs_b2_fused_step": (ms * 1e6 / (Ho * Wo)) / fused_ns_per_ray if fused_ns_per_ray else None}


def frame_time_ms(dev, precision="f32"):
    """t_frame: whole Network.forward (CNNs on PyTorch-ROCm/MIOpen + the HIP hot path) on DTU eval 512x640, 3 views, random
    init; the reference's protocol (run.py:56-73): synchronise, wall clock, drop the first iterations, mean."""
    from gdb_nerf_amd.configs import make_cfg
    from gdb_nerf_amd.networks import make_network
    fr = synthetic.make_frame(512, 640, V=3, seed=0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    batch = {"src_views": {"rgb": t(fr["src_images"]), "extrinsics": t(fr["src_exts"]), "intrinsics": t(fr["src_ints"])},
             "tar_views": {"extrinsics": t(fr["tar_ext"]), "intrinsics": t(fr["tar_int"])}, "near_far": t(fr["near_far"])}
    torch.manual_seed(0)
    net = make_network(make_cfg("configs/dtu_eval.yaml", ["nerf.precision", precision, "nerf.reuse_outputs", "True"])).eval().to(dev)
    times = []
    with torch.no_grad():
        for _ in range(14):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net(batch)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    return 1e3 * float(np.mean(times[3:]))


class Timed:
    """K steps of `fn(False)` bracketed by barrier + synchronize, max over ranks.  The timed region records no events; the dominant
    kernel's duration is sampled in a pass of its own (`sample`: every step carries an event pair around the kernel / the
    collective), so that `kernel_ms` is a mean over >= 20 launches whatever K is."""

    def __init__(self, dist, dev, rehearse, cuda=True):
        self.dist, self.dev, self.rehearse, self.cuda = dist, dev, rehearse, cuda

    def sync(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.cuda:
            torch.cuda.synchronize()

    def rewarm(self, fn, ms):
        """Untimed steps for `ms` of wall time: the clocks this workload holds have settled before anything is measured."""
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(20):
                fn(False)
            torch.cuda.synchronize()

    def run(self, fn, warmup, steps):
        for _ in range(warmup):
            fn(False)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(False)
        self.sync()
        dt = time.perf_counter() - t0
        if self.dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.rehearse else self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def sample(self, fn, n=40):
        """n untimed steps with event pairs (fn(True) appends them to the lists it was given)."""
        for _ in range(n):
            fn(True)
        self.sync()


def self_launch(n):
    """Run this very command line under torch.distributed.run with n ranks on this node (one per GPU) as a child process; relay
    what the ranks print on stdout (rank 0's one JSON line) and the launcher's exit code.  The reference's only process-group
    bootstrap is the DDP launch of train_net.py:106-111 (`torch.distributed.launch`); the driver's N > 1 form is the same thing."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout.splitlines():   # stdout carries rank 0's line and nothing else (gloo / RCCL chatter of the ranks goes to stderr)
        print(line, file=sys.stdout if line.startswith('{"metric"') else sys.stderr, flush=True)
    if p.returncode != 0:
        raise SystemExit(p.returncode)


def plumbing_only(args, world, rank):
    """GDB_BENCH_REHEARSE=1 on a host without any GPU: everything of the N > 1 run except the HIP calls — gloo rendezvous from the
    launcher's environment, the barrier + max-over-ranks timing protocol, the strip all-gather (StripGather on CPU tensors, a
    stand-in renderer that writes exactly the rows it is asked for), the per-rank gather of the report fields, rank 0's one line."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    try:
        wl = WORKLOADS["c1"]
        H, W, C = wl["Ho"] // 2, wl["Wo"] // 2, 41
        truth = torch.arange(H * W * C, dtype=torch.float32).view(H, W * C)
        gather = StripGather(H, W, C, world, rank, "cpu", dist)
        r0, r1 = gather.strip
        timed = Timed(dist, torch.device("cpu"), True, cuda=False)

        def step(sample):
            gather.full.view(H, W * C)[r0:r1] = truth[r0:r1]
            gather.gather()
        dt = timed.run(step, args.warmup, args.steps)
        ok = bool(torch.equal(gather.full.view(H, W * C), truth))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "rows": [r0, r1], "prepare_ms": None, "kernel_ms": None, "allgather_ms": None, "bus_GBps": None})
        if rank == 0:
            print(json.dumps({"metric": "rendered rays/sec, GDB-NeRF hot path (sample+fetch+MLP+composite)", "value": None, "unit": "rays/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(1, args.steps) * 1e3,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "none",
                              "config": {"workload": "c1 (stand-in renderer)", "shard": args.shard},
                              "mode": "rows: one frame, row strips, strip-only plan, packed output, one all-gather",
                              "collective": "gloo (rehearsal, CPU tensors)", "allgather_bytes_per_rank": gather.nbytes,
                              "plumbing_only": True, "world_size": world, "gathered_equals_full_render": ok, "per_rank": per_rank,
                              "note": "no GPU on this host: launch, rendezvous, timing protocol, strip all-gather and report "
                                      "plumbing rehearsed over gloo with a stand-in renderer; not a measurement"}), flush=True)
    finally:
        dist.destroy_process_group()


def ev_ms(pairs):
    return float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else None


def hot_ms(timed, render_fn, n=200):
    """GPU time of the render's launches (the dominant kernel, + the flat schedule's 5 us fix-up launch where that schedule runs): n
    back-to-back render calls on the frame last prepared, between ONE pair of HIP events on the launch stream.  The calls queue up
    behind each other (a render is ~100 us of GPU time, its host side ~10 us), so the figure is GPU time including the launch gaps
    a render really has.  Its inputs are cache-warm, unlike the timed region's (frames cycled through the HBM ring): rocprofv3's
    average over the timed region reads 2-3 % above it (profiles/).  (An event pair around every render of the step loop instead
    leaves event packets between the launches: with two launches per render it read 110 us where rocprofv3 has 92.9 + 5.1; the
    difference of a step loop and a prepare-only loop under-reads, because a prepare-only step is host-bound.)"""
    for _ in range(20):
        render_fn()
    timed.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        render_fn()
    e1.record()
    timed.sync()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)   # one step is ~0.1 ms: 2000 steps = a 0.2 s timed region
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps run for this long before the W warm-up steps, so clocks have ramped (0 = off)")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--smax", type=int, default=0, help="override the workload's S_max (SURVEY §8(a): sweep 3, 6, 8; 0 = the workload's own)")
    ap.add_argument("--sampling", default="config", choices=["config", "adaptive", "fixed"], help="override the workload's sampling mode")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f32x"], help="arithmetic of the NeRF MLP in the fused kernel")
    ap.add_argument("--schedule", type=int, default=0, choices=[0, 1, 2, 3, 4], help="GDB_SCHED_*: 0 auto, 1 slot waves, 2 segment wave, 3 dense, 4 flat")
    ap.add_argument("--path", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--shard", default="rows", choices=["rows", "frames"], help="N > 1: which mode is the headline (both are timed); rows = north_star's partitioning")
    ap.add_argument("--rows-extra", default="c4,c5", help="N > 1: workloads whose rows-mode record rides in the line beside the headline ('' = none)")
    ap.add_argument("--frame-ring", type=int, default=0,
                    help="number of distinct device copies of the input frame the steps cycle through (0 = as many as it takes to "
                         "exceed the 256 MiB Infinity Cache, at most 8): consecutive steps then read their inputs from HBM, not from cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary-precision record, PSNR, measured peaks and t_frame")
    ap.add_argument("--sources-ready-record", action="store_true",
                    help="with --no-extras: still time the `prepare_sources_ready` record (a sweep of target views over fixed source views)")
    ap.add_argument("--streams", type=int, default=1,
                    help="fused path, N = 1 only: render consecutive (independent) frames on this many HIP streams with their own "
                         "workspaces, so one frame's fill/drain overlaps the next; per-launch durations then overlap too (default 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver's N = 1 command form has it: start the N ranks ourselves.  This process has
        # not touched the GPU (no torch.cuda call above) and never will: the ranks are fresh children of torch.distributed.run,
        # rank 0's JSON line and the launcher's exit code are relayed.
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # GDB_BENCH_REHEARSE=1: several ranks share the visible GPU(s) over gloo (collective staged through host memory) — a
    # logic rehearsal of the N > 1 path on a one-GPU box, never a measurement.
    rehearse = os.environ.get("GDB_BENCH_REHEARSE") == "1"
    if rehearse and world > 1 and not torch.cuda.is_available():
        # no GPU at all (the authoring container, the CPU test suite): rehearse the launch / rendezvous / timing / gather / report
        # plumbing alone, with a stand-in for the render.  Says so in the line; carries no measurement.
        return plumbing_only(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
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
    if args.streams > 1 and (args.path != "fused" or world > 1):
        raise SystemExit("--streams > 1 is for the fused path at N = 1")

    wl = dict(WORKLOADS[args.workload])
    if args.smax or args.sampling != "config":   # S_max sweep: a variant of the workload, named as such in config.workload
        wl["S"] = args.smax or wl["S"]
        wl["adaptive"] = wl["adaptive"] if args.sampling == "config" else args.sampling == "adaptive"
        wl["desc"] += f" [override: S_max {wl['S']} {'adaptive' if wl['adaptive'] else 'fixed'}]"
    Ho, Wo, V = wl["Ho"], wl["Wo"], wl["V"]
    H, W = Ho // 2, Wo // 2
    weights_np = synthetic.make_nerf_weights(seed=0)
    prec = PREC[args.precision]

    def make_engine(frame):
        e = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
        e.set_schedule(args.schedule)
        e.precision = prec
        e.load_weights(weights_np)
        e.prepare(frame)
        return e

    frame_np = synthetic.make_frame(Ho, Wo, V=V, scene=wl["scene"], seed=0)  # rows mode: every rank holds the same frame
    frame = to_dev(frame_np, dev)
    # The steps cycle through a ring of distinct device copies of the frame whose total size exceeds the 256 MiB Infinity Cache
    # (c2: 52.8 MB of inputs per frame -> 7 copies), so that a step's inputs come from HBM as they would in an evaluation sweep;
    # with one resident frame the whole working set would sit in L2 / Infinity Cache and the HBM fraction be cache-side.
    in_bytes = sum(v.numel() * v.element_size() for v in frame.values())
    nring = args.frame_ring if args.frame_ring > 0 else max(1, min(8, -(-(320 << 20) // in_bytes)))
    ring = [frame] + [{k: v.clone() for k, v in frame.items()} for _ in range(nring - 1)]
    ring_i = [0]

    def next_frame():
        ring_i[0] = (ring_i[0] + 1) % len(ring)
        return ring[ring_i[0]]
    timed = Timed(dist, dev, rehearse)
    kern_pairs, ag_pairs = [], []

    # ---- N = 1 (and each rank's whole-frame step) ------------------------------------------------------------------
    lanes = []  # one (engine, output buffers, stream) per stream; stream 0 is the current stream
    for i in range(max(1, args.streams)):
        e = make_engine(frame)
        nb = e.n_bundles
        o = (torch.zeros((nb, e.Q), device=dev), torch.zeros((nb,), device=dev), torch.zeros((nb,), device=dev))
        lanes.append((e, o, torch.cuda.current_stream(dev) if i == 0 else torch.cuda.Stream(dev)))
    torch.cuda.synchronize()
    eng, out, _ = lanes[0]
    nb = eng.n_bundles
    counter = [0]

    def events():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step_frame(sample, precision=None, pairs=kern_pairs, eng=eng, out=out):
        """prepare + hot path on this rank's whole frame."""
        if args.streams > 1:  # frame i on stream i % n: prepare + render back to back on that stream
            e, o, st = lanes[counter[0] % len(lanes)]
            counter[0] += 1
            with torch.cuda.stream(st):
                e.prepare(next_frame())
                if sample:
                    e0, e1 = events(); e0.record()
                e.render(0, H, precision, o)
                if sample:
                    e1.record(); pairs.append((e0, e1))
            return
        eng.prepare(next_frame())
        if args.path == "fused":
            if sample:
                e0, e1 = events(); e0.record()
            eng.render(0, H, precision, out)
            if sample:
                e1.record(); pairs.append((e0, e1))
        else:
            s = eng.sample()
            rfd, vox = eng.encode(s["rays_xyz"], s["uvd"], s["ball_radii"], s["samples_per_batch"], s["total"])
            if sample:  # dominant kernel of the unfused chain is the fp32 MLP
                e0, e1 = events(); e0.record()
            sigma, feat = eng.mlp(vox, rfd, s["total"])
            if sample:
                e1.record(); pairs.append((e0, e1))
            eng.composite(sigma, feat, s["z_vals"], s["indices"], nb, s["total"])

    # The render's GPU time on HBM-cold inputs: one engine (= one workspace: camera block, pyramid, plan) per ring copy of the frame, each
    # prepared ONCE on its copy, then n render calls cycling through them between one HIP event pair - the kernel the timed region runs,
    # on the inputs it has there (never the frame of the previous launch), without the prepare launch between two renders.  This is the
    # `kernel_ms` of the roofline; rocprofv3's average for the same kernel over the timed region (profiles/) must agree with it.
    ring_engs = [eng]

    def ring_kernel_ms(precision=None, mk=None, out_=None, n=200):
        mk = mk or make_engine
        if len(ring_engs) < len(ring):
            ring_engs.extend(mk(ring[i]) for i in range(len(ring_engs), len(ring)))
        for i, e in enumerate(ring_engs):
            e.precision = eng.precision if precision is None else precision
            e.prepare(ring[i])
        o = out if out_ is None else out_
        k = [0]

        def one():
            ring_engs[k[0] % len(ring_engs)].render(0, H, precision, o)
            k[0] += 1
        ms = hot_ms(timed, one, n)
        for e in ring_engs:
            e.precision = prec
        return ms

    # ---- N > 1, rows mode: one frame, row strips, packed output, one all-gather --------------------------------------
    def rows_record(wl_name, wl_, frame_np_, steps, warmup, eng_=None, pname_=None):
        """north_star's partitioning of ONE frame of workload `wl_name` over the ranks: contiguous bundle-map row strips; every rank
        prepares (whole pyramid, strip-only plan), renders its strip straight into the packed buffer and ONE all-gather leaves all rows
        on every rank.  Returns the record (rank-0 view + every rank's prepare / kernel / all-gather times)."""
        Ho_, Wo_ = wl_["Ho"], wl_["Wo"]
        H_, W_ = Ho_ // 2, Wo_ // 2
        fr = to_dev(frame_np_, dev)
        nbytes = sum(v.numel() * v.element_size() for v in fr.values())
        nr = args.frame_ring if args.frame_ring > 0 else max(1, min(8, -(-(320 << 20) // nbytes)))
        rg = [fr] + [{k: v.clone() for k, v in fr.items()} for _ in range(nr - 1)]
        e = eng_
        pname_ = pname_ or args.precision   # (c5 rides along at f16: BASELINE.json configs[4] names the fp16 MFMA path)
        if e is None:
            e = HotPathEngine(max_num_samples=wl_["S"], is_adaptive=wl_["adaptive"], device=dev)
            e.set_schedule(args.schedule); e.precision = PREC[pname_]; e.load_weights(weights_np)
        g = StripGather(H_, W_, e.Q + 2, world, rank, dev, dist, stage_cpu=rehearse)
        r0_, r1_ = g.strip
        ri = [0]
        pp, kp, ap_ = [], [], []

        def step(sample):
            ri[0] = (ri[0] + 1) % len(rg)
            if sample:
                p0, p1 = events(); p0.record()
            e.prepare(rg[ri[0]], rows=(r0_, r1_))
            if sample:
                p1.record(); pp.append((p0, p1))
                k0, k1 = events(); k0.record()
            e.render_packed(r0_, r1_, None, g.full)
            if sample:
                k1.record(); kp.append((k0, k1))
                a0, a1 = events(); a0.record()
            g.gather()
            if sample:
                a1.record(); ap_.append((a0, a1))
        timed.rewarm(step, min(args.prewarm_ms, 150.0))
        dt_ = timed.run(step, warmup, steps)
        timed.sample(step)
        # untimed check: the gathered strips equal this rank's own render of the whole frame, bit for bit
        e.prepare(rg[ri[0]])
        ok = bool(torch.equal(g.full, e.render_packed(0, H_)))
        n_s = int(e.sample()["total"].item())
        mine = {"rank": rank, "rows": [r0_, r1_], "prepare_ms": ev_ms(pp), "kernel_ms": ev_ms(kp), "allgather_ms": ev_ms(ap_),
                "bus_GBps": (g.nbytes / (ev_ms(ap_) * 1e-3) / 1e9) if ap_ and ev_ms(ap_) else None, "gathered_equals_full_render": ok}
        per = [None] * world
        dist.all_gather_object(per, mine)
        share_ = (r1_ - r0_) / H_
        af_ = alg_flops(n_s, wl_["V"]) * share_
        rec = {"workload": wl_name, "mode": "rows: one frame, row strips, strip-only plan, packed output, one all-gather", "scaling": "strong",
               "value": Ho_ * Wo_ * steps / dt_, "unit": "rays/s", "ms_per_step": dt_ / steps * 1e3, "steps": steps, "rays_per_step": Ho_ * Wo_,
               "prepare_ms": mine["prepare_ms"], "kernel_ms": mine["kernel_ms"], "allgather_ms": mine["allgather_ms"],
               "allgather_bytes_per_rank": g.nbytes, "bus_GBps": mine["bus_GBps"],
               "collective": "gloo (rehearsal, staged through host memory)" if rehearse else "RCCL all_gather_into_tensor, in place" if g.even else "RCCL all_gather_into_tensor, padded",
               "world_size": world, "gathered_equals_full_render": all(p["gathered_equals_full_render"] for p in per),
               "kernel": e.render_info(None, r0_, r1_)["kernel"], "strip_mfma_frac": (af_ / (mine["kernel_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[pname_]) if mine["kernel_ms"] else None,
               "precision": pname_, "dtype": DTYPE[pname_],
               "n_samples": n_s, "frame_ring": nr, "per_rank": per}
        del g, rg, fr
        return rec, e, (r0_, r1_)

    # clock ramp: a fresh process starts at idle clocks and a step is ~0.1 ms, so W warm-up steps alone can end before
    # the GPU reaches its sustained clock; run untimed steps for a fixed wall time first (not part of W or K)
    timed.rewarm(step_frame, args.prewarm_ms)

    rows_mode = world > 1 and args.shard == "rows"
    extra = {}
    kern_fused_ms = kern_warm_ms = None
    r0, r1 = 0, H
    if world == 1:
        dt = timed.run(step_frame, args.warmup, args.steps)
        if args.path == "fused" and args.streams == 1:
            kern_warm_ms = hot_ms(timed, lambda: eng.render(0, H, None, out))
            kern_fused_ms = ring_kernel_ms() if len(ring) > 1 else kern_warm_ms
            eng.prepare(frame)
        else:
            timed.sample(step_frame)
        rays_per_step, share = Ho * Wo, 1.0
    else:
        # both modes are timed; --shard picks the headline
        rec_rows, _, (r0, r1) = rows_record(args.workload, wl, frame_np, args.steps, args.warmup, eng)
        # frames mode: every rank its own frame (seeded by rank)
        frame = to_dev(synthetic.make_frame(Ho, Wo, V=V, scene=wl["scene"], seed=rank), dev)
        ring[:] = [frame] + [{k: v.clone() for k, v in frame.items()} for _ in range(nring - 1)]
        kern_pairs.clear()
        dt_frames = timed.run(step_frame, args.warmup, args.steps)
        timed.sample(step_frame)
        kern_frames = ev_ms(kern_pairs)
        rec_frames = {"mode": "frames: every rank its own frame, no data-path collective", "scaling": "weak",
                      "value": world * Ho * Wo * args.steps / dt_frames, "ms_per_step": dt_frames / args.steps * 1e3, "kernel_ms": kern_frames}
        # the same rows protocol on the workloads SURVEY 8(e) names as the meaningful multi-GPU points (shorter regions)
        for name in [x for x in args.rows_extra.split(",") if x and x != args.workload and not (args.smax or args.sampling != "config")]:
            try:
                w2 = dict(WORKLOADS[name])
                k2 = max(20, min(args.steps, 200 if name != "c5" else 60))
                extra["rows_" + name] = rows_record(name, w2, synthetic.make_frame(w2["Ho"], w2["Wo"], V=w2["V"], scene=w2["scene"], seed=0), k2, min(args.warmup, 20),
                                                    pname_="f16" if name == "c5" else None)[0]
            except Exception as ex:  # extras never take the headline down
                extra["rows_" + name] = {"error": repr(ex)}
        if rows_mode:
            dt, rays_per_step, share = rec_rows["ms_per_step"] * 1e-3 * args.steps, Ho * Wo, (r1 - r0) / H
            extra.update({k: rec_rows[k] for k in ("prepare_ms", "allgather_ms", "allgather_bytes_per_rank", "bus_GBps", "collective", "world_size",
                                                   "gathered_equals_full_render", "mode")})
            extra["rows_per_rank"] = rec_rows["per_rank"]
            extra["independent_frames"] = rec_frames
            # both figures under names that do not depend on which mode is the headline (a stable key across rounds)
            extra["value_strong_rows"] = rec_rows["value"]; extra["value_weak_frames"] = rec_frames["value"]
            kern_ms_override = rec_rows["kernel_ms"]
        else:
            dt, rays_per_step, share = dt_frames, world * Ho * Wo, 1.0
            extra["single_frame_rows"] = rec_rows
            extra["value_strong_rows"] = rec_rows["value"]; extra["value_weak_frames"] = rec_frames["value"]
            extra.update({"world_size": world, "gathered_equals_full_render": rec_rows["gathered_equals_full_render"]})
            kern_ms_override = kern_frames
        eng.prepare(frame)

    kern_ms = (kern_fused_ms if kern_fused_ms is not None else ev_ms(kern_pairs)) if world == 1 else kern_ms_override
    ms_per_step = dt / args.steps * 1e3
    value = rays_per_step * args.steps / dt

    # ---- roofline of the dominant kernel (per launch, this rank) -------------------------------------------------------
    n_samples = int(eng.sample()["total"].item())  # actual adaptive sample count of the frame (operator mirror, untimed)
    ab = alg_bytes(Ho, Wo, V) * share
    af = alg_flops(n_samples, V) * share
    if args.path == "fused":
        kname = eng.render_info(prec, r0, r1)["kernel"]   # the library's own answer (gdb_render_info): no copy of GDB_SCHED_AUTO's rule here
    else:
        kname = "k_mlp"
        ab = 4.0 * n_samples * (V * eng.P + 8 + 1 + eng.Q) + 4 * 11930  # what that kernel must read + write
    pname = args.precision if args.path == "fused" else "f32"
    hbm_gbs = ab / (kern_ms * 1e-3) / 1e9
    mfma_tf = af / (kern_ms * 1e-3) / 1e12
    hbm_frac, mfma_frac = hbm_gbs / HBM_PEAK_GBS, mfma_tf / MFMA_PEAK_TFLOPS[pname]
    traffic, tsrc = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        # PMC numbers can only be collected under rocprofv3 (tools/profile_round.sh), so they are static between profile runs: the
        # file records the sha256 of the kernel source it was measured on, and a number measured on another source is refused
        try:
            tj = json.load(open(tpath))
            src_sha = kernel_source_sha16()
            if tj.get("_kernel_source_sha256_16") == src_sha:
                tkey = f"{args.workload}:{kname}:{pname}"
                traffic = tj.get(tkey)
                # the file a key's counters were read from (per key since round 6: one generic string named files of OTHER workloads)
                tsrc = (tj.get("_sources", {}).get(tkey) or tj.get("_source")) if traffic is not None else f"profiles/traffic.json holds no entry for {tkey}"
            else:
                tsrc = f"profiles/traffic.json was measured on other kernel sources ({tj.get('_kernel_source_sha256_16')} != {src_sha}): refused"
        except Exception:
            traffic = None
    # Which roofline bounds the kernel: the larger of the two floors.  At fp32 the MLP's algorithmic flops
    # against the 157 TFLOP/s fp32 matrix peak is the longer floor (c2: 74 us vs 9 us of HBM time); with f16 operands
    # (2.5 PFLOP/s) the HBM floor is.
    if args.path == "fused" and mfma_frac >= hbm_frac:
        roof = {"bound": "mfma", "kernel": kname, "achieved": mfma_tf, "peak": MFMA_PEAK_TFLOPS[pname], "unit": "TFLOP/s", "frac": mfma_frac}
    else:
        roof = {"bound": "hbm", "kernel": kname, "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
    roof.update({"traffic": traffic, "traffic_source": tsrc, "kernel_ms": kern_ms, "alg_bytes": ab, "alg_flops": af, "n_samples": n_samples,
                 "kernel_ms_cache_warm": kern_warm_ms,
                 "kernel_samples": 200 if kern_fused_ms is not None else (len(kern_pairs) if world == 1 else 40),
                 "kernel_ms_how": "HBM ring" if (kern_fused_ms is not None and len(ring) > 1) else "one resident frame" if kern_fused_ms is not None else "event pairs on sampled steps",
                 "note": "achieved = ALGORITHMIC bytes / flops (SURVEY.md §8(d)) per frame / kernel_ms; kernel_ms = GPU time of the render's ONE "
                         "launch on HBM-cold inputs: one engine (workspace) per ring copy of the frame, each prepared once, then 200 render calls "
                         "cycling through them between one HIP event pair on the launch stream, untimed, right after the timed region - the "
                         "kernel and the inputs of the timed region without the prepare launch in between; rocprofv3's average for the same "
                         "kernel over the timed region is committed under profiles/ and must agree.  kernel_ms_cache_warm = the same on one "
                         "resident frame.  N > 1: event pairs around the render on 40 sampled steps"})

    res = {
        "metric": "rendered rays/sec, GDB-NeRF hot path (sample+fetch+MLP+composite)", "value": value, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        # `scaling` names how the work of THIS command grows with --gpus (the driver compares the N = 1, 2, 4, 8 lines of one command): with
        # the default --shard rows ONE frame is split over the ranks (total work fixed: "strong"), with --shard frames every rank renders its
        # own frame ("weak").  The N = 1 line carries the same word as the N > 1 lines of the same command (it said
        # "weak" at N = 1 and "strong" at N > 1).
        "higher_is_better": True, "scaling": "strong" if args.shard == "rows" else "weak",
        "vs_baseline": None, "dtype": DTYPE[pname], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "bundle_size": 2, "rays_per_step": rays_per_step,
                   "path": args.path, "precision": pname, "schedule": args.schedule, "shard": args.shard if world > 1 else "none",
                   "prewarm_ms": args.prewarm_ms, "streams": args.streams,
                   "frame_ring": nring, "input_MB_per_frame": round(in_bytes / 1e6, 1),
                   "timed_region": f"{args.steps} steps over a ring of {nring} distinct HBM copies of the frame "
                                   f"({nring * in_bytes / 1e6:.0f} MB > the 256 MiB Infinity Cache): each copy is read {args.steps / nring:.1f} times, "
                                   f"never twice in a row"},
        "t_hot_ms": ms_per_step, "hbm_frac": hbm_frac, "mfma_frac": mfma_frac,
        "roofline": roof,
    }
    res.update(extra)

    def sources_ready_record(pname_, k2_):
        """A sweep of target views over FIXED source views (run.py renders a camera path from one set of source views): the products of
        k_prepare that depend on the sources alone - the feature pyramid(s), the half-precision image copy - are in the workspace from
        the sweep's first frame, and every further step rebuilds the camera block and the plan only (GDB_PREP_SOURCES_READY, ABI v7).
        One engine (workspace) per ring copy of the frame, each prepared in full ONCE, untimed; a step = prepare(sources_unchanged) +
        render on the next copy.  A second record beside the headline, NEVER the headline: the headline's step prepares in full."""
        if len(ring_engs) < len(ring):
            ring_engs.extend(make_engine(ring[i]) for i in range(len(ring_engs), len(ring)))
        for i_, e_ in enumerate(ring_engs):
            e_.precision = PREC[pname_]; e_.prepare(ring[i_])
        kk, pp_ = [0], []

        def step(sample):
            kk[0] = (kk[0] + 1) % len(ring_engs)
            e_ = ring_engs[kk[0]]
            if sample:
                p0, p1 = events(); p0.record()
            e_.prepare(ring[kk[0]], sources_unchanged=True)
            if sample:
                p1.record(); pp_.append((p0, p1))
            e_.render(0, H, PREC[pname_], out)
        timed.rewarm(step, 100.0)
        dt_ = timed.run(step, 50, k2_)
        timed.sample(step)
        for e_ in ring_engs:
            e_.precision = prec
        return {"precision": pname_, "value": Ho * Wo * k2_ / dt_, "ms_per_step": dt_ / k2_ * 1e3, "steps": k2_, "prepare_ms": ev_ms(pp_),
                "vs_headline": (Ho * Wo * k2_ / dt_) / value if pname_ == args.precision else None,
                "note": "step = gdb_prepare_ex(GDB_PREP_SOURCES_READY) + render: camera block and plan rebuilt, pyramids / image copy of the "
                        "fixed source views kept; never the headline (its step prepares in full)"}

    if rank == 0 and world == 1 and args.no_extras and args.sources_ready_record and args.path == "fused" and args.streams == 1:
        res["prepare_sources_ready"] = sources_ready_record(args.precision, max(100, min(args.steps, 1000)))
        eng.prepare(frame)

    if rank == 0 and world == 1 and not args.no_extras and args.path == "fused" and args.streams == 1:
        # Parity in the line itself, against the CPU ORACLE (oracle/gdb_oracle.py, the checker of tests/): the c1-size frame (64x80, BASELINE
        # configs[0]: a frame the oracle finishes in a second) rendered by this build at the headline precision - PSNR delta of the fine
        # RGB (north_star: within 0.05 dB) and max |bundle_feat - oracle|.  The benched frame itself is compared with the library's
        # exact-fp32 operator chain (HIP against HIP: named so; the oracle comparison at c2 .. c5 size lives in tests/test_hip_parity.py).
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import gdb_oracle   # the checker
            c1 = WORKLOADS["c1"]
            f1 = synthetic.make_frame(c1["Ho"], c1["Wo"], V=c1["V"], scene=c1["scene"], seed=0)
            with np.errstate(all="ignore"):
                obf1 = gdb_oracle.hot_path(f1, weights_np, max_num_samples=c1["S"], is_adaptive=c1["adaptive"])[0]
            e1 = HotPathEngine(max_num_samples=c1["S"], is_adaptive=c1["adaptive"], device=dev)
            e1.precision = prec; e1.load_weights(weights_np); e1.prepare(to_dev(f1, dev))
            bf1 = e1.render(0, c1["Ho"] // 2, prec)[0]
            res["psnr_delta_db"] = psnr_delta_db(bf1, torch.from_numpy(obf1).to(dev), c1["Ho"] // 2, c1["Wo"] // 2)
            res["max_abs_err_vs_oracle"] = float((bf1.cpu() - torch.from_numpy(obf1)).abs().max())
            res["parity_frame"] = "c1 64x80 against oracle/gdb_oracle.py (the CPU checker), precision " + pname
            del e1
        except Exception as ex:   # extras never take the headline down
            res["parity_error"] = repr(ex)
        ubf = eng.render_unfused()[0]
        eng.prepare(frame)
        res["psnr_delta_db_vs_hip_fp32_chain"] = psnr_delta_db(eng.render(0, H, prec)[0], ubf, H, W)
        res["max_abs_err_vs_hip_fp32_chain"] = float((eng.render(0, H, prec)[0] - ubf).abs().max())
        # the other precisions, timed the same way on the same frame (shorter region): "secondary" = f16 operands (f32 when the
        # headline is not f32), "secondary_f32x" = the split-f16 path
        # These are extras beside the headline: each gets its own re-warm (>= 100 ms of its own steps: another precision holds
        # another clock) and a fixed region of >= 300 steps whatever --steps says, so a short driver run reports steady-state values.
        k2 = max(300, min(args.steps, 1000))

        def time_other(other):
            pairs2 = []
            fn2 = lambda smp: step_frame(smp, PREC[other], pairs2)
            # the engine prepares for ITS precision (GDB_PREC_F16: the half-precision pyramid is written by k_prepare in the same
            # launch); left at the headline's f32 every f16 render first rebuilt that pyramid in a launch of its own (k_pyr16, +7 us)
            eng.precision = PREC[other]
            timed.rewarm(fn2, 100.0)
            dt2 = timed.run(fn2, 50, k2)
            km2 = ring_kernel_ms(PREC[other]) if len(ring) > 1 else hot_ms(timed, lambda: eng.render(0, H, PREC[other], out))
            eng.precision = PREC[other]; eng.prepare(frame)
            obf = eng.render(0, H, PREC[other])[0]
            eng.precision = prec
            return {"dtype": DTYPE[other], "precision": other, "value": Ho * Wo * k2 / dt2, "ms_per_step": dt2 / k2 * 1e3,
                    "steps": k2, "kernel_ms": km2, "hbm_frac": ab / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "mfma_frac": af / (km2 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[other],
                    "psnr_delta_db_vs_hip_fp32_chain": psnr_delta_db(obf, ubf, H, W), "max_abs_err_vs_hip_fp32_chain": float((obf - ubf).abs().max())}
        others = [p for p in ("f16", "f32", "f32x") if p != args.precision]
        res["secondary"] = time_other(others[0])
        if "f32x" in others[1:]:
            res["secondary_f32x"] = time_other("f32x")
        del ubf
        # north_star's "x 8 samples": S_max 8 on the c2 shape, adaptive (dense schedule) and fixed (segment wave), each with its own
        # engine, re-warm, a region of >= 300 steps and its own ACTUAL sample count (configs/dtu_eval.yaml:7 has S_max 3,
        # dtu_pretrain.yaml:35 has 6; the reference has no 8: SURVEY.md 8(a) note)
        def time_smax(smax, adaptive):
            e8 = HotPathEngine(max_num_samples=smax, is_adaptive=adaptive, device=dev)
            e8.set_schedule(args.schedule); e8.precision = prec; e8.load_weights(weights_np); e8.prepare(frame)
            o8 = tuple(torch.zeros_like(t) for t in out)
            pairs8 = []
            fn8 = lambda smp: step_frame(smp, prec, pairs8, e8, o8)
            timed.rewarm(fn8, 100.0)
            dt8 = timed.run(fn8, 50, k2)
            km8 = hot_ms(timed, lambda: e8.render(0, H, prec, o8))
            ns8 = int(e8.sample()["total"].item())
            af8 = alg_flops(ns8, V)
            return {"S_max": smax, "sampling": "adaptive" if adaptive else "fixed", "precision": args.precision,
                    "kernel": e8.render_info(prec)["kernel"],
                    "value": Ho * Wo * k2 / dt8, "ms_per_step": dt8 / k2 * 1e3, "steps": k2, "kernel_ms": km8, "n_samples": ns8,
                    "hbm_frac": ab / (km8 * 1e-3) / 1e9 / HBM_PEAK_GBS, "mfma_frac": af8 / (km8 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[pname]}
        if args.workload == "c2" and not args.smax:
            try:
                res["smax8_adaptive"] = time_smax(8, True)
                res["smax8_fixed"] = time_smax(8, False)
            except Exception as ex:  # measurement extras never take the headline down
                res["smax8_error"] = repr(ex)
        # throughput of a sweep over independent frames with two frames in flight: frame i on HIP stream i % 2 (own engine,
        # workspace and outputs), so one frame's fill / drain overlaps its neighbour's steady state.  Reported beside the headline,
        # never as it: per-launch durations overlap in this mode.
        e2 = make_engine(frame)
        o2 = tuple(torch.zeros_like(t) for t in out)
        two = [(eng, out, torch.cuda.current_stream(dev)), (e2, o2, torch.cuda.Stream(dev))]
        turn = [0]

        def step_pipe(sample):
            e, o, st = two[turn[0] % 2]
            turn[0] += 1
            with torch.cuda.stream(st):
                e.prepare(next_frame())
                e.render(0, H, prec, o)
        timed.rewarm(step_pipe, 100.0)
        dtp = timed.run(step_pipe, 50, k2)
        res["pipelined_2_streams"] = {"value": Ho * Wo * k2 / dtp, "ms_per_step": dtp / k2 * 1e3, "steps": k2, "precision": args.precision,
                                      "vs_headline": (Ho * Wo * k2 / dtp) / value}
        del e2, o2
        if args.workload == "c2" and args.precision == "f32":
            try:
                res["bundle_size_4"] = b4_record(dev, weights_np, ms_per_step * 1e6 / (Ho * Wo))
            except Exception as ex:  # measurement extras never take the headline down
                res["bundle_size_4"] = {"error": repr(ex)}
        try:
            res["prepare_sources_ready"] = sources_ready_record(args.precision, k2)
            if args.precision != "f16":
                eng.precision = PREC["f16"]
                res["prepare_sources_ready_f16"] = sources_ready_record("f16", k2)
                eng.precision = prec
            eng.prepare(frame)
        except Exception as ex:  # measurement extras never take the headline down
            res["prepare_sources_ready"] = {"error": repr(ex)}
        # The same step (prepare + render) replayed from HIP graphs, one captured per ring copy of the frame: the step enqueues on its
        # stream only - no host sync, no allocation (tests/test_hip_parity.py::test_hot_path_step_replays_from_a_hip_graph) - so a sweep
        # can replay it.  What a replay costs (round 6, tools/graph_probe.py -> profiles/r06/graph_probe_f32.json): a FIXED ~5 us of GPU
        # time per hipGraphLaunch on this ROCm, whatever the graph holds (a one-kernel graph of the render alone: 99.6 us against 94.3
        # eager; one step per graph 113.7 against 108.6; FOUR steps per graph 110.0) - so the one-step-per-graph record reads ~4 % below
        # the eager headline and `steps_per_graph_4` closes most of it; the two kernels themselves run as long either way.
        try:
            graphs = []
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                eng.prepare(ring[0]); eng.render(0, H, prec, out)
            torch.cuda.current_stream(dev).wait_stream(side)
            for fr in ring:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    eng.prepare(fr)
                    eng.render(0, H, prec, out)
                graphs.append(g)
            gi = [0]

            def step_graph(sample):
                gi[0] = (gi[0] + 1) % len(graphs)
                graphs[gi[0]].replay()
            timed.rewarm(step_graph, 100.0)
            dtg = timed.run(step_graph, 50, k2)
            res["hipgraph_replay"] = {"value": Ho * Wo * k2 / dtg, "ms_per_step": dtg / k2 * 1e3, "steps": k2, "precision": args.precision,
                                      "graphs": len(graphs), "vs_headline": (Ho * Wo * k2 / dtg) / value,
                                      "note": "one step per graph; a hipGraphLaunch costs ~5 us of GPU time whatever it holds (tools/graph_probe.py)"}
            del graphs
            graphs4 = []
            for k0 in range(len(ring)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for d in range(4):
                        eng.prepare(ring[(k0 + d) % len(ring)])
                        eng.render(0, H, prec, out)
                graphs4.append(g)
            g4 = [0]

            def step_graph4(sample):
                g4[0] = (g4[0] + 4) % len(graphs4)
                graphs4[g4[0]].replay()
            timed.rewarm(step_graph4, 100.0)
            n4 = max(25, k2 // 4)
            dtg4 = timed.run(step_graph4, 12, n4)
            res["hipgraph_replay"]["steps_per_graph_4"] = {"value": Ho * Wo * 4 * n4 / dtg4, "ms_per_step": dtg4 / (4 * n4) * 1e3, "steps": 4 * n4,
                                                           "vs_headline": (Ho * Wo * 4 * n4 / dtg4) / value}
            del graphs4
            eng.prepare(frame)
        except Exception as ex:  # measurement extras never take the headline down
            res["hipgraph_replay"] = {"error": repr(ex)}
        try:
            pk = measure_peaks(dev)
            pk["hbm_frac_of_measured"] = hbm_gbs / pk["hbm_triad_GBps"]
            pk["mfma_frac_of_measured"] = mfma_tf / (pk["mfma_f16_TFLOPs"] / 3 if pname == "f32x" else pk[f"mfma_{pname}_TFLOPs"])
            res["peaks_measured"] = pk
        except Exception as ex:  # measurement extras never take the headline down
            res["peaks_measured"] = {"error": repr(ex)}
        try:
            res["t_frame_ms"] = frame_time_ms(dev)
            res["t_frame_ms_f32x"] = frame_time_ms(dev, "f32x")   # split-f16 pairs in the fused MLP and in the decoder's convolutions
        except Exception as ex:
            res["t_frame_ms"] = None
            res["t_frame_error"] = repr(ex)
    if dist is not None:
        # every rank's dominant-kernel time and roofline fraction ride in rank 0's line (the line's `roofline` is rank 0's own)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "rows": [r0, r1], "kernel_ms": kern_ms, "roofline_frac": roof["frac"],
                                          "bound": roof["bound"], "hbm_frac": hbm_frac, "mfma_frac": mfma_frac})
        res["per_rank"] = per_rank
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline:
        # N > 1: the 8-thread row only, on rank 0, after the process group is gone (no rank waits on it)
        res["cpu_baseline"] = cpu_baseline(wl, frame_np, weights_np, quick=world > 1)
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
