#!/usr/bin/env python3
"""Why is a HIP-graph replay of the bench step (k_prepare + the render kernel) slower than the eager step?  (VERDICT r05 item 8:
hipgraph_replay 0.1139 ms against 0.1097 eager.)  Times, on the c2 frame at one precision, each for STEPS steps between one pair of
HIP events on the launch stream, after a re-warm:

  eager_ring       the bench's step: prepare(ring[i]) + render, ring of distinct HBM copies of the frame
  eager_one        the same on ONE resident frame
  graph_ring       one captured graph per ring copy, cycled (what bench.py's hipgraph_replay does)
  graph_one        ONE graph replayed back to back (the same frame every step)
  graph_ring_x4    one graph per ring copy holding FOUR consecutive steps (prepare + render of copies i .. i + 3): fewer graph launches
  render_only_*    the render alone (no prepare) eager / one graph: the gap a graph launch puts between two kernels

and the GPU-side gap per step = ms_per_step - (k_prepare + render kernel time measured back to back eager with nothing else).
usage: graph_probe.py [f32|f16] [STEPS=1000]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import WORKLOADS, PREC, to_dev
from gdb_nerf_amd import synthetic
from gdb_nerf_amd.engine import HotPathEngine

pname = sys.argv[1] if len(sys.argv) > 1 else "f32"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
wl = WORKLOADS["c2"]
H = wl["Ho"] // 2
frame = to_dev(synthetic.make_frame(wl["Ho"], wl["Wo"], V=wl["V"], scene=wl["scene"], seed=0), dev)
ring = [frame] + [{k: v.clone() for k, v in frame.items()} for _ in range(6)]
eng = HotPathEngine(max_num_samples=wl["S"], is_adaptive=wl["adaptive"], device=dev)
eng.precision = PREC[pname]; eng.load_weights(synthetic.make_nerf_weights(seed=0)); eng.prepare(frame)
nb = eng.n_bundles
out = (torch.zeros((nb, eng.Q), device=dev), torch.zeros((nb,), device=dev), torch.zeros((nb,), device=dev))


def timed(fn, steps=STEPS, warm_s=0.3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return {"gpu_ms_per_step": e0.elapsed_time(e1) / steps, "host_ms_per_step": (time.perf_counter() - w0) / steps * 1e3}


i = [0]


def eager_ring():
    i[0] = (i[0] + 1) % len(ring)
    eng.prepare(ring[i[0]]); eng.render(0, H, None, out)


def eager_one():
    eng.prepare(frame); eng.render(0, H, None, out)


def render_only():
    eng.render(0, H, None, out)


def prepare_only():
    eng.prepare(frame)


def capture(fn):
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


res = {"precision": pname, "steps": STEPS, "workload": "c2"}
res["eager_ring"] = timed(eager_ring)
res["eager_one"] = timed(eager_one)
res["render_only_eager"] = timed(render_only)
res["prepare_only_eager"] = timed(prepare_only)
eng.prepare(frame)
graphs = [capture(lambda fr=fr: (eng.prepare(fr), eng.render(0, H, None, out))) for fr in ring]
gi = [0]


def graph_ring():
    gi[0] = (gi[0] + 1) % len(graphs)
    graphs[gi[0]].replay()


res["graph_ring"] = timed(graph_ring)
res["graph_one"] = timed(lambda: graphs[0].replay())


def four(k):
    for d in range(4):
        fr = ring[(k + d) % len(ring)]
        eng.prepare(fr); eng.render(0, H, None, out)


graphs4 = [capture(lambda k=k: four(k)) for k in range(len(ring))]
g4 = [0]


def graph_ring_x4():
    g4[0] = (g4[0] + 4) % len(graphs4)
    graphs4[g4[0]].replay()


r4 = timed(graph_ring_x4, steps=STEPS // 4)
res["graph_ring_x4"] = {k: v / 4 for k, v in r4.items()}
eng.prepare(frame)
g_r = capture(render_only)
res["render_only_graph"] = timed(lambda: g_r.replay())
k_sum = res["render_only_eager"]["gpu_ms_per_step"] + res["prepare_only_eager"]["gpu_ms_per_step"]
res["note"] = ("gpu_ms_per_step = HIP events around the whole region / steps; prepare_only + render_only (each back to back with itself) = "
               f"{k_sum:.4f} ms is what the two kernels cost with no gap of another kind between them")
print(json.dumps(res, indent=1))
