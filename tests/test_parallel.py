"""Row-strip sharding of one frame over ranks and the all-gather of rendered strips
(gdb-nerf_amd/parallel.py), exercised with world_size-2/3 gloo process groups on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gdb_nerf_amd.parallel import all_strips, gather_strips, row_strip


@pytest.mark.parametrize("H,world", [(256, 8), (256, 1), (10, 4), (3, 8), (37, 5), (1, 2)])
def test_row_strips_partition_rows(H, world):
    strips = all_strips(H, world)
    assert strips[0][0] == 0 and strips[-1][1] == H
    for (a0, a1), (b0, b1) in zip(strips, strips[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in strips]
    assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
    with pytest.raises(ValueError):
        row_strip(H, world, world)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, H, W, C, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(B * H * W * C, dtype=torch.float32).view(B * H * W, C)  # what a full render would hold
        mine = torch.full_like(full, float("nan"))                                   # this rank renders only its strip
        r0, r1 = row_strip(H, rank, world)
        mine.view(B, H, W, C)[:, r0:r1] = full.view(B, H, W, C)[:, r0:r1]
        out = gather_strips(mine, H, world, dist, B=B)
        ok = bool(torch.equal(out, full))
        depth = torch.full((B * H * W,), -1.0)
        depth.view(B, H, W)[:, r0:r1] = full[:, 0].view(B, H, W)[:, r0:r1]
        ok = ok and bool(torch.equal(gather_strips(depth, H, world, dist, B=B), full[:, 0].contiguous()))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,H,W,C", [(2, 1, 8, 5, 39), (2, 2, 7, 4, 3), (3, 1, 4, 6, 2)])
def test_gather_strips_gloo(world, B, H, W, C):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, H, W, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


class _FakeEngine:
    """Stands in for HotPathEngine.render_packed on the CPU: writes the rows it is asked for, and only those."""

    def __init__(self, truth):
        self.truth = truth

    def render_packed(self, r0, r1, precision, out):
        H = self.truth.shape[0]
        out.view(H, -1)[r0:r1] = self.truth[r0:r1]
        return out


def _packed_worker(rank, world, port, H, W, C, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdb_nerf_amd.parallel import StripGather
        truth = torch.arange(H * W * C, dtype=torch.float32).view(H, W * C)
        g = StripGather(H, W, C, world, rank, "cpu", dist)
        g.full.fill_(float("nan"))
        ok = True
        for rep in range(2):  # the buffers are reused call after call, exactly as bench.py's rows mode does
            _FakeEngine(truth + rep).render_packed(*g.strip, None, g.full)
            ok = ok and bool(torch.equal(g.gather().view(H, W * C), truth + rep))
        ok = ok and g.even == (H % world == 0) and g.nbytes == (world - 1) * (-(-H // world)) * W * C * 4
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,C", [(2, 8, 5, 41), (2, 7, 4, 41), (3, 6, 3, 41), (3, 4, 6, 2)])
def test_packed_strip_gather_gloo(world, H, W, C):
    """bench.py's N > 1 step (render_packed of the rank's strip into StripGather.full, then ONE all_gather_into_tensor):
    in place when the rows divide evenly, through the padded buffer otherwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, H, W, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


class _FakeHotPath:
    """Stands in for HotPathEngine inside Network._render_packed on the CPU: writes the rows it is asked for, and only those."""

    def __init__(self, truth, B, H):
        self.truth, self.B, self.H, self.Q, self.device = truth, B, H, truth.shape[1] - 2, torch.device("cpu")
        self.calls = []

    def render_packed(self, r0=0, r1=None, precision=None, out=None):
        r1 = self.H if r1 is None else r1
        self.calls.append((r0, r1))
        if out is None:
            out = torch.zeros_like(self.truth)
        v, t = out.view(self.B, self.H, -1), self.truth.view(self.B, self.H, -1)
        v[:, r0:r1] = t[:, r0:r1]
        return out


def _network_worker(rank, world, port, B, H, W, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdb_nerf_amd.configs import make_cfg
        from gdb_nerf_amd.networks import make_network
        net = make_network(make_cfg("configs/dtu_eval.yaml", ["nerf.shard", "rows"])).eval()
        ok = net.shard == "rows"
        for rep in range(2):   # the gather buffers are cached per shape and reused frame after frame
            truth = torch.arange(B * H * W * 41, dtype=torch.float32).view(B * H * W, 41) + 1000.0 * rep
            eng = _FakeHotPath(truth, B, H)
            packed = net._render_packed(eng, B, H, W)
            ok = ok and bool(torch.equal(packed, truth)) and eng.calls == [row_strip(H, rank, world)]
        plain = make_network(make_cfg("configs/dtu_eval.yaml")).eval()   # nerf.shard defaults to none: whole frames on every rank
        eng = _FakeHotPath(truth, B, H)
        ok = ok and bool(torch.equal(plain._render_packed(eng, B, H, W), truth)) and eng.calls == [(0, H)]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,H,W", [(2, 1, 8, 5), (2, 1, 7, 4), (2, 2, 5, 3)])
def test_network_forward_shards_rows_over_ranks(world, B, H, W):
    """`nerf.shard: rows` (north_star: rays shard across the GPUs of a node): with torch.distributed initialised,
    Network.forward's hot-path section renders this rank's strip of bundle-map rows and ONE all-gather leaves the whole packed
    bundle map on every rank (even and uneven H, batch > 1); without the key every rank renders whole frames.
    Reference call site: network.py:145-169, driven once per frame by run.py:54-66."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_network_worker, args=(r, world, port, B, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
