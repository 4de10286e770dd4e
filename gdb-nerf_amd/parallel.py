"""Multi-GPU sharding of the hot path.  Bundles are independent end to end,
so the only exchange is the all-gather of rendered row strips when ONE frame is split over
the ranks.  One process per GPU; `dist` is torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).

The exchange unit is the PACKED render (`HotPathEngine.render_packed`): one (n_bundles, Q + 2)
tensor whose row is [bundle_feat | depth | opacity], so a rank's strip of rows is one contiguous
block and the whole exchange is ONE `all_gather_into_tensor` — in place when the rows divide
evenly (every BASELINE config does for 1/2/4/8 ranks), through one preallocated padded buffer
otherwise.  No zero-fill, no per-strip copies, depth and opacity included.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def row_strip(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous strip [r0, r1) of bundle-map rows owned by `rank`; sizes differ by at most one
    row, earlier ranks take the longer strips; ranks beyond H get an empty strip."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def all_strips(H: int, world: int) -> List[Tuple[int, int]]:
    return [row_strip(H, r, world) for r in range(world)]


class StripGather:
    """All-gather of the row strips of one packed render, with every buffer allocated once.

        g = StripGather(H, W, C, world, rank, device, dist)      # C = Q + 2 floats per bundle
        eng.render_packed(*g.strip, out=g.full)                  # this rank writes rows [r0, r1) of g.full
        g.gather()                                               # afterwards g.full holds all rows on every rank

    H % world == 0 (B == 1): in place — the send buffer is this rank's slice of `full`, the receive buffer is `full`
    itself (the all-gather's own in-place form: send == recv + rank * count).  Otherwise the strip goes through
    one padded (world, rows, W*C) buffer.  `nbytes` is what one rank receives per call (the bus-bandwidth numerator)."""

    def __init__(self, H: int, W: int, C: int, world: int, rank: int, device, dist, dtype=torch.float32, stage_cpu: bool = False,
                 force_padded: bool = False):
        self.H, self.W, self.C, self.world, self.rank, self.dist = H, W, C, world, rank, dist
        # stage_cpu: the collective runs on host copies (gloo cannot move device tensors: the one-GPU rehearsal of bench.py)
        self.stage_cpu = stage_cpu and torch.device(device).type != "cpu"
        self.strip = row_strip(H, rank, world)
        self.full = torch.zeros((H * W, C), dtype=dtype, device=device)
        # force_padded: take the padded form although the rows divide evenly (tests: both collective forms on one rank count)
        self.even = (world == 1 or H % world == 0) and not force_padded
        self.rows = -(-H // world)
        self._rowview = self.full.view(H, W * C)
        if not self.even:
            self._recv = torch.empty((world, self.rows, W * C), dtype=dtype, device=device)
            self._send = torch.zeros((self.rows, W * C), dtype=dtype, device=device)  # pad rows zeroed once, never re-written
        self.nbytes = (world - 1) * self.rows * W * C * self.full.element_size()

    def gather(self, always_collective: bool = False) -> torch.Tensor:
        """always_collective: run the collective at world size 1 as well (a one-rank RCCL communicator then sees exactly the call
        an N-rank job makes, in-place aliasing included: tests/test_parallel.py::test_strip_gather_through_one_rank_rccl)."""
        if self.world == 1 and not always_collective:
            return self.full
        r0, r1 = self.strip
        if self.stage_cpu:
            host = gather_strips(self.full.cpu(), self.H, self.world, self.dist)
            self.full.copy_(host)
            return self.full
        if self.even:
            self.dist.all_gather_into_tensor(self.full.view(-1), self._rowview[r0:r1].reshape(-1))
            return self.full
        self._send[: r1 - r0].copy_(self._rowview[r0:r1])
        self.dist.all_gather_into_tensor(self._recv.view(-1), self._send.view(-1))
        for r, (a, b) in enumerate(all_strips(self.H, self.world)):
            if r != self.rank and b > a:
                self._rowview[a:b].copy_(self._recv[r, : b - a])
        return self.full


def gather_strips(full: torch.Tensor, H: int, world: int, dist, B: int = 1) -> torch.Tensor:
    """General form for any per-bundle tensor `full` ((B*H*W, C) or (B*H*W,)) and batch B: on entry each rank has
    written only its own strip of every batch item; on return every rank holds all rows.  Allocates its padded
    buffers per call — the engine path uses StripGather."""
    if world == 1:
        return full
    rank = dist.get_rank()
    v = full.view(B, H, -1)  # (B, H, W*C)
    rows = -(-H // world)
    r0, r1 = row_strip(H, rank, world)
    send = torch.zeros((B, rows, v.shape[2]), dtype=full.dtype, device=full.device)
    send[:, : r1 - r0] = v[:, r0:r1]
    recv = torch.empty((world,) + tuple(send.shape), dtype=full.dtype, device=full.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    for r, (a, b) in enumerate(all_strips(H, world)):
        if r != rank and b > a:
            v[:, a:b] = recv[r][:, : b - a]
    return full
