"""Multi-GPU sharding of the hot path.  Bundles are independent end to end,
so the only exchange is the all-gather of rendered row strips when ONE frame is split over
the ranks.  One process per GPU; `dist` is torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests)."""
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


def gather_strips(full: torch.Tensor, H: int, world: int, dist, B: int = 1) -> torch.Tensor:
    """All-gather the row strips of a per-bundle tensor `full` ((B*H*W, C) or (B*H*W,)), in place:
    on entry each rank has written only its own strip; on return every rank holds all rows.
    Strips are padded to a common row count so that the collective has equal-sized shards."""
    if world == 1:
        return full
    rank = dist.get_rank()
    v = full.view(B, H, -1)  # (B, H, W*C)
    rows = -(-H // world)
    r0, r1 = row_strip(H, rank, world)
    send = torch.zeros((B, rows, v.shape[2]), dtype=full.dtype, device=full.device)
    send[:, : r1 - r0] = v[:, r0:r1]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    for r, (a, b) in enumerate(all_strips(H, world)):
        if r != rank and b > a:
            v[:, a:b] = recv[r][:, : b - a]
    return full
