"""3-D U-Nets that regularise the variance cost volume into a voxel feature volume and a depth
probability (reference networks/gdb_nerf/cost_reg_net.py).  PyTorch-ROCm; upstream of the hot path."""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import conv_block3d, deconv_block3d


def _down(cin, cout):
    return conv_block3d(cin, cout, 3, stride=2, padding=1)


def _same(cin, cout):
    return conv_block3d(cin, cout, 3, padding=1)


def _up(cin, cout):
    return deconv_block3d(cin, cout, 3, stride=2, padding=1, output_padding=1)


class _UNet3d(nn.Module):
    """`depth` down/up levels; layers are attributes conv0..conv{3*depth} as in the checkpoints."""

    def __init__(self, in_channels: int, out_channels: int, base_channels: int, depth: int) -> None:
        super().__init__()
        self._depth = depth
        c = base_channels
        self.conv0 = _same(in_channels, c)
        n = 1
        for lvl in range(depth):  # conv(2l+1): stride-2, conv(2l+2): same resolution
            setattr(self, f"conv{n}", _down(c << lvl, c << (lvl + 1)))
            setattr(self, f"conv{n + 1}", _same(c << (lvl + 1), c << (lvl + 1)))
            n += 2
        for lvl in reversed(range(depth)):
            setattr(self, f"conv{n}", _up(c << (lvl + 1), c << lvl))
            n += 1
        self.feat_head = nn.Conv3d(c, out_channels, 3, padding=1, bias=False)
        self.prob_head = nn.Conv3d(c, 1, 3, padding=1, bias=False)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(B,C,D,H,W) -> feature volume (B,out,D,H,W), depth probability (B,D,H,W) (softmax over D)."""
        skips = [self.conv0(x)]
        n = 1
        for _ in range(self._depth):
            skips.append(getattr(self, f"conv{n + 1}")(getattr(self, f"conv{n}")(skips[-1])))
            n += 2
        y = skips.pop()
        while skips:
            y = skips.pop() + getattr(self, f"conv{n}")(y)
            n += 1
        return self.feat_head(y), F.softmax(self.prob_head(y).squeeze(1), dim=1)


class CostRegNet(_UNet3d):
    def __init__(self, in_channels: int, out_channels: int, base_channels: int) -> None:
        super().__init__(in_channels, out_channels, base_channels, depth=3)


class CostRegNet_small(_UNet3d):
    def __init__(self, in_channels: int, out_channels: int, base_channels: int) -> None:
        super().__init__(in_channels, out_channels, base_channels, depth=2)
