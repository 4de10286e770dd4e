"""Three-level feature pyramid over the source images (reference networks/gdb_nerf/feature_net.py:8-64).
Stays PyTorch-ROCm / MIOpen: it is upstream of the hot path."""
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import conv_block2d


class FeatureNet(nn.Module):
    def __init__(self, base_channels: int = 8, out_channels: Sequence[int] = (32, 16, 8)) -> None:
        super().__init__()
        c = base_channels
        # encoder: full, half and quarter resolution
        self.conv0 = nn.Sequential(conv_block2d(3, c, 3, padding=1), conv_block2d(c, c, 3, padding=1))
        self.conv1 = nn.Sequential(conv_block2d(c, 2 * c, 5, stride=2, padding=2), conv_block2d(2 * c, 2 * c, 3, padding=1))
        self.conv2 = nn.Sequential(conv_block2d(2 * c, 4 * c, 5, stride=2, padding=2), conv_block2d(4 * c, 4 * c, 3, padding=1))
        # top-down path with lateral 1x1 connections
        self.out0 = nn.Conv2d(4 * c, out_channels[0], 1)
        self.inner1 = nn.Conv2d(2 * c, 4 * c, 1)
        self.inner2 = nn.Conv2d(c, 4 * c, 1)
        self.out1 = nn.Conv2d(4 * c, out_channels[1], 3, padding=1, bias=False)
        self.out2 = nn.Conv2d(4 * c, out_channels[2], 3, padding=1, bias=False)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """(N,3,H,W) -> [(N,C0,H/4,W/4), (N,C1,H/2,W/2), (N,C2,H,W)], coarsest first."""
        full = self.conv0(x)
        half = self.conv1(full)
        quarter = self.conv2(half)
        top = quarter
        pyramid = [self.out0(top)]
        for lateral, skip, head in ((self.inner1, half, self.out1), (self.inner2, full, self.out2)):
            top = F.interpolate(top, size=skip.shape[-2:], mode="nearest") + lateral(skip)
            pyramid.append(head(top))
        return pyramid
