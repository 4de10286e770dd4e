"""Residual-dense decoder that turns the 27 non-RGB bundle channels into a full-resolution colour
residual (reference networks/gdb_nerf/decoder_rdn.py).  PyTorch-ROCm; consumer of the hot path."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SEBlock2D(nn.Module):
    """Squeeze-and-excitation channel gate."""

    def __init__(self, channels: int, reduction: int = 16) -> None:
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channels, channels // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channels // reduction, channels, bias=False), nn.Sigmoid())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate = self.fc(self.avg_pool(x).flatten(1))
        return x * gate[:, :, None, None]


class ResidualDenseBlock(nn.Module):
    def __init__(self, num_feats: int, growth_rate: int = 32) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(num_feats, growth_rate, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(num_feats + growth_rate, growth_rate, 3, padding=1, bias=False)
        self.conv3 = nn.Conv2d(num_feats + 2 * growth_rate, num_feats, 3, padding=1, bias=False)
        self.se = SEBlock2D(num_feats)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a = F.relu(self.conv1(x))
        b = F.relu(self.conv2(torch.cat((x, a), 1)))
        return x + self.se(self.conv3(torch.cat((x, a, b), 1)))


class Decoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_feats: int, num_layers: int, upscale_factor: int) -> None:
        super().__init__()
        if upscale_factor <= 0 or upscale_factor & (upscale_factor - 1):
            raise ValueError('`upscale_factor` must be a power of 2.')
        self.upscale_factor = upscale_factor
        self.in_conv = nn.Conv2d(in_channels, num_feats, 3, padding=1)
        self.blocks = nn.Sequential(*(ResidualDenseBlock(num_feats) for _ in range(num_layers)))
        stages = []
        for _ in range(int(round(math.log2(upscale_factor)))):
            stages += [nn.Conv2d(num_feats, 4 * num_feats, 3, padding=1), nn.PixelShuffle(2)]
        self.up = nn.Sequential(*stages)
        self.out_conv = nn.Conv2d(num_feats, out_channels, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,in,H,W) -> (B,out,H*s,W*s)."""
        shallow = self.in_conv(x)
        return self.out_conv(self.up(shallow + self.blocks(shallow)))
