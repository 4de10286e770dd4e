"""Volume-integration operators of the hot path with the reference's signatures
(networks/gdb_nerf/utils.py:19-43, 88-121), backed by `gdb_composite`."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ...engine import HotPathEngine

_ENGINES = {}


def _engine(device) -> HotPathEngine:
    key = str(device)
    if key not in _ENGINES:
        _ENGINES[key] = HotPathEngine(device=device)
    return _ENGINES[key]


def weights_init(m):  # reference utils.py:8-16 (training helper, kept for API parity)
    if isinstance(m, (nn.Linear, nn.Conv2d)):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.zeros_(m.bias)


def render_weight_from_density(sigma: torch.Tensor, ray_indices: torch.Tensor, num_rays: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Normalised transmittance weights per bundle (`gdb_render_weights`).  The second value is the
    reference's `inverse_indices` (dense rank of each sample's bundle among the non-empty bundles)."""
    w = _engine(sigma.device).render_weights(sigma.contiguous(), ray_indices.contiguous(), int(num_rays))
    return w, torch.unique_consecutive(ray_indices, return_inverse=True)[1]


def accumulate_value_along_rays(feat: torch.Tensor, z_vals: torch.Tensor, weights: torch.Tensor, ray_indices: torch.Tensor,
                                num_rays: int, inverse_indices: Optional[torch.Tensor] = None):
    """Segmented sum of weights · [feat | z | 1] per bundle -> (feat_map, depth_map, opacity_map) (`gdb_accumulate`)."""
    return _engine(feat.device).accumulate(weights.contiguous(), feat.contiguous(), z_vals.contiguous(), ray_indices.contiguous(), int(num_rays))
