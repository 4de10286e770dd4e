"""Radiance / density MLP of the hot path (reference networks/gdb_nerf/nerf.py).  The module owns the
parameters under the reference's key names; `forward` runs `gdb_mlp` (exact fp32 HIP kernel)."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ...engine import HotPathEngine, NERF_KEYS


class NeRF(nn.Module):
    def __init__(self, hid_dim: int = 64, feat_dim: int = 16, voxel_dim: int = 8, viewdir_agg: bool = True) -> None:
        super().__init__()
        self.feat_dim, self.viewdir_agg = feat_dim, viewdir_agg
        act = lambda i, o: nn.Sequential(nn.Linear(i, o), nn.ReLU(inplace=True))
        if viewdir_agg:
            self.view_fc = act(4, feat_dim + 3)
        self.global_fc = act(3 * (feat_dim + 3), 32)
        self.agg_w_fc = act(32, 1)
        self.fc = act(32, 16)
        self.lr0 = act(voxel_dim + 16, hid_dim)
        self.sigma = nn.Sequential(nn.Linear(hid_dim, 1), nn.Softplus())
        self.weight = nn.Sequential(nn.Linear(hid_dim + voxel_dim + 16 + feat_dim + 3 + 4, hid_dim), nn.ReLU(inplace=True),
                                    nn.Linear(hid_dim, 1), nn.ReLU(inplace=True))
        self.feat_head = act(hid_dim, voxel_dim)
        self._hid, self._vox = hid_dim, voxel_dim
        self._engine: Optional[HotPathEngine] = None
        self._packed_versions = None

    def param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_engine(self, engine: HotPathEngine) -> None:
        """(Re)pack the weights into `engine` when they changed since the last pack."""
        v = self.param_versions()
        if engine.weights is None or getattr(engine, "_nerf_versions", None) != v:
            engine.load_weights({k: t.detach() for k, t in self.state_dict().items()})
            engine._nerf_versions = v

    def forward(self, vox_feat: torch.Tensor, rgbs_feat_rgb_dir: torch.Tensor, only_geo: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """vox_feat (N,C_v), rgbs_feat_rgb_dir (V,N,3b²+C_f+3+4) -> sigma (N,), feat (N,3b²+C_f+3+C_v)
        (None when only_geo, as nerf.py:104-115)."""
        P = rgbs_feat_rgb_dir.shape[-1]
        b2 = (P - self.feat_dim - 7) // 3
        b = int(round(b2 ** 0.5))
        if self._engine is None or self._engine.b != b:
            self._engine = HotPathEngine(bundle_size=b, feat_dim=self.feat_dim, voxel_dim=self._vox, hid_dim=self._hid,
                                         viewdir_agg=self.viewdir_agg, device=vox_feat.device)
        self.sync_engine(self._engine)
        sigma, feat = self._engine.mlp(vox_feat.contiguous(), rgbs_feat_rgb_dir.contiguous())
        return sigma, (None if only_geo else feat)


assert NERF_KEYS[0] == "view_fc.0"
