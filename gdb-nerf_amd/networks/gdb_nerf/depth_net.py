"""Cascade MVS depth prior: per stage a variance cost volume by homography warping, a 3-D U-Net,
soft-argmax depth and a confidence interval that becomes the next stage's search range and,
after the last stage, the per-bundle depth prior of the hot path (reference
networks/gdb_nerf/depth_net.py:118-198, 399-514).  Inference only; PyTorch-ROCm."""
from types import SimpleNamespace
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .cost_reg_net import CostRegNet, CostRegNet_small


class _AuxStageNerf(nn.Module):
    """Parameter container for the train-only per-stage NeRF of the reference (depth_net.py:201-245).
    The reference builds it unconditionally (its ctor runs in training mode, depth_net.py:40-47), so
    its tensors are in every checkpoint; they are kept here so that a reference `latest.pth` loads
    with strict=True.  Never evaluated at inference."""

    def __init__(self, hid_dim: int, voxel_dim: int, feat_dim: int, viewdir_agg: bool) -> None:
        super().__init__()
        lin = lambda i, o: nn.Sequential(nn.Linear(i, o), nn.ReLU(inplace=True))
        if viewdir_agg:
            self.view_fc = lin(4, feat_dim + 3)
        self.global_fc = lin(3 * (feat_dim + 3), 32)
        self.agg_w_fc = lin(32, 1)
        self.fc = lin(32, 16)
        self.lr0 = lin(voxel_dim + 16, hid_dim)
        self.sigma = nn.Sequential(nn.Linear(hid_dim, 1), nn.Softplus())
        self.color = nn.Sequential(nn.Linear(hid_dim + voxel_dim + 16 + feat_dim + 3 + 4, hid_dim), nn.ReLU(inplace=True),
                                   nn.Linear(hid_dim, 1), nn.ReLU(inplace=True))


def _use_hip(t: torch.Tensor, enabled: bool) -> bool:
    """CUDA tensors take the HIP kernels of the "next" rows N2 / N4 (gdb-nerf_amd/costvol.py; loud failure if the
    library is missing); CPU tensors take the PyTorch formulation below (the CNNs around them are PyTorch too)."""
    return enabled and t.is_cuda and t.dtype == torch.float32


def get_depth_values(near_far: torch.Tensor, num_depth: int, inv_depth: bool) -> torch.Tensor:
    """(B,2,H,W) near/far -> (B,num_depth,H,W) hypotheses, uniform in depth or in disparity (:399-421)."""
    lo, hi = near_far[:, :1], near_far[:, -1:]
    if inv_depth:
        lo, hi = 1.0 / lo, 1.0 / hi
    steps = torch.linspace(0.0, 1.0, num_depth, device=lo.device).view(1, num_depth, 1, 1)
    return lo + (hi - lo) * steps


def build_feature_volume(src_feat, src_exts, src_ints, tar_exts, tar_ints, depth_values, inv_depth) -> torch.Tensor:
    """Variance over source views of the features warped onto the target frustum planes (:424-476).
    src_feat (B,V,C,Hs,Ws); depth_values (B,D,Ht,Wt) -> (B,C,D,Ht,Wt)."""
    B, V, _, Hs, Ws = src_feat.shape
    D, Ht, Wt = depth_values.shape[1:]
    depth = 1.0 / depth_values if inv_depth else depth_values
    # pixel(target) -> pixel(source) homographies through the target projection's inverse
    P_src = src_ints @ src_exts[..., :3, :]
    P_tar = F.pad(tar_ints @ tar_exts[..., :3, :], (0, 0, 0, 1), value=0.0)
    P_tar[..., 3, 3] = 1.0
    Hm = (P_src @ torch.inverse(P_tar).unsqueeze(1)).view(B * V, 3, 4)
    xs, ys = torch.meshgrid(torch.arange(Wt, dtype=src_feat.dtype, device=src_feat.device) + 0.5,
                            torch.arange(Ht, dtype=src_feat.dtype, device=src_feat.device) + 0.5, indexing="xy")
    pix = torch.stack((xs, ys, torch.ones_like(xs)), 0).reshape(1, 3, Ht * Wt)
    d = depth.reshape(B, 1, D, -1).expand(-1, V, -1, -1).reshape(B * V, 1, D, -1)
    p = (Hm[..., :3] @ pix).unsqueeze(2) * d + Hm[..., 3:, None]  # (B*V,3,D,Ht*Wt)
    p = p.permute(0, 2, 3, 1).contiguous()
    g = p[..., :2] / p[..., 2:3].clamp_min(1e-6)
    g[..., 0], g[..., 1] = 2 * g[..., 0] / Ws - 1, 2 * g[..., 1] / Hs - 1
    warped = F.grid_sample(src_feat.flatten(0, 1), g, mode="bilinear", padding_mode="zeros", align_corners=False)
    return torch.var(warped.view(B, V, -1, D, Ht, Wt), dim=1, unbiased=False)


def depth_regression(depth_values, depth_prob, ci_scale: float, inv_depth: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Soft-argmax depth (B,1,H,W) and the confidence interval (B,2,H,W) = mean ∓ ci_scale·std clipped
    to the hypothesis range (:479-514)."""
    mean = (depth_prob * depth_values).sum(1, keepdim=True)
    std = (depth_prob * (depth_values - mean).square()).sum(1, keepdim=True).clamp_min(1e-12).sqrt()
    half = ci_scale * std
    first, last = depth_values[:, :1], depth_values[:, -1:]
    if inv_depth:  # hypotheses run from large to small disparity
        ci = 1.0 / torch.cat((torch.min(mean + half, first), torch.max(mean - half, last)), 1)
        return 1.0 / mean, ci
    return mean, torch.cat((torch.max(mean - half, first), torch.min(mean + half, last)), 1)


class DepthNet(nn.Module):
    def __init__(self, config: SimpleNamespace) -> None:
        super().__init__()
        mvs, fpn = config.mvs, config.fpn
        self.vol_levels = list(mvs.vol_levels)
        self.vol_scales = list(mvs.vol_scales)
        self.num_stages = len(self.vol_levels)
        self.feat_scales = [fpn.feat_scales[l] for l in self.vol_levels]
        self.feat_dims = [fpn.feat_dims[l] for l in self.vol_levels]
        self.ci_scales = list(mvs.ci_scales)
        self.num_depth = list(mvs.num_depth)
        self.inv_depth = list(mvs.inv_depth)
        self.hip_cost_volume = bool(getattr(mvs, "hip_cost_volume", True))
        # the reference indexes feat_dims by the pyramid level again (depth_net.py:32-37); kept for key/shape parity
        nets = [CostRegNet_small(self.feat_dims[self.vol_levels[0]], mvs.voxel_dim, fpn.base_channels)]
        nets += [CostRegNet(self.feat_dims[self.vol_levels[i]], mvs.voxel_dim, fpn.base_channels) for i in range(1, self.num_stages)]
        self.cost_regs = nn.ModuleList(nets)
        self.nerfs = nn.ModuleList(_AuxStageNerf(config.nerf.nerf_hidden_dims, mvs.voxel_dim, self.feat_dims[i], config.nerf.viewdir_agg)
                                   for i in range(self.num_stages - 1))

    def forward(self, src_images, ms_feats: List[torch.Tensor], src_exts, src_ints, tar_exts, tar_ints, near_far):
        """Returns (depths, depth_ranges, vol_ranges, feat_volumes, rgb_predictions) per stage like the
        reference; rgb_predictions (train-time supervision) is always empty here."""
        B, V, _, H0, W0 = src_images.shape
        depths, ranges, vol_ranges, volumes = [], [], [], []
        search = near_far[..., None, None]  # (B,2,1,1)
        for s in range(self.num_stages):
            feats = ms_feats[self.vol_levels[s]]
            K_src = src_ints.clone()
            K_src[..., :2, :] *= self.feat_scales[s]
            K_tar = tar_ints.clone()
            K_tar[:, :2, :] *= self.vol_scales[s]
            Hs, Ws = int(H0 * self.vol_scales[s]), int(W0 * self.vol_scales[s])
            hyp = get_depth_values(search, self.num_depth[s], self.inv_depth[s]).expand(-1, -1, Hs, Ws)
            if _use_hip(feats, self.hip_cost_volume):
                from ... import costvol
                hyp = hyp.contiguous()
                cost = costvol.build_feature_volume(feats, src_exts, K_src, tar_exts, K_tar, hyp, self.inv_depth[s])
                volume, prob = self.cost_regs[s](cost)
                depth, search = costvol.depth_regression(hyp, prob, self.ci_scales[s], self.inv_depth[s])
            else:
                cost = build_feature_volume(feats, src_exts, K_src, tar_exts, K_tar, hyp, self.inv_depth[s])
                volume, prob = self.cost_regs[s](cost)
                depth, search = depth_regression(hyp, prob, self.ci_scales[s], self.inv_depth[s])
            depths.append(depth.squeeze(1))
            ranges.append(search)
            vol_ranges.append(hyp[:, [0, -1]])
            volumes.append(volume)
            if s < self.num_stages - 1:
                search = F.interpolate(search, scale_factor=self.vol_scales[s + 1] / self.vol_scales[s], mode="bilinear", align_corners=False)
        return depths, ranges, vol_ranges, volumes, []
