"""Depth-guided bundle sampler with the reference's method signatures
(networks/gdb_nerf/bundle_sampler.py), every method backed by one entry of the C ABI:
build_rays -> gdb_build_rays, sample -> gdb_sample, encode -> gdb_encode.  These materialise the
reference's intermediates; `Network.forward` uses the fused kernel instead."""
from typing import Dict, List, Tuple, Union

import torch

from ...engine import HotPathEngine


class BundleSampler:
    def __init__(self, global_num_depth: int, max_mipmap_level: int) -> None:
        self.global_num_depth = global_num_depth
        self.max_mipmap_level = max_mipmap_level
        self.H_orig = self.W_orig = None
        self.rays_o = self.rays_d = self.uv = self.tar_pixel_radius = self.z_axis = self.near = self.far = None
        self._tar = None
        self._engines: Dict[tuple, HotPathEngine] = {}
        self._last: HotPathEngine = None

    def _engine(self, device, b: int, S: int, adaptive: bool, inv_depth: bool) -> HotPathEngine:
        key = (str(device), b, S, bool(adaptive), bool(inv_depth))
        if key not in self._engines:
            self._engines[key] = HotPathEngine(bundle_size=b, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv_depth,
                                               global_num_depth=self.global_num_depth, max_mipmap_level=self.max_mipmap_level, device=device)
        return self._engines[key]

    def build_rays(self, tar_exts: torch.Tensor, tar_ints: torch.Tensor, im_size: Union[Tuple[int, int], List[int]],
                   near: torch.Tensor, far: torch.Tensor) -> None:
        self.H_orig, self.W_orig = int(im_size[0]), int(im_size[1])
        self.near, self.far = near, far
        self._tar = {"tar_ext": tar_exts.contiguous().float(), "tar_int": tar_ints.contiguous().float(),
                     "near_far": torch.stack((near, far), -1).contiguous().float()}
        eng = self._engine(tar_exts.device, 1, 1, False, False)
        eng.prepare(self._tar, im_size=(self.H_orig, self.W_orig))
        r = eng.build_rays()
        self.rays_o, self.rays_d, self.uv = r["rays_o"], r["rays_d"], r["uv"]
        self.tar_pixel_radius, self.z_axis = r["tar_pixel_radius"], r["z_axis"]

    def sample(self, depth_range: torch.Tensor, vol_range: torch.Tensor, b_size: int, max_num_samples: int,
               inv_depth: bool = False, is_adaptive: bool = False):
        if self.rays_o is None:
            raise ValueError("Rays have not been built yet. Please call build_rays() first.")
        eng = self._engine(depth_range.device, b_size, max_num_samples, is_adaptive, inv_depth)
        frame = dict(self._tar, depth_range=depth_range.contiguous().float(), vol_range=vol_range.contiguous().float())
        eng.prepare(frame, im_size=(self.H_orig, self.W_orig))
        s = eng.sample()
        n = int(s["total"].item())  # the reference's boolean-mask compaction has the same host-visible size
        self._last, self._total = eng, s["total"]
        spb, per_batch = s["samples_per_bundle"], s["samples_per_batch"]
        if is_adaptive:  # dtype wart of the reference: float counts on the adaptive path (:179,:191,:242)
            spb, per_batch = spb.float(), per_batch.float()
        return (s["rays_xyz"][:n], s["uvd"][:n], s["z_vals"][:n], s["ball_radii"][:n], s["indices"][:n], per_batch, spb)

    def encode(self, src_images, img_feat, feat_volume, rays_xyz, uvd, ball_radii, src_exts, src_ints, tar_exts, samples_per_batch):
        B, V, Cfr, H, W = img_feat.shape
        b = src_images.shape[-1] // W
        eng = self._last if (self._last is not None and self._last.b == b) else self._engine(img_feat.device, b, 1, False, False)
        c = lambda t: t.contiguous().float()
        near_far = self._tar["near_far"] if self._tar is not None else torch.ones((B, 2), device=img_feat.device)
        dr = torch.ones((B, 2, H, W), device=img_feat.device)
        eng.prepare({"src_images": c(src_images), "img_feat": c(img_feat), "feat_volume": c(feat_volume), "depth_range": dr, "vol_range": dr,
                     "src_exts": c(src_exts), "src_ints": c(src_ints), "tar_ext": c(tar_exts), "tar_int": self._tar["tar_int"] if self._tar else c(src_ints[:, 0]),
                     "near_far": near_far})
        n = rays_xyz.shape[0]
        total = torch.tensor([n], dtype=torch.int64, device=img_feat.device)
        return eng.encode(c(rays_xyz), c(uvd), c(ball_radii), samples_per_batch.to(torch.int64).contiguous(), total)
