"""GDB-NeRF network with the reference's plugin surface (networks/gdb_nerf/network.py): same
constructor config keys, same sub-module names (`feature_net`, `depth_net`, `nerf`, `upsampler` —
the checkpoint prefixes), same `forward(batch) -> (ret, mvs_depths, blend_rgbs)`.

The CNNs run on PyTorch-ROCm.  The hot-path section (reference network.py:145-169) runs on the HIP
library: by default as ONE fused kernel (`gdb_render_bundles_fused`); with `hot_path = "mirrors"`
through the operator mirrors, which materialise the reference's intermediates."""
from types import SimpleNamespace
from typing import Any, Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...engine import HotPathEngine
from ...parallel import StripGather, gather_strips
from . import utils
from .bundle_sampler import BundleSampler
from .decoder_rdn import Decoder
from .depth_net import DepthNet
from .feature_net import FeatureNet
from .nerf import NeRF


class Network(nn.Module):
    def __init__(self, config: SimpleNamespace) -> None:
        super().__init__()
        fpn, mvs, nrf = config.fpn, config.mvs, config.nerf
        self.feature_net = FeatureNet(base_channels=fpn.base_channels, out_channels=fpn.feat_dims)
        self.voxel_dim = mvs.voxel_dim
        self.depth_net = DepthNet(config)

        self.max_num_samples = nrf.max_num_samples
        self.b_size = nrf.bundle_size
        if self.b_size <= 0 or self.b_size & (self.b_size - 1):
            raise ValueError('`Bundle size` must be a power of 2.')
        self.inv_depth = mvs.inv_depth[-1]
        self.is_adaptive = nrf.is_adaptive
        self.global_num_depth, self.max_mipmap_level = nrf.global_num_depth, nrf.max_mipmap_level
        self.sampler = BundleSampler(self.global_num_depth, self.max_mipmap_level)

        # pyramid level whose scale is closest to (not below) the bundle map's 1/b   (reference :40-43)
        self.feat_level = next((i for i, s in enumerate(fpn.feat_scales) if s >= 1.0 / self.b_size), len(fpn.feat_scales))
        feat_dim = fpn.feat_dims[self.feat_level]
        self.nerf_hidden_dims, self.viewdir_agg = nrf.nerf_hidden_dims, nrf.viewdir_agg
        self.render_scale = 1.0
        self.nerf = NeRF(self.nerf_hidden_dims, feat_dim, self.voxel_dim, self.viewdir_agg)

        self.dec_layers = nrf.dec_layers
        self.upsampler = Decoder(feat_dim + 3 + self.voxel_dim, 3, num_feats=64, num_layers=self.dec_layers, upscale_factor=self.b_size)
        self.reweighting = nrf.reweighting
        self.hot_path = getattr(nrf, "hot_path", "fused")  # "fused" | "mirrors"
        # arithmetic of the NeRF MLP in the fused kernel: "f32" (fp32 MFMA, the reference's precision; default) | "f16" (f16 operands) |
        # "f32x" (split-f16 operand pairs: fp32-grade at close to the f16 rate)
        self.precision = {"f32": 1, "f16": 0, "f32x": 2}[str(getattr(nrf, "precision", "f32"))]
        # N1: the decoder on the HIP library (fp32 MFMA implicit-GEMM convolutions, channel-last, reading bundle_feat in place);
        # False keeps the PyTorch-ROCm module.  The HIP decoder takes any number of dense blocks up to 16 (every reference config has 3) and
        # bundle_size 2 (one up stage, folded with out_conv) or 4 (two: decoder_rdn.py:59-62 - the first as four sub-pixel convolutions,
        # the second folded; round 6); b = 1 (no up stage) keeps the PyTorch module.
        self.hip_decoder = bool(getattr(nrf, "hip_decoder", True)) and self.b_size in (2, 4) and 1 <= int(self.dec_layers) <= 16
        # Multi-GPU: "rows" = when torch.distributed is initialised, every rank renders one contiguous strip of
        # bundle-map rows of the frame and ONE all-gather of the packed rows (RCCL over xGMI) leaves the whole bundle map on every
        # rank; decoder and merge then run replicated (the decoder's squeeze-excitation takes a global mean over the image,
        # decoder_rdn.py:10,20, so it does not shard by rows).  "none" (default): every rank renders whole frames.
        self.shard = str(getattr(nrf, "shard", "none"))
        if self.shard not in ("none", "rows"):
            raise ValueError(f"nerf.shard must be 'none' or 'rows', got {self.shard!r}")
        # The intermediates of a frame (packed render, decoder image) always live in per-engine buffers reused frame after frame.
        # The tensors in the returned dict are fresh by default, as the reference's are (a caller may keep them across frames);
        # `nerf.reuse_outputs: true` makes them per-engine buffers too, overwritten by the next forward: no allocation at all in
        # the per-frame path, for loops that consume a frame's outputs before the next one (bench.py, tools/bench_network.py).
        self.reuse_outputs = bool(getattr(nrf, "reuse_outputs", False))
        self._feat_dim = feat_dim
        self._engine = None
        self._gather = None

    # ---- hot path ------------------------------------------------------------------------
    def _get_engine(self, device) -> HotPathEngine:
        if self._engine is None or self._engine.device != torch.device(device):
            self._engine = HotPathEngine(bundle_size=self.b_size, max_num_samples=self.max_num_samples, is_adaptive=self.is_adaptive,
                                         inv_depth=self.inv_depth, global_num_depth=self.global_num_depth,
                                         max_mipmap_level=self.max_mipmap_level, feat_dim=self._feat_dim, voxel_dim=self.voxel_dim,
                                         hid_dim=self.nerf_hidden_dims, viewdir_agg=self.viewdir_agg, device=device)
        self._engine.precision = self.precision
        self._engine.reuse_outputs = self.reuse_outputs
        self._engine.reuse_internal = True
        self.nerf.sync_engine(self._engine)
        if self.hip_decoder:
            # (storage, version) per tensor, as NeRF.param_versions: `p.data = ...` and load_state_dict(assign=True) change the
            # storage without touching the version counter.  (`p.data.copy_()` changes neither: call invalidate_packed_weights().)
            v = tuple((p.data_ptr(), p._version) for p in self.upsampler.parameters())
            if getattr(self._engine, "_dec_versions", None) != v:
                self._engine.load_decoder_weights({k: t.detach() for k, t in self.upsampler.state_dict().items()}, self.dec_layers)
                self._engine._dec_versions = v
        return self._engine

    def invalidate_packed_weights(self) -> None:
        """Force a re-pack of the NeRF and decoder weights on the next forward (needed only after an in-place write through
        `.data`, which PyTorch does not version)."""
        if self._engine is not None:
            self._engine._dec_versions = None
            self._engine._nerf_versions = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed_weights()
        return super()._load_from_state_dict(*args, **kwargs)

    def _dist(self):
        """torch.distributed when this forward is to be row-sharded, else None."""
        if self.shard != "rows":
            return None
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() else None

    def _render_packed(self, eng, B: int, H: int, W: int) -> torch.Tensor:
        """The hot path of the prepared frame into packed rows [bundle_feat | depth | opacity] (n_bundles, Q + 2); row-sharded over
        the ranks of the default process group when `nerf.shard: rows` (reference call site network.py:145-169; the eval loop
        run.py:54-66 calls this once per frame on every rank)."""
        if not eng.fused_supported:
            # bundle_size 1 / 4 (configs/dtu_pretrain.yaml:33 "4 for 4*4"; network.py:31-34 accepts any power of two): the fused kernel
            # is built for b = 2, so the frame goes through the HIP operator mirrors (gdb_sample -> gdb_encode -> gdb_mlp ->
            # gdb_composite), whole on every rank (the mirrors take no row strip).
            return eng.render_unfused_packed()
        dist = self._dist()
        if dist is None:
            return eng.render_packed()
        world, rank = dist.get_world_size(), dist.get_rank()
        C = eng.Q + 2
        # RCCL ("nccl") moves device tensors; a gloo group (CPU tests, the one-GPU rehearsal) exchanges host copies
        stage = dist.get_backend() == "gloo"
        if B == 1:  # one contiguous strip per rank: in-place all-gather, buffers cached per shape
            g = self._gather
            if g is None or (g.H, g.W, g.C, g.world, g.rank) != (H, W, C, world, rank) or g.full.device != torch.device(eng.device):
                g = self._gather = StripGather(H, W, C, world, rank, eng.device, dist, stage_cpu=stage)
            r0, r1 = g.strip
            eng.render_packed(r0, r1, None, g.full)
            return g.gather()
        from ...parallel import row_strip
        r0, r1 = row_strip(H, rank, world)
        full = eng.render_packed(r0, r1)  # fresh, zero-filled outside the strip
        if stage and full.is_cuda:
            return gather_strips(full.cpu(), H, world, dist, B=B).to(full.device)
        return gather_strips(full, H, world, dist, B=B)

    def render_bundles(self, rgbs_feat_rgb_dir, vox_feat, z_vals, indices, samples_per_bundle):
        """MLP + normalised alpha composite on materialised samples (reference network.py:54-91)."""
        sigma, feat = self.nerf(vox_feat, rgbs_feat_rgb_dir)
        n_bundles = samples_per_bundle.shape[0]
        weights, inverse = utils.render_weight_from_density(sigma, indices, n_bundles)
        if self.inv_depth:
            z_vals = 1.0 / z_vals
        feat, depth, opacity = utils.accumulate_value_along_rays(feat, z_vals, weights, indices, n_bundles, inverse)
        if self.inv_depth:
            depth = 1.0 / depth
        return feat, depth, opacity

    def forward(self, batch: Dict[str, Any]) -> Tuple[Dict[str, torch.Tensor], List[torch.Tensor], List[torch.Tensor]]:
        src, tar = batch["src_views"], batch["tar_views"]
        near_far = batch["near_far"]
        src_images = src["rgb"]
        B, V, _, Ho, Wo = src_images.shape
        src_exts, tar_exts = src["extrinsics"], tar["extrinsics"]
        src_ints, tar_ints = src["intrinsics"].clone(), tar["intrinsics"].clone()

        if "render_scale" in batch:
            self.render_scale = batch["render_scale"][0].item()
        if self.render_scale != 1.0:
            src_images = F.interpolate(src_images.flatten(0, 1), scale_factor=self.render_scale, mode="bilinear",
                                       align_corners=False).unflatten(0, (B, V))
            Ho, Wo = src_images.shape[-2:]
            src_ints[..., :2, :] *= self.render_scale
            tar_ints[:, :2, :] *= self.render_scale

        ms_feats = [f.unflatten(0, (B, V)) for f in self.feature_net(src_images.flatten(0, 1))]
        mvs_depths, ranges, vol_ranges, volumes, blend_rgbs = self.depth_net(src_images, ms_feats, src_exts, src_ints, tar_exts, tar_ints, near_far)
        depth_range, vol_range, feat_volume, mvs_depth = ranges[-1], vol_ranges[-1], volumes[-1], mvs_depths[-1]

        b = self.b_size
        H, W = Ho // b, Wo // b
        if depth_range.shape[2:] != (H, W):
            depth_range = F.interpolate(depth_range, size=(H, W), mode="bilinear", align_corners=False)
            vol_range = F.interpolate(vol_range, size=(H, W), mode="bilinear", align_corners=False)
            mvs_depth = F.interpolate(mvs_depth.unsqueeze(1), size=(H, W), mode="nearest").squeeze(1)
        img_feat = ms_feats[self.feat_level]
        if img_feat.shape[-2:] != (H, W):
            img_feat = F.interpolate(img_feat.flatten(0, 1), size=(H, W), mode="bilinear", align_corners=False).unflatten(0, (B, V))
        eng = None
        if self.hot_path == "fused":
            c = lambda t: t.contiguous().float()
            eng = self._get_engine(src_images.device)
            # a rank of a row-sharded forward plans its own strip of bundle-map rows only, and - for frames large enough for the bound's
            # launch to pay - builds only the strip's reach of the pyramids (gdb_prepare_rows; any bundle size)
            dist = self._dist()
            rows = None
            if dist is not None:
                from ...parallel import row_strip
                rows = row_strip(H, dist.get_rank(), dist.get_world_size())
            # N3: the kernel resamples the colour channels itself (no torch.cat / F.interpolate of the source images)
            eng.prepare({"src_images": c(src_images), "fpn_feat": c(img_feat), "feat_volume": c(feat_volume),
                         "depth_range": c(depth_range), "vol_range": c(vol_range), "src_exts": c(src_exts), "src_ints": c(src_ints),
                         "tar_ext": c(tar_exts), "tar_int": c(tar_ints), "near_far": c(near_far)}, rows=rows)
            packed = self._render_packed(eng, B, H, W)
            if self.hip_decoder:
                rgb_c = eng.decode(packed)   # reads channels 3 b^2 .. Q-1 of the packed rows in place
            else:
                rgb_c = self.upsampler(packed[:, 3 * b * b:eng.Q].view(B, H, W, -1).permute(0, 3, 1, 2)).contiguous().float()
            # N1: pixel-shuffle + add (+ re-weighting) + the two x b upsamplings in one HIP kernel, on the packed rows in place
            img, nerf_depth, opacity = eng.merge_packed(packed, rgb_c, self.reweighting)
            return {"rgb": img, "nerf_depth": nerf_depth, "mvs_depth": mvs_depth, "opacity": opacity}, mvs_depths, blend_rgbs
        else:
            rgb_lo = F.interpolate(src_images.flatten(0, 1), size=(H, W), mode="bilinear", align_corners=False).unflatten(0, (B, V))
            img_feat_rgb = torch.cat((img_feat, rgb_lo), dim=2)
            self.sampler.build_rays(tar_exts, tar_ints, (Ho, Wo), near_far[:, 0], near_far[:, 1])
            rays_xyz, uvd, z_vals, ball_radii, indices, per_batch, per_bundle = self.sampler.sample(
                depth_range, vol_range, b, self.max_num_samples, self.inv_depth, self.is_adaptive)
            rfd, vox = self.sampler.encode(src_images, img_feat_rgb, feat_volume, rays_xyz, uvd, ball_radii, src_exts, src_ints, tar_exts, per_batch)
            bundle_feat, bundle_depth, bundle_opacity = self.render_bundles(rfd, vox, z_vals, indices, per_bundle)

        nerf_feat = bundle_feat.view(B, H, W, -1).permute(0, 3, 1, 2)
        n_rgb = 3 * b * b
        rgb_c = self.upsampler(nerf_feat[:, n_rgb:])
        rgb_f = F.pixel_shuffle(nerf_feat[:, :n_rgb], b)
        up = lambda t: F.interpolate(t.view(B, 1, H, W), scale_factor=b, mode="bilinear", align_corners=False).squeeze(1)
        img = rgb_c + rgb_f
        if self.reweighting:
            img = 0.5 * (img + rgb_f)
        ret = {"rgb": img, "nerf_depth": up(bundle_depth), "mvs_depth": mvs_depth, "opacity": up(bundle_opacity)}
        return ret, mvs_depths, blend_rgbs
