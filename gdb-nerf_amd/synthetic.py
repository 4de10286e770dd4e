"""Seeded synthetic frames for the hot path.

No dataset or checkpoint exists offline, so tests and `bench.py` drive the path with frames of
the right shape: DTU-scale cameras (mm), smoothed random images, random feature maps whose
last three channels are the 2x2-averaged source RGB (network.py:162-164 of the reference),
a random cost volume and a low-pass depth prior whose width spreads the adaptive sample
counts over 1..S_max.  numpy only; nothing here touches the oracle or the GPU.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

SCENES = {
    # name: (near, far, look-at depth, source-camera circle radius)
    "dtu": (425.0, 905.0, 650.0, 60.0),
    "llff": (1.2, 12.0, 4.0, 0.35),
    "nerf": (2.5, 5.5, 4.0, 0.6),
}


def _box(a: np.ndarray, k: int) -> np.ndarray:
    """k x k box filter over the last two axes, edge-replicated."""
    p = k // 2
    pad = [(0, 0)] * (a.ndim - 2) + [(p, p), (p, p)]
    ap = np.pad(a, pad, mode="edge")
    out = np.zeros_like(a)
    H, W = a.shape[-2:]
    for dy in range(k):
        for dx in range(k):
            out += ap[..., dy:dy + H, dx:dx + W]
    return (out / F32(k * k)).astype(F32)


def look_at_w2c(cam: np.ndarray, target: np.ndarray) -> np.ndarray:
    """World-to-camera 4x4 of a camera at `cam` looking at `target` (x right, y down, z forward)."""
    z = target - cam
    z = z / np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack((x, y, z), axis=0)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ cam
    return E.astype(F32)


def make_frame(Ho: int, Wo: int, V: int = 3, B: int = 1, *, bundle_size: int = 2, feat_dim: int = 16,
               voxel_dim: int = 8, num_depth: int = 8, scene: str = "dtu", seed: int = 0,
               smooth_feat: int = 0, smooth_vol: int = 0, src_focal_scale=None) -> dict:
    """One batch of hot-path inputs, all float32 numpy arrays (layouts as the reference's)."""
    near, far, look, radius = SCENES[scene]
    rng = np.random.default_rng(seed)
    b = bundle_size
    H, W = Ho // b, Wo // b
    R = far - near

    f = 1.125 * Wo
    K = np.array([[f, 0, Wo / 2], [0, f, Ho / 2], [0, 0, 1]], dtype=F32)
    tar_ext = np.broadcast_to(np.eye(4, dtype=F32), (B, 4, 4)).copy()
    tar_int = np.broadcast_to(K, (B, 3, 3)).copy()
    src_exts = np.empty((B, V, 4, 4), dtype=F32)
    for bi in range(B):
        for v in range(V):
            th = 2 * np.pi * (v + 0.25 * bi) / V
            cam = np.array([radius * np.cos(th), radius * np.sin(th), 0.0])
            src_exts[bi, v] = look_at_w2c(cam, np.array([0.0, 0.0, look]))
    src_ints = np.broadcast_to(K, (B, V, 3, 3)).copy()
    if src_focal_scale is not None:  # zoomed source cameras push the footprint onto coarser mips
        for v in range(V):
            src_ints[:, v, 0, 0] *= F32(src_focal_scale[v % len(src_focal_scale)])
            src_ints[:, v, 1, 1] *= F32(src_focal_scale[v % len(src_focal_scale)])

    src_images = _box(rng.random((B, V, 3, Ho, Wo), dtype=F32), 5)
    feat = rng.standard_normal((B, V, feat_dim, H, W), dtype=F32)
    if smooth_feat:
        feat = _box(feat, smooth_feat) * F32(smooth_feat)
    rgb_lo = src_images.reshape(B, V, 3, H, b, W, b).mean(axis=(4, 6), dtype=F32)
    img_feat = np.concatenate((feat, rgb_lo), axis=2).astype(F32)
    feat_volume = rng.standard_normal((B, voxel_dim, num_depth, H, W), dtype=F32)
    if smooth_vol:  # a cost volume without per-voxel white noise (the regularised volume of a trained net is smooth in x, y)
        feat_volume = (_box(feat_volume, smooth_vol) * F32(smooth_vol)).astype(F32)

    mid = _box((near + 0.2 * R + 0.6 * R * rng.random((B, 1, H, W), dtype=F32)).astype(F32), 9)
    half = ((0.2 + 1.8 * rng.random((B, 1, H, W), dtype=F32)) * F32(R / 64)).astype(F32)
    depth_range = np.concatenate((mid - half, mid + half), axis=1).astype(F32)
    vol_range = np.concatenate((np.maximum(mid - F32(R / 16), F32(near)),
                                np.minimum(mid + F32(R / 16), F32(far))), axis=1).astype(F32)
    near_far = np.broadcast_to(np.array([near, far], dtype=F32), (B, 2)).copy()
    return {
        "src_images": src_images, "img_feat": img_feat, "feat_volume": feat_volume,
        "depth_range": depth_range, "vol_range": vol_range,
        "src_exts": src_exts, "src_ints": src_ints, "tar_ext": tar_ext, "tar_int": tar_int,
        "near_far": near_far,
    }


NERF_PARAM_SHAPES = (
    # state-dict key prefix (nerf.py:20-56 of the reference), (out, in)
    ("view_fc.0", lambda C, Cv, hid: (C + 3, 4)),
    ("global_fc.0", lambda C, Cv, hid: (32, 3 * (C + 3))),
    ("agg_w_fc.0", lambda C, Cv, hid: (1, 32)),
    ("fc.0", lambda C, Cv, hid: (16, 32)),
    ("lr0.0", lambda C, Cv, hid: (hid, Cv + 16)),
    ("sigma.0", lambda C, Cv, hid: (1, hid)),
    ("weight.0", lambda C, Cv, hid: (hid, hid + Cv + 16 + C + 3 + 4)),
    ("weight.2", lambda C, Cv, hid: (1, hid)),
    ("feat_head.0", lambda C, Cv, hid: (Cv, hid)),
)


def make_nerf_weights(feat_dim: int = 16, voxel_dim: int = 8, hid: int = 64, seed: int = 0) -> dict:
    """Random MLP weights with nn.Linear's default init ranges (U(-1/sqrt(in), 1/sqrt(in))),
    keyed like the reference NeRF state dict.  Biases are drawn too so that they matter."""
    rng = np.random.default_rng(seed + 1000)
    w = {}
    for name, shp in NERF_PARAM_SHAPES:
        o, i = shp(feat_dim, voxel_dim, hid)
        bound = 1.0 / np.sqrt(i)
        w[name + ".weight"] = rng.uniform(-bound, bound, (o, i)).astype(F32)
        w[name + ".bias"] = rng.uniform(-bound, bound, (o,)).astype(F32)
    return w
