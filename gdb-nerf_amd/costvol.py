"""Torch-facing wrappers of the two "next"-row kernels: the plane-sweep cost volume
and the depth regression of the cascade MVS stage, same signatures as the reference functions
(networks/gdb_nerf/depth_net.py:424-514).  CUDA tensors only — there is no CPU fallback here either."""
import ctypes as C
from typing import Tuple

import torch

from . import _lib


def _c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise ValueError(f"{name} must be a float32 CUDA tensor")
    return t.contiguous()


def build_feature_volume(src_feat, src_exts, src_ints, tar_exts, tar_ints, depth_values, inv_depth: bool, pair_layout: bool = True) -> torch.Tensor:
    """pair_layout: let the library re-lay the source maps channel-pair-interleaved first (even C, V <= 4): bit-identical, faster."""
    lib = _lib.load()
    B, V, Cc, Hs, Ws = src_feat.shape
    D, Ht, Wt = depth_values.shape[1:]
    if depth_values.shape[0] != B or tuple(src_exts.shape) != (B, V, 4, 4) or tuple(src_ints.shape) != (B, V, 3, 3) \
            or tuple(tar_exts.shape) != (B, 4, 4) or tuple(tar_ints.shape) != (B, 3, 3):
        raise ValueError("inconsistent cost-volume shapes")
    args = [_c(t, n) for t, n in ((src_feat, "src_feat"), (src_exts, "src_exts"), (src_ints, "src_ints"), (tar_exts, "tar_exts"),
                                  (tar_ints, "tar_ints"), (depth_values, "depth_values"))]
    out = torch.empty((B, Cc, D, Ht, Wt), device=src_feat.device)
    ws = torch.empty((B * V * 12,), device=src_feat.device)
    # scratch for the channel-pair copy of the source maps (the sweep then loads 16 bytes per channel pair).  Allocated per call:
    # torch's caching allocator hands the block back without a device allocation and keeps it stream-safe (a module-level cache keyed
    # by size was shared by every caller on every stream, and the cascade's two stage sizes evicted each other: ADVICE r04)
    n_pair = B * V * Cc * Hs * Ws if (pair_layout and Cc % 2 == 0 and V <= 4) else 0
    pw = torch.empty((n_pair,), device=src_feat.device) if n_pair else None
    _lib.check(lib.gdb_build_feature_volume_ws(*(t.data_ptr() for t in args), B, V, Cc, Hs, Ws, D, Ht, Wt, int(bool(inv_depth)),
                                               ws.data_ptr(), None if pw is None else pw.data_ptr(), out.data_ptr(),
                                               torch.cuda.current_stream(src_feat.device).cuda_stream))
    return out


def depth_regression(depth_values, depth_prob, ci_scale: float, inv_depth: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    B, D, H, W = depth_values.shape
    dv, pr = _c(depth_values, "depth_values"), _c(depth_prob, "depth_prob")
    if tuple(pr.shape) != (B, D, H, W):
        raise ValueError("depth_prob shape differs from depth_values")
    depth = torch.empty((B, 1, H, W), device=dv.device)
    ci = torch.empty((B, 2, H, W), device=dv.device)
    _lib.check(lib.gdb_depth_regression(dv.data_ptr(), pr.data_ptr(), B, D, H, W, C.c_float(float(ci_scale)), int(bool(inv_depth)),
                                        depth.data_ptr(), ci.data_ptr(), torch.cuda.current_stream(dv.device).cuda_stream))
    return depth, ci
