"""Fixture F8: the reference's own `build_feature_volume`, `get_depth_values` and `depth_regression`
(networks/gdb_nerf/depth_net.py:399-514; imports with no placeholders) on seeded inputs, for both stage
shapes (disparity-uniform coarse stage, depth-uniform fine stage with per-pixel ranges)."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from gdb_nerf_amd import synthetic  # noqa: E402


def main():
    from networks.gdb_nerf import depth_net as ref
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = {}
    for tag, (C, D, scale_f, scale_v, inv, seed) in {"coarse": (32, 12, 0.25, 0.125, True, 1), "fine": (16, 8, 0.5, 0.5, False, 2)}.items():
        Ho, Wo, V, B = 64, 96, 3, 2
        fr = synthetic.make_frame(Ho, Wo, V=V, B=B, seed=seed, src_focal_scale=(1.0, 1.3, 0.8))
        rng = np.random.default_rng(seed)
        Hs, Ws, Ht, Wt = int(Ho * scale_f), int(Wo * scale_f), int(Ho * scale_v), int(Wo * scale_v)
        feat = rng.standard_normal((B, V, C, Hs, Ws)).astype(np.float32)
        Ks, Kt = fr["src_ints"].copy(), fr["tar_int"].copy()
        Ks[..., :2, :] *= scale_f
        Kt[:, :2, :] *= scale_v
        if inv:
            nf = t(fr["near_far"])[..., None, None]
            dv = ref.get_depth_values(nf, D, True).expand(-1, -1, Ht, Wt).contiguous()
        else:
            mid = 500 + 300 * rng.random((B, 1, Ht, Wt)).astype(np.float32)
            rngs = np.concatenate((mid - 20, mid + 25), 1).astype(np.float32)
            dv = ref.get_depth_values(t(rngs), D, False)
        with torch.no_grad():
            vol = ref.build_feature_volume(t(feat), t(fr["src_exts"]), t(Ks), t(fr["tar_ext"]), t(Kt), dv, inv)
            prob = torch.softmax(torch.from_numpy(rng.standard_normal((B, D, Ht, Wt)).astype(np.float32)), 1)
            depth, ci = ref.depth_regression(dv, prob, 1.0, inv)
        out.update({f"{tag}_{k}": v for k, v in dict(src_feat=feat, src_exts=fr["src_exts"], src_ints=Ks, tar_ext=fr["tar_ext"], tar_int=Kt,
                                                      depth_values=dv.numpy(), inv_depth=inv, volume=vol.numpy(), prob=prob.numpy(),
                                                      depth=depth.numpy(), ci=ci.numpy()).items()})
    path = os.path.join(HERE, "F8_costvol.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
