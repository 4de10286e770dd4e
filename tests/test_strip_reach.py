"""The construction behind `gdb_prepare_rows`' partial pyramid (gdb_ops.hip `k_strip_bounds`, round 6), checked on the CPU
against the ORACLE's own sample coordinates: every point a row strip samples lies in the convex body { o + d(x, y) z : (x, y) in the strip's
pixel rectangle, z in [z_min, z_max] of the strip's depth prior }, so while its 8 vertices lie in front of a source camera the bounding
box of their projections (+ the mip / bilinear margins around the CLAMPED coordinate) contains every texel and pixel the strip's taps
read.  `strip_bounds_np` restates the kernel's arithmetic in numpy; the taps are the reference's (bundle_sampler.py:327-359 as restated
by oracle/gdb_oracle.py: sub-ray colour taps at K (E p + t), feature taps at (K / b) (E centre + t) on every mip level, clamp
addressing).  The GPU test `test_prepare_rows_builds_only_the_strips_reach` proves the kernel itself (NaN-filled workspaces)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import gdb_oracle as oracle  # noqa: E402
from gdb_nerf_amd import synthetic  # noqa: E402

PT_W, PT_H, MARGIN = 32, 8, 14.0   # pyramid tile of k_prepare; texel margin of k_strip_bounds (a level-3 tap pair reaches 12, + 2)


def strip_bounds_np(frame, bi, r0, r1, b):
    """k_strip_bounds for batch item bi, bundle-map rows [r0, r1): per view (tile x lo, hi, tile y lo, hi, image row lo, hi, whole)."""
    Ho, Wo = frame["src_images"].shape[-2:]
    H, W = Ho // b, Wo // b
    dr = frame["depth_range"][bi][:, r0:r1]
    zmin, zmax = float(dr.min()), float(dr.max())
    bad = (not np.isfinite(dr).all()) or not (dr > 0).all()
    Ei = np.linalg.inv(frame["tar_ext"][bi].astype(np.float64)); Ki = np.linalg.inv(frame["tar_int"][bi].astype(np.float64))
    M = (Ei[:3, :3] @ Ki).astype(np.float32); o = Ei[:3, 3].astype(np.float32)
    tilesX, tilesY = (W + PT_W - 1) // PT_W, (H + PT_H - 1) // PT_H
    out = []
    for v in range(frame["src_exts"].shape[1]):
        E, K = frame["src_exts"][bi, v].astype(np.float32), frame["src_ints"][bi, v].astype(np.float32)
        whole = bad or not (zmin > 0 and zmax >= zmin)
        us, vs = [], []
        for c in range(8):
            if whole:
                break
            px, py, z = (Wo if c & 1 else 0.0), (r1 * b if c & 2 else r0 * b), (zmax if c & 4 else zmin)
            p = o + (M @ np.array([px, py, 1.0], np.float32)) * np.float32(z)
            im = K @ (E[:3, :3] @ p + E[:3, 3])
            if not (im[2] > 1e-3 * zmin) or not np.isfinite(im).all():
                whole = True
                break
            us.append(im[0] / im[2]); vs.append(im[1] / im[2])
        if whole:
            out.append((0, tilesX - 1, 0, tilesY - 1, 0, Ho - 1, True))
            continue
        xlo, xhi, ylo, yhi = min(us), max(us), min(vs), max(vs)
        cl = lambda t, hi: min(max(t, 0.0), hi)
        tx0, tx1 = max(cl(xlo / b - 0.5, W - 1) - MARGIN, 0.0), min(cl(xhi / b - 0.5, W - 1) + MARGIN, W - 1)
        ty0, ty1 = max(cl(ylo / b - 0.5, H - 1) - MARGIN, 0.0), min(cl(yhi / b - 0.5, H - 1) + MARGIN, H - 1)
        out.append((int(np.floor(tx0)) // PT_W, min(int(np.floor(tx1)) // PT_W, tilesX - 1), int(np.floor(ty0)) // PT_H, min(int(np.floor(ty1)) // PT_H, tilesY - 1),
                    int(max(np.floor(cl(ylo - 0.5, Ho - 1)) - 2, 0)), int(min(np.ceil(cl(yhi - 0.5, Ho - 1)) + 2, Ho - 1)), False))
    return out


@pytest.mark.parametrize("scene,fs,S,adaptive,inv,b", [("dtu", None, 3, True, False, 2), ("nerf", (1.0, 2.3, 0.6), 6, True, False, 2), ("llff", (0.5, 1.0, 4.0), 4, False, True, 2),
                                                      ("dtu", (1.0, 1.5, 0.8), 3, True, False, 4)])
def test_every_tap_of_a_strip_lies_inside_its_bound(scene, fs, S, adaptive, inv, b):
    Ho, Wo, V, world = 128, 192, 3, 4
    frame = synthetic.make_frame(Ho, Wo, V=V, bundle_size=b, scene=scene, seed=5, src_focal_scale=fs)
    H, W = Ho // b, Wo // b
    rays = oracle.build_rays(frame["tar_ext"], frame["tar_int"], Ho, Wo)
    nf = frame["near_far"].astype(np.float32)
    smp = oracle.sample_bundles(rays, frame["depth_range"], frame["vol_range"], nf[:, 0], nf[:, 1], b, S, 64, inv, adaptive)
    rows_of = (smp["indices"] // W) % H                      # bundle-map row of every sample (batch 1)
    pts_all = np.transpose(smp["rays_xyz"], (0, 2, 1))       # (N, bb, 3)
    levels = 3
    nonwhole = 0
    for k in range(world):
        r0, r1 = k * H // world, (k + 1) * H // world
        sel = (rows_of >= r0) & (rows_of < r1)
        pts = pts_all[sel].astype(np.float32)                # the strip's sub-ray points
        bounds = strip_bounds_np(frame, 0, r0, r1, b)
        for v in range(V):
            tx_lo, tx_hi, ty_lo, ty_hi, iy_lo, iy_hi, whole = bounds[v]
            nonwhole += 0 if whole else 1
            E, K = frame["src_exts"][0, v].astype(np.float32), frame["src_ints"][0, v].astype(np.float32)
            cam = pts @ E[:3, :3].T + E[:3, 3]               # (n, bb, 3)   bundle_sampler.py:327-329
            im = cam @ K.T
            zc = np.maximum(im[..., 2], np.float32(1e-6))
            py = np.clip(im[..., 1] / zc - 0.5, 0, Ho - 1)  # grid_sample border, align_corners=False: pixel coordinate - 0.5   :336
            y0 = np.floor(py).astype(int); y1 = np.minimum(y0 + 1, Ho - 1)
            assert y0.min() >= iy_lo and y1.max() <= iy_hi, (k, v, "colour taps", y0.min(), y1.max(), iy_lo, iy_hi)
            cc = cam.mean(axis=1)                            # sphere centre in the camera frame   :340
            Ks = K.copy(); Ks[:2] /= b
            ci = cc @ Ks.T
            z2 = np.maximum(ci[:, 2], np.float32(1e-6))
            tu, tv = ci[:, 0] / z2 / W, ci[:, 1] / z2 / H    # :351-353
            for l in range(levels + 1):                      # every mip level a footprint may select (nvdiffrast clamp addressing)
                Wl, Hl = W >> l, H >> l
                x = np.clip(tu * Wl - 0.5, 0, Wl - 1); y = np.clip(tv * Hl - 0.5, 0, Hl - 1)
                x0 = np.floor(x).astype(int); x1 = np.minimum(x0 + 1, Wl - 1)
                yy0 = np.floor(y).astype(int); yy1 = np.minimum(yy0 + 1, Hl - 1)
                # the level-0 texels a level-l texel is built from lie in the tile that builds it
                assert (x0.min() << l) // PT_W >= tx_lo and (((x1.max() + 1) << l) - 1) // PT_W <= tx_hi, (k, v, l, "x")
                assert (yy0.min() << l) // PT_H >= ty_lo and (((yy1.max() + 1) << l) - 1) // PT_H <= ty_hi, (k, v, l, "y")
    assert nonwhole > 0   # (the bound was actually formed for some strip and view: the test is not vacuous)
