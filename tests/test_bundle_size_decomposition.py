"""The algebra behind the fused `bundle_size` 1 / 4 path (gdb_fused.hip: `render_list_body<.., BB = 1>` + `k_bundle_colours`, round 6), checked
on the CPU with the ORACLE alone (oracle/gdb_oracle.py, the restatement of the reference): (1) everything of a sample but its 3 b^2 sub-ray
colours hangs on the bundle's MEAN ray, and the mean of the b^2 sub-ray directions / coordinates / points is the direction / coordinate /
point of the ray through the bundle's mean PIXEL (bundle_sampler.py:67-71, :99-104, :254-256: the rays are linear in the pixel);
(2) the MLP never sees the colours (nerf.py:98), so a bundle's colour outputs are
    sum_k W_k  sum_v w_kv  rgb_kv        W = normalised composite weights (utils.py:35-41), w = the views' softmax (nerf.py:108-110)
with the two sums exchanged - what the second launch computes from the weights the first one leaves.  The GPU tests prove the kernels
(tests/test_hip_parity.py::test_fused_bundle_size_1_and_4_vs_oracle, fixture F7d)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import gdb_oracle as oracle  # noqa: E402
from gdb_nerf_amd import synthetic  # noqa: E402

F32 = np.float32


@pytest.mark.parametrize("b,Ho,Wo,V,S,adaptive,inv,scene", [(4, 64, 96, 3, 3, True, False, "dtu"), (4, 48, 64, 2, 6, False, True, "nerf"), (1, 12, 20, 3, 4, True, False, "llff")])
def test_centre_ray_and_exchanged_sums(b, Ho, Wo, V, S, adaptive, inv, scene):
    frame = synthetic.make_frame(Ho, Wo, V=V, bundle_size=b, scene=scene, seed=13, src_focal_scale=(1.0, 1.6, 0.7))
    w = synthetic.make_nerf_weights(seed=4)
    H, W = Ho // b, Wo // b
    rays = oracle.build_rays(frame["tar_ext"], frame["tar_int"], Ho, Wo)
    nf = frame["near_far"].astype(F32)
    smp = oracle.sample_bundles(rays, frame["depth_range"], frame["vol_range"], nf[:, 0], nf[:, 1], b, S, 64, inv, adaptive)
    # ---- (1) the bundle's mean ray is the ray through its mean pixel --------------------------------------------------------------
    bun = oracle.assemble_bundles(rays, frame["depth_range"], frame["vol_range"], b)
    c2w = np.linalg.inv(frame["tar_ext"][0].astype(np.float64)); kinv = np.linalg.inv(frame["tar_int"][0].astype(np.float64))
    M = c2w[:3, :3] @ kinv
    xs = (np.arange(W) * b + 0.5 * b)[None, :].repeat(H, 0); ys = (np.arange(H) * b + 0.5 * b)[:, None].repeat(W, 1)
    pix = np.stack((xs, ys, np.ones_like(xs)), -1).reshape(-1, 3)
    d_centre = (pix @ M.T).astype(F32)                                    # ray_dir(mean pixel)
    d_mean = bun["d"].mean(axis=-1, dtype=F32)                            # mean of the b^2 sub-ray directions   :99
    scale = float(np.abs(d_mean).max())
    assert np.abs(d_centre - d_mean).max() <= 4e-6 * scale
    uv_centre = np.stack((2 * xs / Wo - 1, 2 * ys / Ho - 1), -1).reshape(-1, 2).astype(F32)
    assert np.abs(uv_centre - bun["uv"]).max() <= 2e-6
    centre_pts = smp["rays_xyz"].mean(axis=-1, dtype=F32)                 # mean of the sub-ray points   :256
    z = smp["z_vals"]
    o = bun["o"][smp["indices"]]
    assert np.abs(o + d_centre[smp["indices"]] * z[:, None] - centre_pts).max() <= 4e-6 * float(np.abs(centre_pts).max())
    # ---- (2) the colours: composite of the blend = the exchanged double sum ------------------------------------------------------
    rfd, vox = oracle.encode(frame["src_images"], frame["img_feat"], frame["feat_volume"], smp["rays_xyz"], smp["uvd"], smp["ball_radii"],
                             frame["src_exts"], frame["src_ints"], frame["tar_ext"], smp["samples_per_batch"], Ho, Wo, 3)
    nb = H * W
    bf, depth, opac = oracle.render_bundles(w, rfd, vox, smp["z_vals"], smp["indices"], nb, inv)
    sigma, feat, bw = oracle.nerf_mlp(w, vox, rfd, return_blend_weights=True)
    Wk = oracle.render_weights(sigma, smp["indices"], nb)                 # normalised composite weight of every sample
    ncol = 3 * b * b
    # the MLP's outputs do not depend on the colours (nerf.py:98): zeroed colours leave sigma, the blend weights and the other channels unchanged
    rfd0 = rfd.copy(); rfd0[..., :ncol] = 0
    s0, f0, bw0 = oracle.nerf_mlp(w, vox, rfd0, return_blend_weights=True)
    assert np.array_equal(s0, sigma) and np.array_equal(bw0, bw) and np.array_equal(f0[:, ncol:], feat[:, ncol:])
    wv = (Wk[None, :, None] * bw).astype(F32)                             # (V, N, 1): what the list kernel leaves per (sample, view)
    col = np.zeros((nb, ncol), F32)
    np.add.at(col, smp["indices"], np.sum(wv * rfd[..., :ncol], axis=0, dtype=F32))
    assert np.abs(col - bf[:, :ncol]).max() <= 2e-6
