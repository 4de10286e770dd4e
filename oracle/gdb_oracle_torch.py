"""Pure-PyTorch CPU restatement of the GDB-NeRF hot path (A1-A7 of SURVEY.md 8(a)) - the CPU BASELINE of BASELINE.md 3.

TEST INFRASTRUCTURE ONLY, like `gdb_oracle.py` beside it: only `tests/` and the `cpu_baseline` leg of `bench.py` import it.
It exists because the numpy oracle is element-wise single-threaded numpy (its 256-thread row was slower than its 8-thread row):
this one uses the torch CPU kernels the reference itself would run on a CPU (`F.grid_sample`, `torch.var_mean`, `F.linear`,
`cumprod`, `index_add_`), which thread over `torch.set_num_threads(n)`.  It is pinned to the numpy oracle - and through it to
the reference's golden fixtures - by `tests/test_oracle_golden.py::test_torch_restatement_matches_the_numpy_oracle`.

Each function cites the reference lines it follows (KLMAV-CUC/GDB-NeRF, networks/gdb_nerf/).  The three third-party ops
(`nvdiffrast.torch.texture`, `nerfacc.volrend.*`) are restated from their published semantics exactly as in `gdb_oracle.py`
(parity unpinned by the reference; see that file's header).
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F


def _t(x) -> torch.Tensor:
    return torch.as_tensor(x, dtype=torch.float32)


def build_rays(tar_ext: torch.Tensor, tar_int: torch.Tensor, Ho: int, Wo: int) -> Dict[str, torch.Tensor]:
    """bundle_sampler.py:30-74: pixel-centre grid, unnormalised directions [x, y, 1] (R_c2w K^-1)^T, pixel radius."""
    ys, xs = torch.meshgrid(torch.arange(Ho, dtype=torch.float32) + 0.5, torch.arange(Wo, dtype=torch.float32) + 0.5, indexing="ij")
    uv = torch.stack((2.0 * xs / Wo - 1.0, 2.0 * ys / Ho - 1.0), dim=-1)                      # :53-56
    c2w = torch.inverse(tar_ext)                                                              # :62
    pix = torch.stack((xs, ys, torch.ones_like(xs)), dim=-1).reshape(1, -1, 3)
    M = c2w[:, :3, :3] @ torch.inverse(tar_int)                                               # :67-70
    rays_d = (pix @ M.transpose(1, 2)).reshape(-1, Ho, Wo, 3)
    pixr = 1.0 / torch.sqrt(tar_int[:, 0, 0] * tar_int[:, 1, 1] * math.pi)                    # :74
    return {"rays_d": rays_d, "uv": uv, "rays_o": c2w[:, :3, 3], "z_axis": c2w[:, :3, 2], "tar_pixel_radius": pixr}


def sample_bundles(rays, depth_range, vol_range, near, far, b, S_max, global_num_depth, inv_depth, adaptive):
    """bundle_sampler.py:76-265: bundle assembly, (adaptive) sample counts, bundle-major / sample-minor compaction, per-sample
    geometry."""
    B, Ho, Wo, _ = rays["rays_d"].shape
    H, W, bb = Ho // b, Wo // b, b * b
    if inv_depth:                                                                             # :224-226
        depth_range, vol_range, near, far = 1.0 / depth_range, 1.0 / vol_range, 1.0 / near, 1.0 / far
    d = rays["rays_d"].reshape(B, H, b, W, b, 3).permute(0, 1, 3, 5, 2, 4).reshape(B, H, W, 3, bb)   # channel c, sub-ray by*b+bx (:100)
    bundle_d = d.mean(dim=-1)                                                                 # :99
    cos = (bundle_d * rays["z_axis"][:, None, None]).sum(-1) / bundle_d.norm(dim=-1)          # :102
    uvm = rays["uv"].reshape(H, b, W, b, 2).mean(dim=(1, 3))                                  # :104
    disk = b * rays["tar_pixel_radius"]                                                       # :106
    nb = B * H * W
    nearv, farv = depth_range[:, 0].reshape(nb), depth_range[:, 1].reshape(nb)
    if adaptive:                                                                              # :156-191
        miniv = ((far - near).abs() / global_num_depth)[:, None, None].expand(B, H, W).reshape(nb)   # :227-232
        spb = torch.ceil((farv - nearv).abs() / miniv).clamp(1, S_max)                        # :179
    else:
        spb = torch.full((nb,), float(S_max))
    cnt = spb.to(torch.int64)
    indices = torch.repeat_interleave(torch.arange(nb), cnt)                                  # :182-189
    first = torch.cumsum(cnt, 0) - cnt
    k = (torch.arange(indices.numel()) - first[indices]).to(torch.float32)
    n_i, f_i, c_i = nearv[indices], farv[indices], spb[indices]
    step = (f_i - n_i) / c_i
    t0, t1 = n_i + step * k, n_i + step * (k + 1.0)                                           # :183
    z = 0.5 * (t0 + t1)                                                                       # :246
    vn, vf = vol_range[:, 0].reshape(nb)[indices], vol_range[:, 1].reshape(nb)[indices]
    dn = 2.0 * (z - vn) / (vf - vn) - 1.0                                                     # :247
    if inv_depth:
        z = 1.0 / z                                                                           # :250-251
    bidx = indices // (H * W)
    uvd = torch.cat((uvm.reshape(1, H * W, 2).expand(B, -1, -1).reshape(nb, 2)[indices], dn[:, None]), dim=1)
    o = rays["rays_o"][bidx]
    xyz = o[:, :, None] + d.reshape(nb, 3, bb)[indices] * z[:, None, None]                   # :254-255
    ctr = xyz.mean(dim=-1)                                                                    # :256
    dist = (ctr - o).norm(dim=-1)                                                             # :259
    cosb, diskb = cos.reshape(nb), disk[:, None, None].expand(B, H, W).reshape(nb)
    tt = torch.sqrt(torch.clamp(1.0 / (cosb * cosb) - 1.0, min=1e-12)) - diskb
    unit = diskb * cosb / torch.sqrt(tt * tt + 1.0)                                           # :262
    per_batch = torch.zeros(B, dtype=torch.int64).index_add_(0, torch.arange(nb) // (H * W), cnt)   # :242
    return {"rays_xyz": xyz, "uvd": uvd, "z_vals": z, "ball_radii": unit[indices] * dist, "indices": indices,
            "samples_per_batch": per_batch, "samples_per_bundle": spb}


def build_mips(tex: torch.Tensor, max_level: int) -> List[torch.Tensor]:
    """nvdiffrast mip chain: 2x2 box average while both extents stay even (tex (V,C,H,W))."""
    levels = [tex]
    while len(levels) <= max_level:
        h, w = levels[-1].shape[-2:]
        if h < 2 or w < 2 or h % 2 or w % 2:
            break
        levels.append(F.avg_pool2d(levels[-1], 2))
    return levels


def texture_mip(pyramid: List[torch.Tensor], uv: torch.Tensor, level: torch.Tensor) -> torch.Tensor:
    """nvdiffrast.torch.texture(tex, uv, mip_level_bias=level, boundary_mode='clamp', max_mip_level=L), linear-mipmap-linear
    (bundle_sampler.py:355-359): texel centres at (i + .5)/W, clamp-to-edge = grid_sample(border, align_corners=False) on
    2 uv - 1; level clamped to [0, L], NaN -> 0.  uv (V,N,2), level (V,N) -> (V,N,C)."""
    L = len(pyramid) - 1
    lv = torch.nan_to_num(level, nan=0.0, neginf=0.0, posinf=float(L)).clamp(0.0, float(L))
    l0 = torch.floor(lv)
    l1 = torch.clamp(l0 + 1.0, max=float(L))
    fr = (lv - l0)[..., None]
    g = (2.0 * uv - 1.0)[:, :, None, :]
    out = torch.zeros(uv.shape[0], uv.shape[1], pyramid[0].shape[1])
    for l, tex in enumerate(pyramid):
        use0, use1 = l0 == l, (l1 == l) & (fr[..., 0] > 0)
        if not (use0.any() or use1.any()):
            continue
        s = F.grid_sample(tex, g, mode="bilinear", padding_mode="border", align_corners=False)[:, :, :, 0].permute(0, 2, 1)
        out = out + s * (use0[..., None] * (1.0 - fr) + use1[..., None] * fr)
    return out


def encode(src_images, img_feat, feat_volume, smp, src_exts, src_ints, tar_exts, b, Ho, Wo, max_mip_level):
    """bundle_sampler.py:267-371."""
    B, V, Cf, H, W = img_feat.shape
    bb = b * b
    xyz, uvd, ball = smp["rays_xyz"], smp["uvd"], smp["ball_radii"]
    N = xyz.shape[0]
    tar_c = torch.inverse(tar_exts)[:, :3, 3]
    src_c = torch.inverse(src_exts)[..., :3, 3]                                               # :304-305
    Ks = src_ints.clone()
    Ks[..., :2, :] = Ks[..., :2, :] / b                                                       # :311-312
    src_pixr = 1.0 / torch.sqrt(Ks[..., 0, 0] * Ks[..., 1, 1] * math.pi)                      # :313
    out = torch.empty(V, N, 3 * bb + Cf + 4)
    vox = torch.empty(N, feat_volume.shape[1])
    start = 0
    for bi in range(B):                                                                       # :318
        n = int(smp["samples_per_batch"][bi])
        sl = slice(start, start + n)
        vox[sl] = F.grid_sample(feat_volume[bi:bi + 1], uvd[sl].view(1, 1, 1, n, 3), mode="bilinear", padding_mode="border",
                                align_corners=False)[0, :, 0, 0].t()                          # :322-324
        pts = xyz[sl].permute(0, 2, 1).reshape(-1, 3)
        ph = torch.cat((pts, torch.ones(pts.shape[0], 1)), dim=1)
        cam = (ph @ src_exts[bi].transpose(1, 2))[..., :3]                                    # (V, n*bb, 3)   :327-329
        im = cam @ src_ints[bi].transpose(1, 2)                                               # :332
        zc = im[..., 2:].clamp(min=1e-6)                                                      # :333
        g = torch.cat((2.0 * (im[..., :1] / zc) / Wo - 1.0, 2.0 * (im[..., 1:2] / zc) / Ho - 1.0), dim=-1)   # :334
        rgb = F.grid_sample(src_images[bi], g[:, :, None, :], mode="bilinear", padding_mode="border", align_corners=False)[..., 0]
        out[:, sl, :3 * bb] = rgb.reshape(V, 3, n, bb).permute(0, 2, 1, 3).reshape(V, n, 3 * bb)   # channel c*b^2 + sub   :336-337
        ccam = cam.reshape(V, n, bb, 3).mean(dim=2)                                           # :340
        dist = ccam.norm(dim=-1)
        sec2 = (dist / ccam[..., 2]) ** 2                                                     # :343-344
        a = torch.sqrt(torch.clamp((dist / ball[sl][None]) ** 2 - 1.0, min=1e-12))
        c = torch.sqrt(torch.clamp(sec2 - 1.0, min=1e-12))
        level = torch.log2(sec2 / (a + c) / src_pixr[bi][:, None])                            # :346-348
        cim = ccam @ Ks[bi].transpose(1, 2)                                                   # :351-352
        zc2 = cim[..., 2].clamp(min=1e-6)
        uv = torch.stack((cim[..., 0] / zc2 / W, cim[..., 1] / zc2 / H), dim=-1)              # :353
        out[:, sl, 3 * bb:3 * bb + Cf] = texture_mip(build_mips(img_feat[bi], max_mip_level), uv, level)   # :355-359
        ctr = xyz[sl].mean(dim=-1)
        td = F.normalize(ctr - tar_c[bi][None], dim=-1)                                       # :362-367
        sd = F.normalize(ctr[None] - src_c[bi][:, None], dim=-1)
        out[:, sl, 3 * bb + Cf:3 * bb + Cf + 3] = F.normalize(td[None] - sd, dim=-1)
        out[:, sl, 3 * bb + Cf + 3] = (td[None] * sd).sum(-1)
        start += n
    return out, vox


def nerf_mlp(w: Dict[str, torch.Tensor], vox: torch.Tensor, x: torch.Tensor, feat_dim: int = 16, viewdir_agg: bool = True):
    """nerf.py:58-115."""
    lin = lambda name, t: F.linear(t, w[name + ".weight"], w[name + ".bias"])
    f = x[..., -(feat_dim + 7):]                                                              # :98
    feat, dirs = f[..., :feat_dim + 3], f[..., feat_dim + 3:]
    g = feat + F.relu(lin("view_fc.0", dirs)) if viewdir_agg else feat                        # :69-71
    var, mean = torch.var_mean(g, dim=0, keepdim=True)                                        # :73
    G = F.relu(lin("global_fc.0", torch.cat((g, var.expand_as(g), mean.expand_as(g)), dim=-1)))   # :77-78
    a = torch.softmax(F.relu(lin("agg_w_fc.0", G)), dim=0)                                    # :79
    im = F.relu(lin("fc.0", (G * a).sum(0)))                                                  # :80-82
    h = torch.cat((vox, im), dim=-1)
    xh = F.relu(lin("lr0.0", h))                                                              # :100-101
    sigma = F.softplus(lin("sigma.0", xh))[:, 0]                                              # :102
    wi = torch.cat((xh[None].expand(x.shape[0], -1, -1), h[None].expand(x.shape[0], -1, -1), f), dim=-1)
    wv = torch.softmax(F.relu(lin("weight.2", F.relu(lin("weight.0", wi)))), dim=0)           # :106-109
    out = (x[..., :-4] * wv).sum(0)                                                           # :110
    return sigma, torch.cat((out, F.relu(lin("feat_head.0", xh))), dim=-1)                    # :111-113


def render_bundles(w, rfd, vox, z, indices, n_bundles, inv_depth, feat_dim=16, viewdir_agg=True):
    """network.py:54-91 with utils.py:19-43 (alpha, exclusive transmittance per bundle, normalisation) and utils.py:88-121."""
    sigma, feat = nerf_mlp(w, vox, rfd, feat_dim, viewdir_agg)
    alpha = 1.0 - torch.exp(-sigma)                                                           # utils.py:34
    cnt = torch.bincount(indices, minlength=n_bundles)
    first = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(indices.numel()) - first[indices]
    S = int(cnt.max()) if cnt.numel() else 0
    tab = torch.ones(n_bundles, S + 1)
    tab[indices, k + 1] = 1.0 - alpha                                                         # padded per-bundle table
    T = torch.cumprod(tab, dim=1)[indices, k]                                                 # exclusive product   (nerfacc, utils.py:35)
    wgt = alpha * T
    den = torch.zeros(n_bundles).index_add_(0, indices, wgt).clamp(min=1e-6)                  # utils.py:38-41
    wgt = wgt / den[indices]
    zz = 1.0 / z if inv_depth else z                                                          # network.py:83-84
    vals = torch.cat((feat, zz[:, None], torch.ones_like(zz)[:, None]), dim=1) * wgt[:, None]   # utils.py:109-110
    acc = torch.zeros(n_bundles, vals.shape[1]).index_add_(0, indices, vals)
    depth = acc[:, -2]
    return acc[:, :-2], (1.0 / depth if inv_depth else depth), acc[:, -1]                     # network.py:88-89


def hot_path(frame, weights, *, bundle_size=2, max_num_samples=3, is_adaptive=True, inv_depth=False, global_num_depth=64,
             max_mipmap_level=3, feat_dim=16, viewdir_agg=True):
    """build_rays -> sample -> encode -> render_bundles (network.py:145-169) on one frame dict (numpy or torch); returns torch
    tensors (bundle_feat (N_b, Q), depth, opacity)."""
    with torch.no_grad():
        fr = {k: _t(v) for k, v in frame.items()}
        w = {k: _t(v) for k, v in weights.items()}
        Ho, Wo = fr["src_images"].shape[-2:]
        rays = build_rays(fr["tar_ext"], fr["tar_int"], Ho, Wo)
        smp = sample_bundles(rays, fr["depth_range"], fr["vol_range"], fr["near_far"][:, 0], fr["near_far"][:, 1], bundle_size,
                             max_num_samples, global_num_depth, inv_depth, is_adaptive)
        rfd, vox = encode(fr["src_images"], fr["img_feat"], fr["feat_volume"], smp, fr["src_exts"], fr["src_ints"], fr["tar_ext"],
                          bundle_size, Ho, Wo, max_mipmap_level)
        return render_bundles(w, rfd, vox, smp["z_vals"], smp["indices"], smp["samples_per_bundle"].shape[0], inv_depth, feat_dim, viewdir_agg)
