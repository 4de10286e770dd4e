"""Generate the golden fixtures in this directory by running the reference's own code.

Runs ONLY in the authoring container (needs /root/reference, CPU torch); the `.npz` files it
writes are committed and are what travels to the GPU box.  Usage:

    python tests/golden/make_golden.py

What is reference-pinned and what is not
----------------------------------------
* `nerf.py` imports with no help; `F3_*` are pure reference outputs.
* `bundle_sampler.py` and `utils.py` import two third-party CUDA packages that do not exist
  in this image (`nvdiffrast`, `nerfacc`).  To load the two files at all, placeholder modules
  of those names are registered; `build_rays` / `sample` never call into them, so `F1_*`,
  `F2_*` are pure reference outputs.  `encode` calls `nvdiffrast.torch.texture` once
  (bundle_sampler.py:355-359) and `utils.py` calls `nerfacc.volrend` twice (:35, :110); the
  placeholders route exactly those three calls to the oracle's restatements, so
  - `F4_*`: rgbs (ch 0..11), dir (ch 31..34), vox_feat and the mip *levels* handed to the
    texture call are reference arithmetic; the 19 mip-fetched channels (12..30) are
    "parity unpinned" (they show only that reference + restatement compose as expected);
  - `F5_*`, `F6_*`: depend on the nerfacc restatement -> "parity unpinned" for the
    composite, pinned for everything upstream of it.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/networks/gdb_nerf"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import gdb_oracle as oracle  # noqa: E402
from gdb_nerf_amd import synthetic  # noqa: E402

CAPTURE = {}


def _texture(tex, uv, mip_level_bias=None, boundary_mode="clamp", max_mip_level=None, **kw):
    assert boundary_mode == "clamp" and not kw
    pyr = oracle.build_mips(tex.numpy(), max_mip_level)
    CAPTURE.setdefault("tex_levels", []).append(mip_level_bias.squeeze(-1).numpy().copy())
    CAPTURE.setdefault("tex_uv", []).append(uv.squeeze(2).numpy().copy())
    out = oracle.texture_mip(pyr, uv.squeeze(2).numpy(), mip_level_bias.squeeze(-1).numpy())
    return torch.from_numpy(out).unsqueeze(2)


def _render_weight_from_alpha(alpha, ray_indices=None, n_rays=None):
    w = oracle.weights_from_alpha(alpha.numpy(), ray_indices.numpy(), n_rays)
    return torch.from_numpy(w), None


def _accumulate_along_rays(weights, values, ray_indices, n_rays):
    acc = np.zeros((n_rays, values.shape[1]), dtype=np.float32)
    np.add.at(acc, ray_indices.numpy(), (values.numpy() * weights.numpy()[:, None]).astype(np.float32))
    return torch.from_numpy(acc)


def _placeholders():
    nv = types.ModuleType("nvdiffrast")
    nvt = types.ModuleType("nvdiffrast.torch")
    nvt.texture = _texture
    nv.torch = nvt
    na = types.ModuleType("nerfacc")
    nav = types.ModuleType("nerfacc.volrend")
    nav.render_weight_from_alpha = _render_weight_from_alpha
    nav.accumulate_along_rays = _accumulate_along_rays
    na.volrend = nav
    sys.modules.update({"nvdiffrast": nv, "nvdiffrast.torch": nvt, "nerfacc": na, "nerfacc.volrend": nav})


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_sampler(ref_bs, frame, b, S_max, adaptive, inv_depth, gnd=64, max_mip=3):
    Ho, Wo = frame["src_images"].shape[-2:]
    s = ref_bs.BundleSampler(gnd, max_mip)
    nf = t(frame["near_far"])
    s.build_rays(t(frame["tar_ext"]), t(frame["tar_int"]), (Ho, Wo), nf[:, 0], nf[:, 1])
    out = s.sample(t(frame["depth_range"]), t(frame["vol_range"]), b, S_max, inv_depth, adaptive)
    return s, out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref_nerf = _load("nerf")
    _placeholders()
    ref_bs = _load("bundle_sampler")
    ref_utils = _load("utils")

    # ---- F1 build_rays, F2 sample (fixed / adaptive, depth / disparity) ---------------
    frame = synthetic.make_frame(32, 48, V=3, B=2, seed=1)
    s, _ = run_sampler(ref_bs, frame, 2, 3, True, False)
    save("F1_build_rays", tar_ext=frame["tar_ext"], tar_int=frame["tar_int"], Ho=32, Wo=48,
         rays_o=s.rays_o.numpy(), z_axis=s.z_axis.numpy(), rays_d=s.rays_d.numpy(),
         uv=s.uv.numpy(), tar_pixel_radius=s.tar_pixel_radius.numpy())
    f2 = dict(tar_ext=frame["tar_ext"], tar_int=frame["tar_int"], near_far=frame["near_far"],
              depth_range=frame["depth_range"], vol_range=frame["vol_range"], Ho=32, Wo=48)
    for tag, (S_max, adaptive, inv) in {"fix6": (6, False, False), "ada3": (3, True, False),
                                        "ada6inv": (6, True, True), "fix2inv": (2, False, True)}.items():
        _, o = run_sampler(ref_bs, frame, 2, S_max, adaptive, inv)
        names = ("rays_xyz", "uvd", "z_vals", "ball_radii", "indices", "samples_per_batch", "samples_per_bundle")
        for n, v in zip(names, o):
            f2[f"{tag}_{n}"] = v.numpy()
        f2[f"{tag}_spb_dtype"] = str(o[6].dtype)
    save("F2_sample", **f2)
    # bundle size 4 (the "4*4" variant the YAML comments mention)
    frame4 = synthetic.make_frame(32, 64, V=2, B=1, bundle_size=4, seed=2)
    _, o = run_sampler(ref_bs, frame4, 4, 3, True, False)
    save("F2_sample_b4", tar_ext=frame4["tar_ext"], tar_int=frame4["tar_int"], near_far=frame4["near_far"],
         depth_range=frame4["depth_range"], vol_range=frame4["vol_range"], Ho=32, Wo=64,
         **{n: v.numpy() for n, v in zip(("rays_xyz", "uvd", "z_vals", "ball_radii", "indices",
                                          "samples_per_batch", "samples_per_bundle"), o)})

    # ---- F3 NeRF.forward, V in {2,3,5}, reference default init -------------------------
    for V in (2, 3, 5):
        torch.manual_seed(10 + V)
        net = ref_nerf.NeRF(64, 16, 8, True).eval()
        for p in net.parameters():  # default init leaves biases tiny; make them count
            if p.ndim == 1:
                torch.nn.init.uniform_(p, -0.3, 0.3)
        N = 257
        vox = torch.randn(N, 8)
        x = torch.randn(V, N, 35)
        with torch.no_grad():
            sigma, feat = net(vox, x)
        save(f"F3_nerf_V{V}", vox_feat=vox.numpy(), rgbs_feat_dir=x.numpy(), sigma=sigma.numpy(),
             feat=feat.numpy(), **{"w." + k: v.numpy() for k, v in net.state_dict().items()})
    torch.manual_seed(20)
    net = ref_nerf.NeRF(64, 16, 8, False).eval()
    vox, x = torch.randn(64, 8), torch.randn(3, 64, 35)
    with torch.no_grad():
        sigma, feat = net(vox, x)
    save("F3_nerf_noviewdir", vox_feat=vox.numpy(), rgbs_feat_dir=x.numpy(), sigma=sigma.numpy(),
         feat=feat.numpy(), **{"w." + k: v.numpy() for k, v in net.state_dict().items()})

    # ---- F4 encode, F5 render_bundles, F6 whole hot path -------------------------------
    for tag, kw, (S_max, adaptive, inv) in (
            ("dtu", dict(Ho=32, Wo=48, V=3, B=2, seed=3), (3, True, False)),
            ("nerfinv", dict(Ho=32, Wo=32, V=2, B=1, seed=4, scene="nerf"), (4, False, True)),
            ("mips", dict(Ho=64, Wo=64, V=4, B=1, seed=5, src_focal_scale=(1.0, 2.2, 5.0, 20.0)), (2, True, False))):
        frame = synthetic.make_frame(**kw)
        Ho, Wo = kw["Ho"], kw["Wo"]
        CAPTURE.clear()
        s, o = run_sampler(ref_bs, frame, 2, S_max, adaptive, inv)
        rays_xyz, uvd, z_vals, ball, idx, per_batch, spb = o
        rfd, vox = s.encode(t(frame["src_images"]), t(frame["img_feat"]), t(frame["feat_volume"]), rays_xyz,
                            uvd, ball, t(frame["src_exts"]), t(frame["src_ints"]), t(frame["tar_ext"]), per_batch)
        levels = np.concatenate(CAPTURE["tex_levels"], axis=1)
        tex_uv = np.concatenate(CAPTURE["tex_uv"], axis=1)
        save(f"F4_encode_{tag}", **frame, S_max=S_max, adaptive=adaptive, inv_depth=inv,
             rays_xyz=rays_xyz.numpy(), uvd=uvd.numpy(), ball_radii=ball.numpy(),
             samples_per_batch=per_batch.numpy(), rgbs_feat_dir=rfd.numpy(), vox_feat=vox.numpy(),
             tex_levels=levels, tex_uv=tex_uv)

        torch.manual_seed(30)
        net = ref_nerf.NeRF(64, 16, 8, True).eval()
        for p in net.parameters():
            if p.ndim == 1:
                torch.nn.init.uniform_(p, -0.3, 0.3)
        with torch.no_grad():
            sigma, feat = net(vox, rfd)
            nb = spb.shape[0]
            wts, inv_idx = ref_utils.render_weight_from_density(sigma, idx, nb)
            zz = 1.0 / z_vals if inv else z_vals
            bf, depth, opac = ref_utils.accumulate_value_along_rays(feat, zz, wts, idx, nb, inv_idx)
            if inv:
                depth = 1.0 / depth
        save(f"F5_render_{tag}", sigma=sigma.numpy(), feat=feat.numpy(), z_vals=z_vals.numpy(),
             indices=idx.numpy(), n_bundles=nb, inv_depth=inv, weights=wts.numpy(),
             bundle_feat=bf.numpy(), depth=depth.numpy(), opacity=opac.numpy(),
             **{"w." + k: v.numpy() for k, v in net.state_dict().items()})
        save(f"F6_hotpath_{tag}", **frame, S_max=S_max, adaptive=adaptive, inv_depth=inv,
             bundle_feat=bf.numpy(), depth=depth.numpy(), opacity=opac.numpy(),
             samples_per_bundle=spb.numpy(),
             **{"w." + k: v.numpy() for k, v in net.state_dict().items()})


if __name__ == "__main__":
    main()
