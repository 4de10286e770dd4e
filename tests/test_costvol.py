""""Next" rows N2 / N4: cost-volume build and depth regression.  CPU: the oracle against
fixture F8 (the reference's own functions, tests/golden/make_golden_costvol.py — fully pinned, no third-party
op involved).  GPU: the HIP kernels through the C ABI against F8 and the oracle."""
import numpy as np
import pytest
import torch

import gdb_oracle as oracle
from conftest import load_golden, max_abs


@pytest.fixture(scope="module")
def f8():
    return load_golden("F8_costvol")


def _case(fx, tag):
    return {k[len(tag) + 1:]: v for k, v in fx.items() if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", ["coarse", "fine"])
def test_oracle_cost_volume_matches_reference(f8, tag):
    c = _case(f8, tag)
    vol = oracle.build_feature_volume(c["src_feat"], c["src_exts"], c["src_ints"], c["tar_ext"], c["tar_int"], c["depth_values"], bool(c["inv_depth"]))
    assert vol.shape == c["volume"].shape
    assert max_abs(vol, c["volume"]) <= 1e-4  # values up to ~6
    d, ci = oracle.depth_regression(c["depth_values"], c["prob"], 1.0, bool(c["inv_depth"]))
    assert max_abs(d, c["depth"]) <= 1e-5 * float(np.abs(c["depth"]).max())
    assert max_abs(ci, c["ci"]) <= 1e-5 * float(np.abs(c["ci"]).max())


def test_oracle_depth_values():
    nf = np.array([[425.0, 905.0]], np.float32)[..., None, None]
    lin = oracle.get_depth_values(nf, 8, False)[0, :, 0, 0]
    assert np.allclose(lin, np.linspace(425, 905, 8), rtol=1e-6)
    inv = oracle.get_depth_values(nf, 8, True)[0, :, 0, 0]
    assert np.allclose(inv, np.linspace(1 / 425, 1 / 905, 8), rtol=1e-6)


def test_bilinear_zeros_padding():
    img = np.arange(12, dtype=np.float32).reshape(1, 3, 4)
    g = lambda px, size: 2 * (px + 0.5) / size - 1  # pixel centre -> normalised
    out = oracle.bilinear_zeros(img, np.array([g(1, 4), g(-1, 4), g(3.5, 4), g(1e9, 4)], np.float32),
                                np.array([g(1, 3), g(1, 3), g(2, 3), g(1, 3)], np.float32))
    assert out[0, 0] == 5 and out[0, 1] == 0 and abs(out[0, 2] - 5.5) < 1e-6 and out[0, 3] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["coarse", "fine"])
def test_hip_cost_volume_matches_reference(f8, tag):
    from gdb_nerf_amd import costvol
    c = _case(f8, tag)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(c[k])).cuda()
    vol = costvol.build_feature_volume(t("src_feat"), t("src_exts"), t("src_ints"), t("tar_ext"), t("tar_int"), t("depth_values"), bool(c["inv_depth"]))
    e = max_abs(vol.cpu().numpy(), c["volume"])
    print(f"cost volume {tag}: max abs err {e:.3e}")
    assert e <= 1e-4
    # the default takes the channel-pair re-layout of the source maps (gdb_build_feature_volume_ws); the planar form: bit-identical
    planar = costvol.build_feature_volume(t("src_feat"), t("src_exts"), t("src_ints"), t("tar_ext"), t("tar_int"), t("depth_values"), bool(c["inv_depth"]),
                                          pair_layout=False)
    assert torch.equal(vol, planar)
    d, ci = costvol.depth_regression(t("depth_values"), t("prob"), 1.0, bool(c["inv_depth"]))
    assert max_abs(d.cpu().numpy(), c["depth"]) <= 1e-5 * float(np.abs(c["depth"]).max())
    assert max_abs(ci.cpu().numpy(), c["ci"]) <= 1e-5 * float(np.abs(c["ci"]).max())


@pytest.mark.gpu
def test_hip_cost_volume_edges_vs_oracle():
    """Views that look away / far-off planes: most taps fall outside the source image (zeros padding), points
    behind a camera hit the z clamp; ragged width; 5 views."""
    from gdb_nerf_amd import costvol, synthetic
    rng = np.random.default_rng(3)
    fr = synthetic.make_frame(48, 72, V=5, B=1, seed=8, src_focal_scale=(1.0, 4.0, 0.3, 9.0, 1.0))
    fr["src_exts"][0, 4, :3, 3] += np.array([400.0, -250.0, 900.0], np.float32)  # a camera far off to the side / behind
    feat = rng.standard_normal((1, 5, 8, 24, 36)).astype(np.float32)
    Ks, Kt = fr["src_ints"].copy(), fr["tar_int"].copy()
    Ks[..., :2, :] *= 0.5; Kt[:, :2, :] *= 0.25
    dv = oracle.get_depth_values(np.broadcast_to(np.array([30.0, 2000.0], np.float32)[None, :, None, None], (1, 2, 12, 18)), 6, True)
    want = oracle.build_feature_volume(feat, fr["src_exts"], Ks, fr["tar_ext"], Kt, dv, True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = costvol.build_feature_volume(t(feat), t(fr["src_exts"]), t(Ks), t(fr["tar_ext"]), t(Kt), t(dv), True)
    assert (want == 0).mean() < 0.9 and np.isfinite(want).all()
    assert max_abs(got.cpu().numpy(), want) <= 1e-4
    with pytest.raises(ValueError):
        costvol.build_feature_volume(t(feat), t(fr["src_exts"]), t(Ks), t(fr["tar_ext"])[:, :3], t(Kt), t(dv), True)
