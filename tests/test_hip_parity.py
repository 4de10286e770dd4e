"""Parity of the HIP hot path (through the C ABI) with the oracle and the golden fixtures.
Runs on a real MI355X: `pytest -m gpu`.

Tolerances (float32, absolute unless stated):
  * integer work (sample counts, compaction order, indices) — exact;
  * geometry in world units (mm, |x| ~ 1e3): 2e-3 abs (2 ulp of the magnitude involved);
  * unit-scale values fetched by interpolation / MLP outputs, fp32 kernels: 2e-4 (the source
    of difference is summation order and the 1-ulp freedom of the camera inverses, amplified
    by white-noise feature gradients);
  * fused kernel, GDB_PREC_F16 (f16 MFMA operands): 2e-3 abs on bundle_feat, PSNR delta <= 0.05 dB (north_star);
  * fused kernel, GDB_PREC_F32 / F32X (fp32 MFMA, the reference's precision; split-f16 pairs): 3 x the largest error observed on
    MI355X per frame class (profiles/r05/parity_observed.txt) - 5e-5 on the fixture-size frames, 3e-4 at c2, 5e-4 on the larger
    frames (fp32 coordinate noise on white-noise features: FUSED_TOL_F32) and 2e-5 on the smooth-feature c2 frame, where that noise
    is out of the way and the bound sees the arithmetic; PSNR delta <= 0.05 dB.
"""
import numpy as np
import pytest
import torch

import gdb_oracle as oracle
from conftest import FRAME_KEYS, frame_of, load_golden, max_abs, nerf_weights_of
import ctypes as C

from gdb_nerf_amd import _lib, synthetic
from gdb_nerf_amd.engine import HotPathEngine

pytestmark = pytest.mark.gpu


def dev_frame(frame):
    return {k: torch.from_numpy(np.ascontiguousarray(frame[k])).cuda() for k in FRAME_KEYS}


def npy(t):
    return t.detach().cpu().numpy()


@pytest.fixture(params=[(1, 0), (2, 0), (3, 0), (4, 0), (1, 1), (2, 1), (3, 1), (4, 1), (1, 2), (2, 2), (3, 2), (4, 2)],
                ids=["slot-waves-f16", "segment-wave-f16", "dense-f16", "flat-f16", "slot-waves-f32", "segment-wave-f32", "dense-f32", "flat-f32",
                     "slot-waves-f32x", "segment-wave-f32x", "dense-f32x", "flat-f32x"])
def mode(request):
    """(schedule, precision): the four work decompositions of the fused kernel x the three MLP precisions
    (include/gdb_nerf_hip.h GDB_SCHED_*, GDB_PREC_*).  Per-engine settings, passed on every call of the C ABI."""
    return request.param


def fused_tol(mode, size="small"):
    """size: "small" = the fixture-size frames (<= 128 px), "c2" = 512x640, "large" = c3 .. c5."""
    if mode[1] == 0:
        return FUSED_TOL
    return {"small": FUSED_TOL_F32_SMALL, "c2": FUSED_TOL_F32_C2, "large": FUSED_TOL_F32}[size]


def engine_for(frame, weights=None, mode=None, **cfg):
    eng = HotPathEngine(**cfg)
    if mode is not None:
        eng.set_schedule(mode[0])
        eng.precision = mode[1]
    if weights is not None:
        eng.load_weights(weights)
    eng.prepare(dev_frame(frame))
    return eng


def oracle_rays(frame):
    Ho, Wo = frame["src_images"].shape[-2:]
    return oracle.build_rays(frame["tar_ext"], frame["tar_int"], Ho, Wo)


def test_build_rays_vs_golden():
    fx = load_golden("F1_build_rays")
    frame = synthetic.make_frame(32, 48, V=3, B=2, seed=1)
    assert np.array_equal(frame["tar_ext"], fx["tar_ext"])
    eng = engine_for(frame)
    r = eng.build_rays()
    for k in ("rays_o", "z_axis", "rays_d", "uv", "tar_pixel_radius"):
        assert max_abs(npy(r[k]), fx[k]) <= 2e-6, k


@pytest.mark.parametrize("tag,S,adaptive,inv", [("fix6", 6, False, False), ("ada3", 3, True, False),
                                                ("ada6inv", 6, True, True), ("fix2inv", 2, False, True)])
def test_sample_vs_golden(tag, S, adaptive, inv):
    fx = load_golden("F2_sample")
    frame = synthetic.make_frame(32, 48, V=3, B=2, seed=1)
    assert np.array_equal(frame["depth_range"], fx["depth_range"])
    eng = engine_for(frame, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    s = eng.sample()
    n = int(s["total"].item())
    assert n == fx[tag + "_indices"].shape[0]
    assert np.array_equal(npy(s["samples_per_bundle"]), fx[tag + "_samples_per_bundle"].astype(np.int32))
    assert np.array_equal(npy(s["indices"][:n]), fx[tag + "_indices"])
    assert np.array_equal(npy(s["samples_per_batch"]), fx[tag + "_samples_per_batch"].astype(np.int64))
    assert max_abs(npy(s["rays_xyz"][:n]), fx[tag + "_rays_xyz"]) <= 2e-3
    assert max_abs(npy(s["uvd"][:n]), fx[tag + "_uvd"]) <= 2e-6
    assert max_abs(npy(s["z_vals"][:n]), fx[tag + "_z_vals"]) <= 2e-4
    assert max_abs(npy(s["ball_radii"][:n]), fx[tag + "_ball_radii"]) <= 1e-5 * float(np.abs(fx[tag + "_ball_radii"]).max())


def test_sample_bundle_size_4_vs_golden():
    fx = load_golden("F2_sample_b4")
    frame = synthetic.make_frame(32, 64, V=2, B=1, bundle_size=4, seed=2)
    eng = engine_for(frame, bundle_size=4, max_num_samples=3, is_adaptive=True)
    s = eng.sample()
    n = int(s["total"].item())
    assert np.array_equal(npy(s["indices"][:n]), fx["indices"])
    assert max_abs(npy(s["rays_xyz"][:n]), fx["rays_xyz"]) <= 2e-3
    assert max_abs(npy(s["ball_radii"][:n]), fx["ball_radii"]) <= 1e-5


@pytest.mark.parametrize("Ho,Wo,V,B,levels", [(64, 80, 3, 1, 3), (96, 72, 2, 2, 3), (40, 104, 2, 1, 3), (512, 640, 3, 1, 3),
                                               (64, 80, 2, 1, 1), (64, 80, 2, 1, 0)])
def test_feature_pyramid_is_bit_exact(Ho, Wo, V, B, levels):
    """k_prepare's pyramid (layout change + 2x2 box averages) against the oracle's mip construction: exact.
    (40, 104): W = 52 -> level widths 26, 13: ragged tiles and a chain that stops at the first odd extent."""
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, seed=17)
    eng = engine_for(frame, max_mipmap_level=levels)
    got = eng.feature_pyramid()
    for bi in range(B):
        want = oracle.build_mips(np.transpose(frame["img_feat"][bi], (0, 2, 3, 1)), levels)
        assert len(got) == len(want)
        for l, w in enumerate(want):
            assert np.array_equal(npy(got[l][bi]).view(np.uint32), w.view(np.uint32)), (bi, l)


@pytest.mark.parametrize("Ho,Wo,V,B,levels", [(64, 80, 3, 1, 3), (96, 72, 2, 2, 3), (40, 104, 2, 1, 3), (512, 640, 3, 1, 3), (12, 12, 2, 2, 3),
                                               (64, 80, 2, 1, 0)])
def test_half_precision_pyramid_is_the_fp32_one_rounded_once(Ho, Wo, V, B, levels):
    """The copy of the pyramid GDB_PREC_F16 gathers from (gdb_prepare_ex GDB_PREP_PYR16; include/gdb_nerf_hip.h): every level equals
    the oracle's fp32 mip level rounded to nearest half ONCE (the box filter runs in fp32), bit for bit - written by k_prepare itself
    for an engine whose precision is f16, and by the conversion launch an f16 render makes when prepare was not asked for it.
    (12, 12): a 6x6 map whose 3x3 level 1 gives the (batch, view) blocks a stride that is not a multiple of 16 bytes."""
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, seed=17)
    want = [[oracle.build_mips(np.transpose(frame["img_feat"][bi], (0, 2, 3, 1)), levels)] for bi in range(B)]
    eng = HotPathEngine(max_mipmap_level=levels)
    eng.precision = 0
    eng.load_weights(synthetic.make_nerf_weights(seed=1))
    eng.prepare(dev_frame(frame))                      # k_prepare writes both pyramids
    assert eng._pyr16_ready
    a = eng.feature_pyramid16()
    eng2 = engine_for(frame, synthetic.make_nerf_weights(seed=1), max_mipmap_level=levels)   # fp32 engine: no half-precision copy yet
    assert not eng2._pyr16_ready
    r16 = eng2.render(precision=0)[0].clone()          # ... the f16 render converts the fp32 pyramid itself
    b = eng2.feature_pyramid16()
    for bi in range(B):
        w = want[bi][0]
        assert len(a) == len(w) == len(b)
        for l, wl in enumerate(w):
            h16 = wl.astype(np.float16)
            assert np.array_equal(npy(a[l][bi]).view(np.uint16), h16.view(np.uint16)), (bi, l, "k_prepare")
            assert np.array_equal(npy(b[l][bi]).view(np.uint16), h16.view(np.uint16)), (bi, l, "k_pyr16")
    assert torch.equal(eng.render()[0], r16)           # both routes render the same image, bit for bit


def test_f16_engine_prepares_the_half_precision_pyramid_alone():
    """gdb_prepare_ex(GDB_PREP_PYR16 | GDB_PREP_PYR16_ONLY), what a PREC_F16 engine's prepare() asks for: the fp32 pyramid in the
    workspace is NOT written (a sentinel survives), the half-precision copy and the f16 render are bit-identical to the engine that
    writes both, and whatever needs the fp32 pyramid later on the same frame (an f32 render, feature_pyramid(), the operator mirror
    encode() inside render_unfused()) gets it through a second, full prepare.  The flag alone is rejected."""
    frame = synthetic.make_frame(96, 128, V=3, B=2, seed=4)
    w = synthetic.make_nerf_weights(seed=2)
    both = HotPathEngine(is_adaptive=True); both.precision = 0; both.f16_only_prepare = False; both.load_weights(w)
    only = HotPathEngine(is_adaptive=True); only.precision = 0; only.load_weights(w)
    fr = dev_frame(frame)
    both.prepare(fr); only.prepare(fr)                                   # (sizes the workspace)
    lay = (C.c_size_t * 7)()
    _lib.check(only.lib.gdb_pyramid_layout(C.byref(only.cfg), C.byref(only._frame), lay))
    n32 = 4 * int(lay[1]) * only._frame.B * only._frame.V
    only._ws[int(lay[0]):int(lay[0]) + n32].fill_(0xA5)                   # sentinel over the whole fp32 pyramid
    only.prepare(fr)
    assert only._pyr16_ready and not only._pyr32_ready and both._pyr32_ready
    assert bool((only._ws[int(lay[0]):int(lay[0]) + n32] == 0xA5).all())  # untouched
    for a, b in zip(only.feature_pyramid16(), both.feature_pyramid16()):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert torch.equal(only.render()[0], both.render()[0])
    assert not only._pyr32_ready                                         # an f16 render does not need it
    r32 = only.render(precision=1)[0].clone()                            # ... an f32 render does: full prepare first
    assert only._pyr32_ready
    assert torch.equal(r32, both.render(precision=1)[0])
    for a, b in zip(only.feature_pyramid(), both.feature_pyramid()):
        assert torch.equal(a, b)
    only.prepare(fr)
    assert not only._pyr32_ready
    assert torch.equal(only.render_unfused()[0], both.render_unfused()[0]) and only._pyr32_ready
    rc = only.lib.gdb_prepare_ex(C.byref(only.cfg), C.byref(only._frame), None, _lib.PREP_PYR16_ONLY, only._ws.data_ptr(), only._ws.numel(), None)
    assert rc != 0 and "GDB_PREP_PYR16" in only.lib.gdb_last_error().decode()


@pytest.mark.parametrize("tag", ["dtu", "nerfinv", "mips"])
def test_encode_vs_golden(tag):
    """Inputs are the reference's own sample arrays; outputs against the reference's encode."""
    fx = load_golden("F4_encode_" + tag)
    S, adaptive, inv = int(fx["S_max"]), bool(fx["adaptive"]), bool(fx["inv_depth"])
    eng = engine_for(frame_of(fx), max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    n = fx["rays_xyz"].shape[0]
    c = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a if dt is None else a.astype(dt))).cuda()
    rfd, vox = eng.encode(c(fx["rays_xyz"]), c(fx["uvd"]), c(fx["ball_radii"]), c(fx["samples_per_batch"], np.int64),
                          torch.tensor([n], dtype=torch.int64, device="cuda"))
    ref = fx["rgbs_feat_dir"]
    assert max_abs(npy(rfd)[..., :12], ref[..., :12]) <= 2e-5   # per-ray RGB
    assert max_abs(npy(vox), fx["vox_feat"]) <= 2e-5            # voxel feature
    assert max_abs(npy(rfd)[..., 31:], ref[..., 31:]) <= 2e-5   # view-direction code
    assert max_abs(npy(rfd)[..., 12:31], ref[..., 12:31]) <= 2e-4  # mip fetch


@pytest.mark.parametrize("name,viewdir", [("F3_nerf_V2", True), ("F3_nerf_V3", True), ("F3_nerf_V5", True),
                                          ("F3_nerf_noviewdir", False)])
def test_mlp_vs_golden(name, viewdir):
    fx = load_golden(name)
    eng = HotPathEngine(viewdir_agg=viewdir)
    eng.load_weights(nerf_weights_of(fx))
    sigma, feat = eng.mlp(torch.from_numpy(fx["vox_feat"]).cuda(), torch.from_numpy(fx["rgbs_feat_dir"]).cuda())
    assert max_abs(npy(sigma), fx["sigma"]) <= 1e-5
    assert max_abs(npy(feat), fx["feat"]) <= 1e-5


@pytest.mark.parametrize("tag", ["dtu", "nerfinv", "mips"])
def test_composite_vs_golden(tag):
    fx = load_golden("F5_render_" + tag)
    eng = HotPathEngine(inv_depth=bool(fx["inv_depth"]))
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    w, bf, depth, opac = eng.composite(c(fx["sigma"]), c(fx["feat"]), c(fx["z_vals"]), c(fx["indices"]), int(fx["n_bundles"]))
    assert max_abs(npy(w), fx["weights"]) <= 1e-6
    assert max_abs(npy(bf), fx["bundle_feat"]) <= 2e-6
    assert max_abs(npy(depth), fx["depth"]) <= 2e-6 * float(np.abs(fx["depth"]).max())
    assert max_abs(npy(opac), fx["opacity"]) <= 1e-6


def test_composite_empty_and_transparent_bundles():
    eng = HotPathEngine()
    sigma = torch.tensor([0.0, 0.0, 5.0, 1.0], device="cuda")
    idx = torch.tensor([0, 0, 2, 2], dtype=torch.int64, device="cuda")
    feat = torch.ones((4, 3), device="cuda")
    w, bf, depth, opac = eng.composite(sigma, feat, torch.ones(4, device="cuda"), idx, 4)
    ow = oracle.render_weights(npy(sigma), npy(idx), 4)
    assert max_abs(npy(w), ow) <= 1e-7
    assert np.all(npy(bf)[[0, 1, 3]] == 0) and np.all(npy(opac)[[0, 1, 3]] == 0)
    assert abs(float(opac[2]) - 1) < 1e-6


@pytest.mark.parametrize("tag", ["dtu", "nerfinv", "mips"])
def test_unfused_hot_path_vs_golden(tag):
    fx = load_golden("F6_hotpath_" + tag)
    eng = engine_for(frame_of(fx), nerf_weights_of(fx), max_num_samples=int(fx["S_max"]),
                     is_adaptive=bool(fx["adaptive"]), inv_depth=bool(fx["inv_depth"]))
    bf, depth, opac = eng.render_unfused()
    assert max_abs(npy(bf), fx["bundle_feat"]) <= 2e-4
    assert max_abs(npy(depth), fx["depth"]) <= 1e-5 * float(np.abs(fx["depth"]).max())
    assert max_abs(npy(opac), fx["opacity"]) <= 2e-6


@pytest.mark.parametrize("Ho,Wo,V,B,S,adaptive,scene", [(64, 80, 3, 1, 3, True, "dtu"), (64, 80, 3, 1, 6, False, "dtu"),
                                                         (96, 64, 4, 2, 6, True, "nerf"), (48, 80, 2, 1, 3, True, "llff")])
def test_unfused_hot_path_vs_oracle(Ho, Wo, V, B, S, adaptive, scene):
    """c1-size frames (BASELINE.json configs[0]) and friends against the oracle on the same seeded inputs."""
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, scene=scene, seed=11, src_focal_scale=(1.0, 1.7, 3.1))
    w = synthetic.make_nerf_weights(seed=3)
    obf, od, oo, aux = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=adaptive, return_intermediates=True)
    eng = engine_for(frame, w, max_num_samples=S, is_adaptive=adaptive)
    s = eng.sample()
    assert np.array_equal(npy(s["samples_per_bundle"]), aux["samples"]["samples_per_bundle"])
    bf, depth, opac = eng.render_unfused()
    assert max_abs(npy(bf), obf) <= 2e-4
    assert max_abs(npy(depth), od) <= 1e-5 * float(np.abs(od).max())
    assert max_abs(npy(opac), oo) <= 2e-6


def test_engine_rejects_bad_shapes():
    frame = dev_frame(synthetic.make_frame(32, 48))
    eng = HotPathEngine()
    with pytest.raises(ValueError, match="prepare"):
        eng.sample()
    bad = dict(frame); bad["img_feat"] = frame["img_feat"][:, :, :18].contiguous()
    with pytest.raises(ValueError, match="img_feat"):
        eng.prepare(bad)
    bad = dict(frame); bad["src_images"] = frame["src_images"].double()
    with pytest.raises(ValueError, match="float32"):
        eng.prepare(bad)
    with pytest.raises(ValueError, match="power of 2"):
        HotPathEngine(bundle_size=3)


# ---------------------------------------------------------------------------------------------
# fused production kernel (f16 MFMA MLP, f32 accumulate, f32 fetch / composite)
# ---------------------------------------------------------------------------------------------
FUSED_TOL = 2e-3      # GDB_PREC_F16: abs, on O(1) bundle features; observed ~2e-4 max / 1.4e-5 rms (printed by the tests)
# GDB_PREC_F32: two fp32 implementations of the same path differ through their pixel coordinates (reciprocal-based vs IEEE
# division, one pre-multiplied 3x4 projection vs two matrix products: ~1 ulp of a coordinate of several hundred pixels = 1e-4 px),
# which the white-noise test features (gradient O(1) per pixel) turn into ~1e-4 of feature error: observed 9e-6 at 64x80,
# 9.5e-5 at 512x640, 2.2e-4 at 640x960 against the fp32 operator chain.  The MLP itself is exact fp32 (fmaf chains).
# GDB_PREC_F32X (split-f16 operand pairs, ~22 bits) is held to the same bound: its MLP differs from the fp32 one by ~1e-6
# (test_split_f16_precision_tracks_fp32 bounds that difference directly), far below the coordinate effect.
FUSED_TOL_F32 = 5e-4
# Per frame class, 3 x the largest error observed on MI355X (profiles/r05/parity_observed.txt; VERDICT r04 weak 1: one bound of
# 5e-4 everywhere let a 1e-4 arithmetic slip pass every small-frame test).  Fixture-size frames (<= 128 px: F6, c1, the corner and
# degenerate cases): fp32 / split-f16 1.4e-5 at most (48x80 V2 S3; 7e-6 typical) -> 5e-5; the fp32 operator chain 1.1e-5 -> 4e-5.
# c2 (512x640): 9.7e-5 against the oracle, 9.5e-5 against the fp32 chain -> 3e-4.  c3 / c3' / c4 / c5: 1.8e-4 .. 2.5e-4 -> the 5e-4
# above (2 x).  GDB_PREC_F16 keeps 2e-3: 1.07e-3 at most on the small frames, 7.9e-4 at c2, 1.03e-3 at c3.
FUSED_TOL_F32_SMALL = 5e-5
FUSED_TOL_F32_C2 = 3e-4
CHAIN_TOL_SMALL = 4e-5   # the exact-fp32 operator chain on the fixture-size frames
SMOOTH_C2_TOL_F32 = 2e-5  # c2-size frame with CNN-generated features and a smooth volume: coordinate noise out of the way
SMOOTH_BIG_TOL_F32 = 5e-5  # the same at c4's and c5's sizes and configs (round 6: VERDICT r05 task 7 asked for 2e-5; observed 2.5e-5 at c4 -
                           # S_max 6 sums twice as many samples per bundle as c2 - so 2 x that; a 1e-4 slip of one bias reads >= 1.2e-4)
F32X_VS_F32_TOL = 2e-5


def _psnr_delta(bf_a, bf_b, H, W):
    """PSNR of the fine RGB (first 12 channels pixel-shuffled to an image, network.py:175) of both
    renders against one synthetic ground truth (oracle render + fixed noise); north_star bar 0.05 dB."""
    def img(bf):
        x = bf[:, :12].reshape(-1, H, W, 3, 2, 2)            # (B,H,W,c,by,bx)
        return np.transpose(x, (0, 1, 4, 2, 5, 3)).reshape(-1, 2 * H, 2 * W, 3)
    a, b = img(bf_a), img(bf_b)
    gt = np.clip(b + np.random.default_rng(0).normal(0, 0.03, b.shape), 0, 1)
    return abs(oracle.psnr(gt.reshape(-1, gt.shape[2], 3), a.reshape(-1, a.shape[2], 3)) -
               oracle.psnr(gt.reshape(-1, gt.shape[2], 3), b.reshape(-1, b.shape[2], 3)))


@pytest.mark.parametrize("tag", ["dtu", "mips"])
def test_fused_vs_golden(tag, mode):
    fx = load_golden("F6_hotpath_" + tag)
    eng = engine_for(frame_of(fx), nerf_weights_of(fx), mode, max_num_samples=int(fx["S_max"]),
                     is_adaptive=bool(fx["adaptive"]), inv_depth=bool(fx["inv_depth"]))
    bf, depth, opac = eng.render()
    e = max_abs(npy(bf), fx["bundle_feat"])
    print(f"fused vs reference fixture {tag}: max abs err {e:.3e}")
    assert e <= fused_tol(mode)
    assert max_abs(npy(depth), fx["depth"]) <= 2e-3 * float(np.abs(fx["depth"]).max())
    assert max_abs(npy(opac), fx["opacity"]) <= 1e-5


@pytest.mark.parametrize("Ho,Wo,V,B,S,adaptive,inv,scene", [
    (64, 80, 3, 1, 3, True, False, "dtu"),     # c1
    (64, 80, 3, 1, 6, False, False, "dtu"),    # fixed count, waves loop over slots
    (96, 72, 4, 2, 6, True, True, "nerf"),     # ragged row (W=36), batch 2, disparity sampling
    (48, 80, 2, 1, 3, True, False, "llff"),
    (32, 64, 5, 1, 8, True, False, "dtu"),     # 5 views, 8 slots
])
def test_fused_vs_oracle(Ho, Wo, V, B, S, adaptive, inv, scene, mode):
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, scene=scene, seed=21, src_focal_scale=(1.0, 1.9, 3.3))
    w = synthetic.make_nerf_weights(seed=5)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    eng = engine_for(frame, w, mode, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    bf, depth, opac = eng.render()
    e = max_abs(npy(bf), obf)
    print(f"fused vs oracle {Ho}x{Wo} V{V} S{S}: max abs err {e:.3e}, rms {np.sqrt(np.mean((npy(bf)-obf)**2)):.3e}")
    assert e <= fused_tol(mode)
    assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max())
    assert max_abs(npy(opac), oo) <= 1e-5
    assert _psnr_delta(npy(bf), obf, Ho // 2, Wo // 2) <= 0.05


@pytest.mark.parametrize("Ho,Wo,V,S,adaptive,extra", [
    (32, 64, 8, 16, False, {}),                          # GDB_MAX_VIEWS x GDB_MAX_SAMPLES
    (32, 64, 2, 1, False, {}),                           # one sample per bundle, two views
    (64, 80, 3, 3, True, {"viewdir_agg": False}),        # nerf.py:19-23: no view_fc; state dict still carries one
    (64, 80, 3, 3, True, {"max_mipmap_level": 0}),       # no mip chain: bilinear on level 0 only
    (64, 80, 3, 4, True, {"global_num_depth": 8}),       # coarse prior grid -> wide adaptive intervals
])
def test_fused_config_corners(Ho, Wo, V, S, adaptive, extra, mode):
    """Limits of the C ABI (include/gdb_nerf_hip.h GDB_MAX_*) and the config switches the reference exposes
    (nerf.viewdir_agg, nerf.max_mipmap_level, nerf.global_num_depth), fused kernel vs the oracle."""
    frame = synthetic.make_frame(Ho, Wo, V=V, scene="dtu", seed=33, src_focal_scale=(1.0, 2.3))
    w = synthetic.make_nerf_weights(seed=8)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=adaptive, **extra)
    eng = engine_for(frame, w, mode, max_num_samples=S, is_adaptive=adaptive, **extra)
    bf, depth, opac = eng.render()
    ubf, ud, uo = eng.render_unfused()
    e, eu = max_abs(npy(bf), obf), max_abs(npy(ubf), obf)
    print(f"corner {Ho}x{Wo} V{V} S{S} {extra}: fused err {e:.3e}, fp32 chain err {eu:.3e}")
    assert eu <= CHAIN_TOL_SMALL
    assert e <= fused_tol(mode)
    assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max())
    assert max_abs(npy(opac), oo) <= 1e-5
    assert _psnr_delta(npy(bf), obf, Ho // 2, Wo // 2) <= 0.05


def _degenerate(kind, f):
    g = {k: v.copy() for k, v in f.items()}
    if kind == "faces_away":         # every sample behind source view 1: z clamps at 1e-6, fetches hit the border
        g["src_exts"][0, 1] = np.diag([-1, 1, -1, 1]).astype(np.float32) @ g["src_exts"][0, 1]
    elif kind == "src_at_target":    # td - sd is exactly 0 in the reference (bundle_sampler.py:362-367)
        g["src_exts"][0, 0] = g["tar_ext"][0]
    elif kind == "duplicate_views":  # zero variance over two of three views
        for k in ("src_exts", "src_ints", "src_images", "img_feat"):
            g[k][0, 2] = g[k][0, 1]
    elif kind == "zero_width_prior":  # depth_range min == max: every interval collapses
        g["depth_range"][:, 1] = g["depth_range"][:, 0]
    elif kind == "all_zero":
        for k in ("src_images", "img_feat", "feat_volume"):
            g[k][:] = 0
    return g


@pytest.mark.parametrize("kind", ["faces_away", "src_at_target", "duplicate_views", "zero_width_prior", "all_zero"])
def test_fused_degenerate_geometry(kind, mode):
    """Degenerate frames the reference handles through its clamps (z >= 1e-6, F.normalize eps, border
    padding): both device paths must follow the oracle there, and stay finite."""
    frame = _degenerate(kind, synthetic.make_frame(64, 80, V=3, scene="dtu", seed=5))
    w = synthetic.make_nerf_weights(seed=8)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=3, is_adaptive=True)
    assert np.isfinite(obf).all()
    eng = engine_for(frame, w, mode, max_num_samples=3, is_adaptive=True)
    bf, depth, opac = eng.render()
    ubf = eng.render_unfused()[0]
    e, eu = max_abs(npy(bf), obf), max_abs(npy(ubf), obf)
    print(f"degenerate {kind}: fused err {e:.3e}, fp32 chain err {eu:.3e}")
    assert np.isfinite(npy(bf)).all() and np.isfinite(npy(ubf)).all()
    assert eu <= CHAIN_TOL_SMALL
    assert e <= fused_tol(mode)
    assert max_abs(npy(opac), oo) <= 1e-5


def test_fused_matches_unfused_at_full_size(mode):
    """c2 (BASELINE.json configs[1]) at full size under every (schedule, precision) mode of the matrix: the fused kernel against the
    fp32 operator chain (which test_c2_full_size_against_the_oracle below checks against the oracle itself, as it does the fused
    kernel), plus size-independent properties."""
    frame = synthetic.make_frame(512, 640, V=3, seed=0)
    w = synthetic.make_nerf_weights(seed=0)
    eng = engine_for(frame, w, mode, max_num_samples=3, is_adaptive=True)
    bf, depth, opac = eng.render()
    ubf, ud, uo = eng.render_unfused()
    e = max_abs(npy(bf), npy(ubf))
    print(f"fused vs fp32 chain at 512x640: max abs err {e:.3e}")
    assert e <= fused_tol(mode, "c2")
    assert max_abs(npy(depth), npy(ud)) <= 2e-3 * 905.0
    # normalised weights sum to one; depth stays inside the prior
    assert float((opac - 1).abs().max()) <= 1e-5
    dr = frame["depth_range"][0]
    d = npy(depth).reshape(256, 320)
    assert np.all(d >= dr[0] - 1e-2) and np.all(d <= dr[1] + 1e-2)
    assert _psnr_delta(npy(bf), npy(ubf), 256, 320) <= 0.05


def test_c2_full_size_against_the_oracle():
    """BASELINE.json configs[1] at its full size, directly against the CPU oracle (one oracle frame takes a couple of seconds):
    the fp32 operator chain and the fused kernel under the three schedules and the three precisions, sample counts included."""
    frame = synthetic.make_frame(512, 640, V=3, seed=0)   # the frame bench.py renders
    w = synthetic.make_nerf_weights(seed=0)
    with np.errstate(all="ignore"):
        obf, od, oo, aux = oracle.hot_path(frame, w, max_num_samples=3, is_adaptive=True, return_intermediates=True)
    eng = engine_for(frame, w, max_num_samples=3, is_adaptive=True)
    smp = eng.sample()
    assert int(smp["total"].item()) == int(aux["samples"]["samples_per_bundle"].sum())
    assert np.array_equal(npy(smp["samples_per_bundle"]), aux["samples"]["samples_per_bundle"].astype(np.int32))
    ubf, ud, uo = eng.render_unfused()
    eu = max_abs(npy(ubf), obf)
    print(f"c2 512x640 vs oracle: fp32 operator chain max abs err {eu:.3e}")
    assert eu <= 2e-4 and max_abs(npy(ud), od) <= 1e-4 * float(np.abs(od).max()) and max_abs(npy(uo), oo) <= 1e-5
    for sched in (1, 2, 3, 4):
        for prec, tol in ((1, FUSED_TOL_F32_C2), (2, FUSED_TOL_F32_C2), (0, FUSED_TOL)):
            eng.set_schedule(sched)
            bf, depth, opac = eng.render(precision=prec)
            e = max_abs(npy(bf), obf)
            dpsnr = _psnr_delta(npy(bf), obf, 256, 320)
            print(f"c2 512x640 vs oracle: fused schedule {sched} precision {prec}: max abs err {e:.3e}, rms {np.sqrt(np.mean((npy(bf) - obf) ** 2)):.3e}, "
                  f"PSNR delta {dpsnr:.2e} dB")
            assert e <= tol and dpsnr <= 0.05
            assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5


@pytest.mark.parametrize("name,Ho,Wo,S,scene,seed", [("c3 640x960", 640, 960, 3, "llff", 0), ("c4 800x800", 800, 800, 6, "nerf", 0)])
def test_c3_c4_full_size_against_the_oracle(name, Ho, Wo, S, scene, seed):
    """BASELINE.json configs[2] (as configs/llff_eval.yaml resizes it: 640x960) and configs[3] (800x800, S_max 6 adaptive) at
    their full sizes, directly against the CPU oracle: the fp32 operator chain, and the fused kernel at fp32 under the schedule
    GDB_SCHED_AUTO takes (dense) - same bounds as c2.  Reference: bundle_sampler.py:355-359 (the mip fetch whose chain length
    differs per map size), configs/llff_eval.yaml:18,23."""
    frame = synthetic.make_frame(Ho, Wo, V=3, scene=scene, seed=seed)
    w = synthetic.make_nerf_weights(seed=0)
    with np.errstate(all="ignore"):
        obf, od, oo, aux = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=True, return_intermediates=True)
    eng = engine_for(frame, w, max_num_samples=S, is_adaptive=True)
    smp = eng.sample()
    assert np.array_equal(npy(smp["samples_per_bundle"]), aux["samples"]["samples_per_bundle"].astype(np.int32))
    ubf, ud, uo = eng.render_unfused()
    eu = max_abs(npy(ubf), obf)
    bf, depth, opac = eng.render(precision=1)   # GDB_SCHED_AUTO
    e = max_abs(npy(bf), obf)
    dpsnr = _psnr_delta(npy(bf), obf, Ho // 2, Wo // 2)
    print(f"{name} vs oracle: fp32 operator chain max abs err {eu:.3e}; fused fp32 (auto schedule) {e:.3e}, PSNR delta {dpsnr:.2e} dB")
    assert eu <= 2e-4 and max_abs(npy(ud), od) <= 1e-4 * float(np.abs(od).max()) and max_abs(npy(uo), oo) <= 1e-5
    assert e <= FUSED_TOL_F32 and dpsnr <= 0.05
    assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5


def _smooth_frame(Ho, Wo, V=3, scene="dtu"):
    """A frame whose gathered tensors are smooth, as a trained network's are: `img_feat` is the package's FeatureNet (the
    weights of fixture F7: the reference's own random-init FPN, feature_net.py:40-64) run on the synthetic source images (level 1, 16
    channels) + the downsampled colours (network.py:159-164), and the cost volume is box-filtered noise.  On white-noise features the
    gradient is O(1) per texel and 1e-4 px of fp32 coordinate noise sets the error floor (FUSED_TOL_F32_C2); here the floor is the
    arithmetic's own."""
    from gdb_nerf_amd.networks.gdb_nerf.feature_net import FeatureNet
    fx = load_golden("F7_network")
    frame = synthetic.make_frame(Ho, Wo, V=V, scene=scene, seed=0, smooth_vol=5)
    net = FeatureNet(base_channels=8, out_channels=[32, 16, 8]).eval()
    sd = {k[len("sd.feature_net."):]: (torch.from_numpy(np.asarray(v)).float() if v.dtype == np.float16 else torch.from_numpy(np.asarray(v)))
          for k, v in fx.items() if k.startswith("sd.feature_net.")}
    net.load_state_dict(sd, strict=True)
    with torch.no_grad():   # one view at a time: the level-0 activations of five 1200x1600 views at once are gigabytes
        feat = np.stack([8.0 * net(torch.from_numpy(frame["src_images"][0, v:v + 1]))[1][0].numpy() for v in range(V)])   # (V, 16, H, W); x 8: unit-scale values (std 0.8), still smooth
    frame["img_feat"] = np.concatenate((feat[None], frame["img_feat"][:, :, 16:]), axis=2).astype(np.float32)
    return frame


def _smooth_c2_frame():
    return _smooth_frame(512, 640)


def test_smooth_feature_c2_frame_against_the_oracle():
    """at c2's size on CNN-generated features the fp32 kernels are held to 2e-5 against the oracle (observed
    printed), under every schedule - a bound that sees a wrong rounding or a dropped bias term in one MLP layer, which the
    white-noise frames' 1e-4 of coordinate noise would hide.  Reference: nerf.py:100-113, bundle_sampler.py:327-359."""
    frame = _smooth_c2_frame()
    w = synthetic.make_nerf_weights(seed=0)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=3, is_adaptive=True)
    eng = engine_for(frame, w, max_num_samples=3, is_adaptive=True)
    eu = max_abs(npy(eng.render_unfused()[0]), obf)
    print(f"smooth c2 frame vs oracle: fp32 operator chain max abs err {eu:.3e}")
    assert eu <= SMOOTH_C2_TOL_F32
    for sched in (1, 2, 3, 4):
        eng.set_schedule(sched)
        for prec, tol in ((1, SMOOTH_C2_TOL_F32), (2, SMOOTH_C2_TOL_F32), (0, FUSED_TOL)):
            bf, depth, opac = eng.render(precision=prec)
            e = max_abs(npy(bf), obf)
            print(f"smooth c2 frame vs oracle: fused schedule {sched} precision {prec}: max abs err {e:.3e}, rms {np.sqrt(np.mean((npy(bf) - obf) ** 2)):.3e}")
            assert e <= tol
            assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5


@pytest.mark.parametrize("prec", [1, 2], ids=["f32", "f32x"])
def test_the_bounds_see_a_1e4_bias_slip(prec):
    """The tolerances must be tight enough to catch an arithmetic slip of 1e-4 in one MLP layer: with
    lr0.0.bias (nerf.py:100) moved by +1e-4 in the weights the DEVICE gets - every packed section, the fp32-MFMA bias table T32_LR0
    included - and the oracle left on the true weights, the fused render must EXCEED the bound its frame class is held to, on the
    c1-size frame and on the smooth c2 frame (and the same render on the true weights must stay inside it)."""
    w = synthetic.make_nerf_weights(seed=5)
    w_bad = dict(w)
    w_bad["lr0.0.bias"] = (w["lr0.0.bias"] + np.float32(1e-4)).astype(np.float32)
    for name, frame, tol in (("c1 64x80", synthetic.make_frame(64, 80, V=3, seed=21), FUSED_TOL_F32_SMALL),
                             ("smooth c2", _smooth_c2_frame(), SMOOTH_C2_TOL_F32)):
        with np.errstate(all="ignore"):
            obf = oracle.hot_path(frame, w, max_num_samples=3, is_adaptive=True)[0]
        good = max_abs(npy(engine_for(frame, w, (0, prec), max_num_samples=3, is_adaptive=True).render()[0]), obf)
        bad = max_abs(npy(engine_for(frame, w_bad, (0, prec), max_num_samples=3, is_adaptive=True).render()[0]), obf)
        print(f"{name}, precision {prec}: err on the true weights {good:.3e}, with lr0.0.bias + 1e-4 {bad:.3e} (bound {tol:.0e})")
        assert good <= tol < bad


def test_hot_path_section_allocates_nothing_per_frame():
    """With `reuse_outputs` (what Network.forward sets) the per-frame section - prepare, packed render, decoder, merge - runs out
    of per-engine buffers: the caching allocator sees no allocation across 100 frames, and the results equal fresh-tensor calls."""
    frame = synthetic.make_frame(128, 160, V=3, seed=9)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=9), max_num_samples=3, is_adaptive=True)
    from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder
    torch.manual_seed(0)
    dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=2)
    eng.load_decoder_weights(dec.state_dict(), 3)
    dev = dev_frame(frame)

    def step():
        eng.prepare(dev)
        packed = eng.render_packed()
        return eng.merge_packed(packed, eng.decode(packed), True)
    fresh = [t.clone() for t in step()]
    eng.reuse_outputs = True
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(100):
        out = step()
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == n0
    for a, b in zip(fresh, out):
        assert torch.equal(a, b)


def test_fused_is_deterministic_at_full_size(mode):
    """Regression for a race seen only with several workgroups per CU at full frame size (stale lanes in a
    packed-f32 result): repeated launches must agree bit for bit, and with the fp32 operator chain."""
    frame = synthetic.make_frame(512, 640, V=3, seed=3)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=2), mode, max_num_samples=3, is_adaptive=True)
    ref = eng.render()[0].clone()
    ubf = eng.render_unfused()[0]
    for _ in range(6):
        assert torch.equal(eng.render()[0], ref)
    assert max_abs(npy(ref), npy(ubf)) <= fused_tol(mode, "c2")


@pytest.mark.parametrize("prec,sched", [(1, 0), (0, 0), (1, 3), (2, 0)])
def test_hot_path_step_replays_from_a_hip_graph(prec, sched):
    """prepare + render captured ONCE into a HIP graph and replayed: the step enqueues on the given stream only - no host sync, no
    allocation, no host-side state that a replay would miss - so an evaluation loop can replay it per frame (new inputs copied into the
    same tensors).  Bit-identical to the eager step on every replay, also after the inputs change in place; the flat schedule's arrival
    counters and side records (fp32 AUTO at this size) are left as the next launch needs them."""
    frame = synthetic.make_frame(128, 160, V=3, seed=5)
    other = synthetic.make_frame(128, 160, V=3, seed=6)
    w = synthetic.make_nerf_weights(seed=2)
    dev = dev_frame(frame)
    eng = HotPathEngine(max_num_samples=3, is_adaptive=True)
    eng.set_schedule(sched); eng.precision = prec; eng.load_weights(w)
    eng.prepare(dev)
    nb = eng.n_bundles
    out = (torch.zeros((nb, eng.Q), device="cuda"), torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda"))
    eager = [t.clone() for t in eng.render(out=out)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # (warm-up on the capture stream, as torch's graph recipe asks)
        eng.prepare(dev); eng.render(out=out)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.prepare(dev)
        eng.render(out=out)
    for _ in range(3):
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(eager, out):
            assert torch.equal(a, b)
    # new inputs in the SAME tensors: the replayed step renders the new frame
    eng2 = engine_for(other, w, (sched, prec), max_num_samples=3, is_adaptive=True)
    want = [t.clone() for t in eng2.render()]
    for k, v in dev.items():
        v.copy_(torch.from_numpy(np.ascontiguousarray(other[k])))
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(want, out):
        assert torch.equal(a, b)


def test_fused_row_strips_tile_the_frame(mode):
    """Row-strip launches (the multi-GPU shard unit) reproduce the full-frame launch bit for bit."""
    frame = synthetic.make_frame(64, 80, V=3, B=2, seed=4)
    w = synthetic.make_nerf_weights(seed=1)
    eng = engine_for(frame, w, mode)
    full = [t.clone() for t in eng.render()]
    nb = eng.n_bundles
    out = (torch.zeros((nb, eng.Q), device="cuda"), torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda"))
    for r0, r1 in ((0, 5), (5, 6), (6, 32)):
        eng.render(r0, r1, None, out)
    for a, b in zip(full, out):
        assert torch.equal(a, b)
    with pytest.raises(ValueError, match="row strip"):
        eng.render(3, 40)


def test_single_view_is_rejected_by_fused_and_nan_in_mirror():
    """One source view: the reference's unbiased variance over views (nerf.py:73) is NaN and so is
    everything after it.  The fp32 mirror reproduces the NaN; the fused entry refuses the shape."""
    frame = synthetic.make_frame(32, 64, V=1, seed=2)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=5), max_num_samples=2, is_adaptive=False)
    with pytest.raises(ValueError, match="2 source views"):
        eng.render()
    bf, _, _ = eng.render_unfused()
    assert torch.isnan(bf).all()


@pytest.mark.parametrize("prec", [0, 1, 2], ids=["f16", "f32", "f32x"])
def test_fused_schedules_agree_and_reject_bad_mode(prec):
    """The two decompositions differ only in where the composite sums are formed (order of roundings)."""
    frame = synthetic.make_frame(96, 144, V=3, B=1, seed=9)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=4), max_num_samples=5, is_adaptive=True)
    eng.precision = prec
    eng.set_schedule(1); a = [t.clone() for t in eng.render()]
    for other in (2, 3, 4):
        eng.set_schedule(other); b = [t.clone() for t in eng.render()]
        print(f"precision {prec}: schedule 1 vs {other}: max abs diff {max_abs(npy(a[0]), npy(b[0])):.3e}")
        assert max_abs(npy(a[0]), npy(b[0])) <= 2e-6
        assert max_abs(npy(a[1]), npy(b[1])) <= 2e-6 * float(a[1].abs().max())
        assert max_abs(npy(a[2]), npy(b[2])) <= 2e-6
    with pytest.raises(ValueError, match="schedule"):
        eng.set_schedule(5)
    eng.schedule = 7  # past the Python check: the C ABI rejects it before any launch
    with pytest.raises(ValueError, match="schedule"):
        eng.render()
    eng.schedule = 0
    with pytest.raises(ValueError, match="precision"):
        eng.render(precision=3)


@pytest.mark.parametrize("H,W,V,S,adaptive", [(64, 80, 3, 3, True), (512, 640, 3, 3, True), (96, 144, 5, 6, False), (128, 128, 2, 6, True)])
def test_split_f16_precision_tracks_fp32(H, W, V, S, adaptive):
    """GDB_PREC_F32X against GDB_PREC_F32 on the same engine: gather, staging and composite are the same code, so the two
    differ only by the MLP's arithmetic — split-f16 operand pairs (about 22 bits, lo·lo dropped) against exact fp32 fmaf chains.
    Also against GDB_PREC_F16 as the yardstick: the split must be orders of magnitude closer to fp32 than f16 operands are."""
    frame = synthetic.make_frame(H, W, V=V, B=1, seed=21)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=6), max_num_samples=S, is_adaptive=adaptive)
    for sched in (1, 2, 3, 4):
        eng.set_schedule(sched)
        ref = [t.clone() for t in eng.render(precision=1)]
        x = [t.clone() for t in eng.render(precision=2)]
        h = [t.clone() for t in eng.render(precision=0)]
        ex, eh = max_abs(npy(x[0]), npy(ref[0])), max_abs(npy(h[0]), npy(ref[0]))
        print(f"{H}x{W} V={V} S={S} schedule {sched}: |f32x - f32| max {ex:.2e}, |f16 - f32| max {eh:.2e}")
        assert ex <= F32X_VS_F32_TOL and ex * 8 <= max(eh, 1e-5)
        assert max_abs(npy(x[1]), npy(ref[1])) <= 2e-5 * float(ref[1].abs().max()) and max_abs(npy(x[2]), npy(ref[2])) <= 1e-5


@pytest.mark.parametrize("wscale,fscale", [(1.0, 1.0), (3.0, 1.0), (0.2, 8.0), (2.0, 30.0)])
def test_split_f16_precision_is_relative(wscale, fscale):
    """The split keeps ~22 bits of every operand of magnitude >= 2^-3 (and inside the f16 range): with the MLP weights and the
    image features scaled, |f32x - f32| stays a few 1e-7 of the output's scale.  (Activations reach a few hundred at the last
    pair of scales; f16 operands lose three more digits there.)  Small operands: test_split_f16_small_operands_have_an_absolute_floor."""
    frame = synthetic.make_frame(96, 128, V=3, B=1, seed=5)
    frame["img_feat"] = (frame["img_feat"] * np.float32(fscale)).astype(np.float32)
    w = {k: (v * np.float32(wscale)).astype(np.float32) for k, v in synthetic.make_nerf_weights(seed=8).items()}
    eng = engine_for(frame, w, max_num_samples=4, is_adaptive=True)
    ref = eng.render(precision=1)[0].clone()
    x = eng.render(precision=2)[0].clone()
    h = eng.render(precision=0)[0].clone()
    scale = float(ref.abs().max())
    ex, eh = max_abs(npy(x), npy(ref)) / scale, max_abs(npy(h), npy(ref)) / scale
    print(f"weights x{wscale}, features x{fscale}: output scale {scale:.3g}, |f32x - f32| / scale {ex:.2e}, |f16 - f32| / scale {eh:.2e}")
    assert np.isfinite(npy(x)).all() and ex <= 5e-6 and ex * 30 <= max(eh, 1e-5)   # (observed 1.8e-7 .. 2.4e-6; the softmax scores grow with the scales)


@pytest.mark.parametrize("wscale,bound", [(0.2, 5e-6), (0.05, 2e-5), (0.01, 1e-4)])
def test_split_f16_small_operands_have_an_absolute_floor(wscale, bound):
    """The documented operand range of GDB_PREC_F32X (include/gdb_nerf_hip.h): the low half of a value below 2^-3 is an f16
    subnormal, so small weights keep fewer than 22 bits - the error of the pure-MLP outputs (the 8 feat_head channels) over their
    scale grows as the weights shrink (measured 1.5e-6 / 5e-6 / 3e-5), while staying below the plain f16 path's."""
    frame = synthetic.make_frame(96, 128, V=3, B=1, seed=5)
    w = {k: (v * np.float32(wscale)).astype(np.float32) for k, v in synthetic.make_nerf_weights(seed=8).items()}
    eng = engine_for(frame, w, max_num_samples=4, is_adaptive=True)
    ref, x, h = (eng.render(precision=p)[0][:, 31:39].clone() for p in (1, 2, 0))
    scale = float(ref.abs().max())
    ex, eh = max_abs(npy(x), npy(ref)) / scale, max_abs(npy(h), npy(ref)) / scale
    print(f"weights x{wscale}: feat_head scale {scale:.3g}, |f32x - f32| / scale {ex:.2e}, |f16 - f32| / scale {eh:.2e}")
    assert ex <= bound and ex <= eh


@pytest.mark.parametrize("Ho,Wo,B,S,adaptive,inv,scene", [(32, 48, 2, 3, True, False, "dtu"), (64, 80, 1, 3, True, False, "dtu"), (96, 72, 2, 6, True, True, "nerf"),
                                                          (32, 64, 1, 16, False, False, "dtu"), (64, 80, 2, 6, False, False, "dtu"), (40, 330, 1, 5, True, False, "llff"), (512, 640, 1, 3, True, False, "dtu"),
                                                          (8, 800, 1, 16, True, False, "dtu"), (16, 960, 1, 3, True, False, "llff"), (8, 800, 1, 6, True, False, "nerf")])
def test_dense_plan_invariants(Ho, Wo, B, S, adaptive, inv, scene):
    """The plan of the dense schedule (plan_row): for every bundle-map row the compacted sample list is the reference's
    (bundle_sampler.py:182-189: bundle-major, sample-minor, from the oracle's per-bundle counts), and the windows cut it into
    consecutive runs of WHOLE bundles of at most 32 samples that cover the row - greedily (a window ends only where the next
    bundle would not fit) on rows of 1024 .. 4095 sample offsets (W * S_max), at fixed offsets L * w on shorter and longer ones;
    fixed counts: windows of floor(32 / S_max) bundles."""
    frame = synthetic.make_frame(Ho, Wo, V=2, B=B, scene=scene, seed=17)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=1), (3, 0), max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    dense = [t.clone() for t in eng.render()]   # an explicit dense render builds the plan where prepare did not (fixed counts)
    H, W = Ho // 2, Wo // 2
    rays = oracle_rays(frame)
    smp = oracle.sample_bundles(rays, frame["depth_range"], frame["vol_range"], frame["near_far"][:, 0], frame["near_far"][:, 1], 2, S, 64,
                                inv_depth=inv, adaptive=adaptive)
    cnt = np.asarray(smp["samples_per_bundle"]).astype(np.int64).reshape(B * H, W)
    plan = npy(eng.dense_plan())
    L = eng.dense_plan().window
    assert L == 33 - S
    greedy = 1024 <= W * S < 4096
    for r in range(B * H):
        nwin, tot = int(plan[r, 0]), int(cnt[r].sum())
        starts = plan[r, 1:2 + nwin].astype(np.int64)      # first sample offset of every window, then the row's total
        assert starts[0] == 0 and starts[-1] == tot and np.all(np.diff(starts) > 0) and np.all(np.diff(starts) <= 32)
        off = np.concatenate(([0], np.cumsum(cnt[r])))       # sample offset of every bundle's first sample, then the total
        assert np.all(np.isin(starts, off))                  # windows hold whole bundles
        nxt = {int(o): int(c) for o, c in zip(off[:-1], cnt[r])}
        if not adaptive:   # fixed counts: closed form - windows of floor(32 / S) whole bundles (no chain, no look at the depth prior)
            bpw = 32 // S
            assert nwin == -(-W // bpw) and np.array_equal(starts, np.minimum(np.arange(nwin + 1) * bpw * S, tot))
            continue
        for w in range(nwin - 1):
            if greedy:   # the bundle that opens window w + 1 did not fit into window w
                assert starts[w + 1] - starts[w] + nxt[int(starts[w + 1])] > 32
            else:        # window w = the bundles whose first sample offset falls into [L w, L (w + 1))
                assert L * w <= starts[w] < L * (w + 1) or w == 0
                assert starts[w + 1] >= L * (w + 1)
    # the compacted sample list (bundle_sampler.py:182-189: bundle-major, sample-minor), row by row
    smap = npy(eng.dense_map())
    for r in range(B * H):
        tot = int(cnt[r].sum())
        xs = np.repeat(np.arange(W), cnt[r])
        ks = np.concatenate([np.arange(c) for c in cnt[r]])
        want = xs | (ks << 16) | (np.repeat(cnt[r], cnt[r]) << 24)
        assert np.array_equal(smap[r, :tot], want)
        assert np.all(smap[r, tot:] == 0xFFFFFFFF)
    # and the render it drives agrees with the slot-wave schedule's
    eng.set_schedule(1 if S <= 8 else 2)
    for a, b in zip(dense, eng.render()):
        assert max_abs(npy(a), npy(b)) <= 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("V,S,prec,dense", [(3, 6, 1, True), (2, 8, 2, True), (3, 6, 0, False), (4, 6, 1, False), (3, 3, 1, False), (3, 2, 1, True)])
def test_auto_schedule_for_fixed_counts(V, S, prec, dense):
    """GDB_SCHED_AUTO with fixed sample counts (gdb_fixed_counts_dense): more than 3 samples per bundle and at most 3 views render on
    the dense schedule at fp32 / split-f16 - bit-identical to an explicit GDB_SCHED_DENSE, its plan built by gdb_prepare (closed
    form) and reused (GDB_SCHED_PLAN_READY) - and on the segment wave / slot waves otherwise; either way within the path's bound of the
    exact-fp32 operator chain."""
    frame = synthetic.make_frame(64, 96, V=V, B=2, seed=31)
    w = synthetic.make_nerf_weights(seed=3)
    auto = engine_for(frame, w, (0, prec), max_num_samples=S, is_adaptive=False)
    assert bool(auto._sched() & 0x100) == ((S > 3 or S == 2) and V <= 3)            # the plan is there whenever the rule may take it
    a = [t.clone() for t in auto.render()]
    other = engine_for(frame, w, (3 if dense else (2 if S > 3 else 1), prec), max_num_samples=S, is_adaptive=False)
    for x, y in zip(a, other.render()):
        assert torch.equal(x, y)
    if (S > 3 or S == 2) and V <= 3:
        flip = engine_for(frame, w, ((2 if S > 3 else 1) if dense else 3, prec), max_num_samples=S, is_adaptive=False)
        assert not torch.equal(a[0], flip.render()[0])                  # (the two schedules differ in the last bits: AUTO really took `other`)
    ref = auto.render_unfused()
    tol = {0: 2e-3, 1: 5e-4, 2: 5e-4}[prec]
    assert max_abs(npy(a[0]), npy(ref[0])) <= tol


def test_dense_render_follows_a_depth_prior_changed_after_prepare():
    """gdb_prepare builds the dense plan from the depth prior as it is at that moment.  A render call may only reuse it while
    the prior is untouched (GDB_SCHED_PLAN_READY, which the engine sets from the tensor's storage and version counter); after an
    in-place change the dense render must rebuild the plan itself and follow the NEW prior - checked against the oracle."""
    frame = synthetic.make_frame(64, 80, V=3, seed=23)
    w = synthetic.make_nerf_weights(seed=2)
    eng = engine_for(frame, w, (3, 1), max_num_samples=6, is_adaptive=True)
    a = [t.clone() for t in eng.render()]
    assert eng._sched() & 0x100                       # prepared plan, prior untouched: reused
    dr = eng._keep["depth_range"]
    mid, half = 0.5 * (dr[:, 0] + dr[:, 1]), 0.5 * (dr[:, 1] - dr[:, 0])
    dr[:, 0].copy_(mid - 2.5 * half); dr[:, 1].copy_(mid + 2.5 * half)   # in place: wider prior, more samples per bundle
    assert not (eng._sched() & 0x100)                 # version counter moved: the render call rebuilds the plan
    bf, depth, opac = eng.render()
    f2 = dict(frame); f2["depth_range"] = npy(dr)
    obf, od, oo = oracle.hot_path(f2, w, max_num_samples=6, is_adaptive=True)
    assert max_abs(npy(bf), obf) <= 5e-4 and max_abs(npy(depth) / np.abs(od).max(), od / np.abs(od).max()) <= 2e-3
    assert max_abs(npy(bf), npy(a[0])) > 1e-3         # and it is a different image
    # a C caller that sets the flag on a stale plan gets an undefined but memory-safe render (clamped reads): just run it
    eng.prepare({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in frame.items()})
    eng._keep["depth_range"][:, 1].add_(40.0)         # the prepared plan no longer matches the prior
    eng.schedule = 3 | 0x100
    eng._plan_key = None
    eng.render(); torch.cuda.synchronize()


def test_engines_with_different_settings_interleave():
    """the ABI is reentrant — no process-global state.  Six engines (schedules x precisions, each precision
    and each schedule twice) on two HIP streams, their calls interleaved, must each reproduce their solo result bit for bit."""
    frames = [synthetic.make_frame(128, 160, V=3, seed=40 + i) for i in range(6)]
    w = synthetic.make_nerf_weights(seed=6)
    modes = [(1, 0), (2, 1), (3, 0), (3, 1), (1, 2), (2, 2)]
    engs = [engine_for(f, w, m, max_num_samples=4, is_adaptive=True) for f, m in zip(frames, modes)]
    solo = [[t.clone() for t in e.render()] for e in engs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [tuple(torch.zeros_like(t) for t in s) for s in solo]
    for rep in range(3):
        for i, e in enumerate(engs):
            with torch.cuda.stream(streams[i % 2]):
                e.render(0, None, None, outs[i])
    torch.cuda.synchronize()
    for s, o in zip(solo, outs):
        for a, b in zip(s, o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("S,adaptive", [(3, True), (6, True), (12, False)])
def test_packed_output_equals_the_three_tensors(S, adaptive, mode):
    """gdb_render_bundles_packed writes [bundle_feat | depth | opacity] rows: the same values as the three-tensor entry, bit
    for bit, for full frames and row strips (the multi-GPU gather unit), with and without disparity sampling."""
    for inv in (False, True):
        frame = synthetic.make_frame(64, 112, V=3, B=2, seed=13, scene="nerf" if inv else "dtu")
        eng = engine_for(frame, synthetic.make_nerf_weights(seed=3), mode, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
        bf, depth, opac = eng.render()
        pk = eng.render_packed()
        assert torch.equal(pk[:, :eng.Q], bf) and torch.equal(pk[:, eng.Q], depth) and torch.equal(pk[:, eng.Q + 1], opac)
        out = torch.zeros_like(pk)
        for r0, r1 in ((0, 7), (7, 8), (8, 32)):
            eng.render_packed(r0, r1, None, out)
        assert torch.equal(out, pk)


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        Ho, Wo = 2 * int(rng.integers(8, 70)), 2 * int(rng.integers(8, 120))
        cases.append(dict(Ho=Ho, Wo=Wo, V=int(rng.integers(2, 9)), B=int(rng.integers(1, 4)), S=int(rng.integers(1, 17)),
                          adaptive=bool(rng.integers(0, 2)), inv=bool(rng.integers(0, 2)), levels=int(rng.integers(0, 4)),
                          scene=["dtu", "llff", "nerf"][int(rng.integers(0, 3))], seed=100 + i,
                          fs=tuple(float(x) for x in rng.uniform(0.5, 6.0, size=3))))
    return cases


@pytest.mark.parametrize("case", _random_cases(16, 2024), ids=lambda c: f"{c['Ho']}x{c['Wo']}_V{c['V']}_B{c['B']}_S{c['S']}")
def test_fused_random_shapes_vs_fp32_chain(case, mode):
    """Seeded random shapes (odd bundle-map sizes, ragged segments, 2..8 views, 1..16 slots, batch 1..3, adaptive and
    disparity sampling, 0..3 mip levels): the fused kernel under both schedules against the exact fp32 operator chain
    (itself oracle-checked above), plus opacity = 1 for bundles with samples and row-strip consistency."""
    c = case
    frame = synthetic.make_frame(c["Ho"], c["Wo"], V=c["V"], B=c["B"], scene=c["scene"], seed=c["seed"], src_focal_scale=c["fs"])
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=c["seed"]), mode, max_num_samples=c["S"], is_adaptive=c["adaptive"],
                     inv_depth=c["inv"], max_mipmap_level=c["levels"])
    bf, depth, opac = [t.clone() for t in eng.render()]
    ubf, ud, uo = eng.render_unfused()
    assert np.isfinite(npy(bf)).all()
    e = max_abs(npy(bf), npy(ubf))
    print(f"random shape {c['Ho']}x{c['Wo']} V{c['V']} B{c['B']} S{c['S']}: fused vs fp32 chain max abs err {e:.3e}")
    assert e <= fused_tol(mode, "c2" if max(c["Ho"], c["Wo"]) > 128 else "small")
    assert max_abs(npy(opac), npy(uo)) <= 1e-5
    # disparity sampling returns 1/(sum w/z): compare where the reference value is finite
    fin = np.isfinite(npy(ud))
    assert np.array_equal(fin, np.isfinite(npy(depth)))
    assert max_abs(npy(depth)[fin], npy(ud)[fin]) <= 2e-3 * max(1.0, float(np.abs(npy(ud)[fin]).max()))
    H = c["Ho"] // 2
    out = tuple(torch.zeros_like(t) for t in (bf, depth, opac))
    cut = max(1, H // 3)
    eng.render(0, cut, None, out); eng.render(cut, H, None, out)
    for a, b in zip((bf, depth, opac), out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("b,rew,B,Ho,Wo", [(2, False, 1, 64, 80), (2, True, 2, 48, 72), (4, False, 1, 64, 96), (1, True, 1, 10, 14), (2, False, 1, 512, 640)])
def test_merge_vs_oracle_and_torch(b, rew, B, Ho, Wo):
    """N1 (network.py:170-182): gdb_merge against the oracle (exact for the image, ulp-level for the upsampled maps) and
    against the torch calls the reference makes."""
    import torch.nn.functional as F
    frame = synthetic.make_frame(Ho, Wo, V=2, B=B, bundle_size=b, seed=6)
    eng = engine_for(frame, bundle_size=b)
    H, W, Q = Ho // b, Wo // b, eng.Q
    rng = np.random.default_rng(12)
    bf = rng.standard_normal((B * H * W, Q)).astype(np.float32)
    rgb_c = rng.standard_normal((B, 3, Ho, Wo)).astype(np.float32)
    dep = rng.uniform(400, 900, (B * H * W,)).astype(np.float32)
    opa = rng.uniform(0, 1, (B * H * W,)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    img, d, o = eng.merge(t(bf), t(rgb_c), t(dep), t(opa), rew)
    oimg, od, oo = oracle.merge(bf, rgb_c, dep, opa, B, H, W, b, rew)
    assert np.array_equal(npy(img), oimg)
    assert max_abs(npy(d), od) <= 3e-7 * 900 and max_abs(npy(o), oo) <= 3e-7
    nerf_feat = torch.from_numpy(bf).view(B, H, W, -1).permute(0, 3, 1, 2)
    rgb_f = F.pixel_shuffle(nerf_feat[:, :3 * b * b], b)
    ref = torch.from_numpy(rgb_c) + rgb_f
    if rew:
        ref = 0.5 * (ref + rgb_f)
    assert np.array_equal(npy(img), ref.numpy())
    # decoder output absent, maps not requested
    img0, d0, o0 = eng.merge(t(bf))
    assert d0 is None and o0 is None and np.array_equal(npy(img0), rgb_f.numpy())
    with pytest.raises(ValueError, match="bundle_feat"):
        eng.merge(t(bf)[:-1])


@pytest.mark.parametrize("Ho,Wo,b,V,B", [(64, 80, 2, 3, 1), (96, 72, 2, 2, 2), (64, 96, 4, 2, 1), (40, 104, 2, 2, 1), (24, 40, 1, 2, 1)])
def test_prepare_from_fpn_features(Ho, Wo, b, V, B):
    """N3 (network.py:159-164): gdb_prepare_fpn builds the pyramid from the FPN level and the source images; the colour
    channels must equal F.interpolate(src_images -> (H, W)) (oracle.build_img_feat), the rest the pyramid of gdb_prepare."""
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, bundle_size=b, seed=19)
    fpn = np.ascontiguousarray(frame["img_feat"][:, :, :16])
    want_feat = oracle.build_img_feat(fpn, frame["src_images"])
    eng = HotPathEngine(bundle_size=b)
    d = dev_frame(frame)
    d.pop("img_feat")
    d["fpn_feat"] = torch.from_numpy(fpn).cuda()
    eng.prepare(d)
    got = eng.feature_pyramid()
    for bi in range(B):
        want = oracle.build_mips(np.transpose(want_feat[bi], (0, 2, 3, 1)), 3)
        assert len(got) == len(want)
        for l, w in enumerate(want):
            g = npy(got[l][bi])
            assert np.array_equal(g[..., :16].view(np.uint32), w[..., :16].view(np.uint32))   # features: pure data movement + exact averages
            assert max_abs(g[..., 16:], w[..., 16:]) <= 3e-7                                     # colours: resampled in fp32
    with pytest.raises(ValueError, match="not both"):
        eng.prepare({**dev_frame(frame), "fpn_feat": d["fpn_feat"]})


@pytest.mark.parametrize("name,Ho,Wo,V,S,adaptive,scene", [("c3", 640, 960, 3, 3, True, "llff"), ("c3p", 756, 1008, 3, 3, True, "llff"),
                                                           ("c4", 800, 800, 3, 6, True, "nerf"),
                                                           ("c5", 1200, 1600, 5, 6, False, "dtu")])
def test_fused_matches_fp32_chain_at_baseline_sizes(name, Ho, Wo, V, S, adaptive, scene, mode):
    """BASELINE.json configs[2..4] at full size — c3 as the reference's YAML resizes it (640x960) and c3p as BASELINE.json
    words it (1008x756: a 378x504 bundle map, whose odd half-height stops the mip chain after one level) — fused kernel, both
    schedules and precisions, against the fp32 operator chain (oracle-checked above, at c2's full size too) + size-independent
    properties."""
    frame = synthetic.make_frame(Ho, Wo, V=V, scene=scene, seed=1)
    eng = engine_for(frame, synthetic.make_nerf_weights(seed=0), mode, max_num_samples=S, is_adaptive=adaptive)
    bf, depth, opac = eng.render()
    ubf, ud, uo = eng.render_unfused()
    e = max_abs(npy(bf), npy(ubf))
    print(f"{name} {Ho}x{Wo} V{V} S{S}: fused vs fp32 chain max abs err {e:.3e}")
    assert e <= fused_tol(mode, "large")
    assert max_abs(npy(opac), npy(uo)) <= 1e-5
    assert max_abs(npy(depth), npy(ud)) <= 2e-3 * float(ud.abs().max())
    assert float((opac - 1).abs().max()) <= 1e-5            # every bundle has samples: normalised weights sum to one
    assert _psnr_delta(npy(bf), npy(ubf), Ho // 2, Wo // 2) <= 0.05
    assert torch.equal(eng.render()[0], bf)                  # repeatable bit for bit


@pytest.mark.parametrize("offset", [30.0, 100.0])
@pytest.mark.parametrize("sched", [1, 2, 3, 4], ids=["slot-waves", "segment-wave", "dense", "flat"])
def test_fp32_variance_over_views_on_large_magnitude_features(sched, offset):
    """the reference's `torch.var_mean` over the views (nerf.py:73) is two-pass; the fp32 core's one-pass form
    accumulates around a shift (d = g_v - g_0), so its cancellation is relative to the spread of g over the views, not to |g|^2.
    Every view's feature channels carry a common offset of +-30 / +-100 (unnormalised FPN features of a real checkpoint: a large
    mean under an O(1) spread over the views, where sum(g^2) - V mean^2 loses digits): fused fp32 against the oracle, relative to
    the output's scale.  (On the CPU restatement the plain sum-of-squares form moves the output by 8e-7 of its scale at offset 100,
    the shifted form by 3e-7 = the two-pass value; end to end the bound below is set by the fp32 fetch noise.)"""
    frame = synthetic.make_frame(64, 96, V=3, B=1, seed=13)
    c = (offset * np.random.default_rng(3).choice([-1.0, 1.0], size=(1, 1, 16, 1, 1))).astype(np.float32)
    frame["img_feat"][:, :, :16] += c
    w = synthetic.make_nerf_weights(seed=5)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=3, is_adaptive=True)
    eng = engine_for(frame, w, (sched, 1), max_num_samples=3, is_adaptive=True)
    bf = npy(eng.render()[0])
    scale = float(np.abs(obf).max())
    e = max_abs(bf, obf) / scale
    eu = max_abs(npy(eng.render_unfused()[0]), obf) / scale
    print(f"feature offset {offset}, schedule {sched}: output scale {scale:.3g}; fused fp32 vs oracle {e:.2e} of it (operator chain {eu:.2e})")
    assert e <= 1e-5 and eu <= 1e-5


@pytest.mark.parametrize("name,Ho,Wo,V,S,adaptive,scene,precs", [("c3p 756x1008", 756, 1008, 3, 3, True, "llff", (1,)),
                                                                 ("c5 1200x1600 V5", 1200, 1600, 5, 6, False, "dtu", (1, 0))])
def test_c3p_c5_full_size_against_the_oracle(name, Ho, Wo, V, S, adaptive, scene, precs):
    """BASELINE.json configs[2] at its literal size (1008x756: a 378x504 bundle map whose mip chain stops at level 1) and
    configs[4] (1600x1200, 5 views - the config BASELINE.json words as the fp16 MFMA path) DIRECTLY against the oracle, fused
    kernel under GDB_SCHED_AUTO at fp32 (and at f16 for c5), same bounds as c2.  (~25 s of numpy for c5, once.)"""
    frame = synthetic.make_frame(Ho, Wo, V=V, scene=scene, seed=0)
    w = synthetic.make_nerf_weights(seed=0)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=adaptive)
    eng = engine_for(frame, w, None, max_num_samples=S, is_adaptive=adaptive)
    for prec in precs:
        bf, depth, opac = eng.render(precision=prec)
        e = max_abs(npy(bf), obf)
        dpsnr = _psnr_delta(npy(bf), obf, Ho // 2, Wo // 2)
        print(f"{name} vs oracle, precision {('f16', 'f32', 'f32x')[prec]} (auto schedule): max abs err {e:.3e}, PSNR delta {dpsnr:.2e} dB")
        assert e <= (FUSED_TOL if prec == 0 else FUSED_TOL_F32) and dpsnr <= 0.05
        assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5


@pytest.mark.parametrize("name,Ho,Wo,V,S,adaptive,scene,kernel", [("c4 800x800 S6 adaptive", 800, 800, 3, 6, True, "nerf", "k_render_dense"),
                                                                   ("c5 1200x1600 V5 S6 fixed", 1200, 1600, 5, 6, False, "dtu", "k_render_solo")])
def test_smooth_feature_c4_c5_frames_against_the_oracle_and_a_slipped_bias(name, Ho, Wo, V, S, adaptive, scene, kernel):
    """VERDICT r05 task 7: the large frames had only the white-noise bound (5e-4 at fp32: 2 x the coordinate noise), so a 1e-4 slip that
    shows only with five views, S_max 6, the segment-wave kernel (c5) or the ds_bpermute composite of the list kernels (S_max > 4: c4)
    passed.  On CNN-generated features and a smooth volume the fp32 render under GDB_SCHED_AUTO - `kernel` - is held against
    the oracle at SMOOTH_BIG_TOL_F32 (f16: 2e-3), and the same render with lr0.0.bias + 1e-4 on the device must EXCEED that bound.
    Reference: nerf.py:100-113 (the layer), utils.py:34-41 (the composite), bundle_sampler.py:327-359."""
    frame = _smooth_frame(Ho, Wo, V, scene)
    w = synthetic.make_nerf_weights(seed=0)
    w_bad = dict(w)
    w_bad["lr0.0.bias"] = (w["lr0.0.bias"] + np.float32(1e-4)).astype(np.float32)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, max_num_samples=S, is_adaptive=adaptive)
    eng = engine_for(frame, w, None, max_num_samples=S, is_adaptive=adaptive)
    assert eng.render_info(1)["kernel"] == kernel
    for prec, tol in ((1, SMOOTH_BIG_TOL_F32), (0, FUSED_TOL)):
        bf, depth, opac = eng.render(precision=prec)
        e = max_abs(npy(bf), obf)
        print(f"smooth {name} vs oracle, precision {('f16', 'f32')[prec]} ({eng.render_info(prec)['kernel']}): max abs err {e:.3e}, rms {np.sqrt(np.mean((npy(bf) - obf) ** 2)):.3e}")
        assert e <= tol
        assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5
    bad = max_abs(npy(engine_for(frame, w_bad, None, max_num_samples=S, is_adaptive=adaptive).render(precision=1)[0]), obf)
    print(f"smooth {name}: with lr0.0.bias + 1e-4 on the device the fp32 render is {bad:.3e} off the oracle (bound {SMOOTH_BIG_TOL_F32:.0e})")
    assert bad > SMOOTH_BIG_TOL_F32


@pytest.mark.parametrize("B,S,adaptive", [(1, 3, True), (2, 3, True), (1, 6, False), (2, 6, False), (2, 6, True)])
@pytest.mark.parametrize("sched", [3, 4], ids=["dense", "flat"])
def test_prepare_rows_plans_one_strip_and_renders_it_bit_identically(B, S, adaptive, sched):
    """gdb_prepare_rows (ABI v6: the list schedules' plan for ONE rank's row strip) directly: for batch 1
    and 2, adaptive and fixed counts, under both list schedules - prepare(rows=(r0, r1)) then render_packed(r0, r1) equals the same
    rows of a full prepare + render bit for bit (the plan's row mapping with one launch per batch item included), an EMPTY strip is
    accepted, and a render outside the planned strip still matches (the engine sees the strip does not cover it: plan rebuilt)."""
    frame = synthetic.make_frame(64, 80, V=3, B=B, seed=9)
    w = synthetic.make_nerf_weights(seed=3)
    H, W = 32, 40
    full = engine_for(frame, w, (sched, 1), max_num_samples=S, is_adaptive=adaptive).render_packed().clone().view(B, H, W, -1)
    eng = HotPathEngine(max_num_samples=S, is_adaptive=adaptive); eng.set_schedule(sched); eng.precision = 1; eng.load_weights(w)
    eng.strip_reach = True if B == 1 else None   # the strip's reach of the pyramid forced (batch 1) / decided by size (batch 2: whole at this size)
    fr = dev_frame(frame)
    for r0, r1 in ((0, 7), (7, 8), (8, 32), (5, 5)):
        eng.prepare(fr, rows=(r0, r1))
        got = eng.render_packed(r0, r1).view(B, H, W, -1)
        assert torch.equal(got[:, r0:r1], full[:, r0:r1]), (r0, r1)
        assert not got[:, :r0].any() and not got[:, r1:].any()          # rows outside the strip are not written (fresh zero-filled tensor)
    eng.prepare(fr, rows=(8, 16))
    out = eng.render_packed(0, H).view(B, H, W, -1)                     # outside the planned strip: the render rebuilds the plan
    assert torch.equal(out, full)


@pytest.mark.parametrize("prec", [1, 0], ids=["f32", "f16"])
@pytest.mark.parametrize("S,adaptive", [(3, True), (6, False)])
def test_prepare_with_sources_ready_rebuilds_cameras_and_plan_only(prec, S, adaptive):
    """GDB_PREP_SOURCES_READY (ABI v7; HotPathEngine.prepare(sources_unchanged=True)): a sweep of target views over fixed source views
    keeps the feature pyramid(s) and the half-precision image copy of the workspace and rebuilds the camera block and the plan.  The
    render after such a prepare - new target pose, new depth prior, same source tensors - is bit-identical to a fresh engine's full
    prepare + render; the pyramid in the workspace is provably NOT rewritten (a sentinel frame's sources would show); the promise is
    ignored (full prepare) when the source tensors are other storage.  bundle_sampler.py:304-313 (camera terms per call), :355-359."""
    w = synthetic.make_nerf_weights(seed=2)
    a = synthetic.make_frame(96, 128, V=3, seed=4)
    b = synthetic.make_frame(96, 128, V=3, seed=5)
    # frame 2: the SOURCES of a, the target side and the priors of b, the target camera moved
    mixed = dict(a)
    for k in ("depth_range", "vol_range", "feat_volume"):
        mixed[k] = b[k]
    te = a["tar_ext"].copy(); te[:, 0, 3] += 7.5; te[:, 1, 3] -= 3.0
    mixed["tar_ext"] = te
    want = engine_for(mixed, w, (0, prec), max_num_samples=S, is_adaptive=adaptive).render_packed().clone()
    eng = HotPathEngine(max_num_samples=S, is_adaptive=adaptive); eng.precision = prec; eng.load_weights(w)
    da = dev_frame(a)
    eng.prepare(da)
    first = eng.render_packed().clone()
    dm = dict(da)
    for k in ("depth_range", "vol_range", "feat_volume", "tar_ext"):
        dm[k] = torch.from_numpy(np.ascontiguousarray(mixed[k])).cuda()
    eng.prepare(dm, sources_unchanged=True)
    assert eng._src_key is not None
    assert torch.equal(eng.render_packed(), want) and not torch.equal(want, first)
    # the source-only products were NOT rebuilt: overwrite the source tensors' CONTENTS (same storage) and prepare again with the promise
    # - the render still shows the old sources (that is the promise's meaning); a full prepare then shows the new ones
    # (the fp32 kernels read their colour taps from d_src_images itself at render time - only GDB_PREC_F16 reads a copy prepare made)
    da["img_feat"].mul_(0.5)
    if prec == 0:
        da["src_images"].mul_(0.5)
    eng.prepare(dm, sources_unchanged=True)
    assert torch.equal(eng.render_packed(), want)
    eng.prepare(dm)
    assert not torch.equal(eng.render_packed(), want)
    # other storage than last time: the promise is ignored, a full prepare runs
    dm2 = {k: v.clone() for k, v in dm.items()}
    eng.prepare(dm2, sources_unchanged=True)
    half = dict(mixed); half["img_feat"] = (0.5 * mixed["img_feat"]).astype(np.float32)
    if prec == 0:
        half["src_images"] = (0.5 * mixed["src_images"]).astype(np.float32)
    assert torch.equal(eng.render_packed(), engine_for(half, w, (0, prec), max_num_samples=S, is_adaptive=adaptive).render_packed())


def test_prepare_pyr16_without_source_images_is_refused():
    """GDB_PREP_PYR16 also copies d_src_images to half precision (the f16 kernels' colour taps); a frame that carries a
    feature map but no source images would leave that copy uninitialised for a later render told GDB_SCHED_PYR16_READY: GDB_E_BADARG."""
    frame = synthetic.make_frame(64, 80, V=3, seed=1)
    eng = HotPathEngine(); eng.precision = 0
    eng.prepare(dev_frame(frame))
    f = eng._frame
    saved = f.d_src_images
    f.d_src_images = None
    rc = eng.lib.gdb_prepare_ex(C.byref(eng.cfg), C.byref(f), None, _lib.PREP_PYR16, eng._ws.data_ptr(), eng._ws.numel(), None)
    f.d_src_images = saved
    assert rc == -1 and "d_src_images" in eng.lib.gdb_last_error().decode()
    assert eng.lib.gdb_prepare_ex(C.byref(eng.cfg), C.byref(f), None, 64, eng._ws.data_ptr(), eng._ws.numel(), None) == -1   # unknown flag bit
    rc = eng.lib.gdb_prepare_rows(C.byref(eng.cfg), C.byref(f), None, _lib.PREP_STRIP_REACH | _lib.PREP_STRIP_WHOLE, 0, 4, eng._ws.data_ptr(), eng._ws.numel(), None)
    assert rc == -1 and "exclude" in eng.lib.gdb_last_error().decode()


@pytest.mark.parametrize("prec", [1, 0, 2], ids=["f32", "f16", "f32x"])
@pytest.mark.parametrize("b,Ho,Wo,V,B,S,adaptive,inv,scene", [
    (4, 64, 96, 3, 1, 3, True, False, "dtu"),      # the F7d shape: 16 x 24 bundles of 16 rays
    (4, 128, 160, 2, 2, 6, True, True, "nerf"),    # batch 2, disparity sampling, S_max 6 (the ds_bpermute composite)
    (4, 96, 128, 5, 1, 6, False, False, "dtu"),    # five views, fixed counts
    (1, 24, 40, 3, 1, 3, True, False, "dtu"),      # bundle_size 1: one ray per bundle
    (1, 32, 48, 2, 2, 4, False, False, "llff"),
])
def test_fused_bundle_size_1_and_4_vs_oracle(b, Ho, Wo, V, B, S, adaptive, inv, scene, prec):
    """Round 6 (VERDICT r05 "missing 2" / task 5): `nerf.bundle_size` 1 and 4 (configs/dtu_pretrain.yaml:33 "4 for 4*4", network.py:31-34)
    on the FUSED entries - the dense list kernel on the bundles' centre rays, then k_bundle_colours for the 3 b^2 sub-ray colours
    (gdb_render_info: fused 1, schedule 3, one launch more than b = 2).  Against the oracle at the small-frame bounds of b = 2, against
    the operator-mirror chain, packed rows = the three tensors, row strips = the full render bit for bit.
    Reference: bundle_sampler.py:76-120 (any b), :327-337 (the b^2 colours), nerf.py:98,110, utils.py:109-119."""
    frame = synthetic.make_frame(Ho, Wo, V=V, B=B, bundle_size=b, scene=scene, seed=31, src_focal_scale=(1.0, 1.7, 2.9))
    w = synthetic.make_nerf_weights(seed=6)
    with np.errstate(all="ignore"):
        obf, od, oo = oracle.hot_path(frame, w, bundle_size=b, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    eng = HotPathEngine(bundle_size=b, max_num_samples=S, is_adaptive=adaptive, inv_depth=inv)
    eng.precision = prec; eng.load_weights(w); eng.prepare(dev_frame(frame))
    info = eng.render_info()
    assert info["fused"] == 1 and info["schedule"] == 3 and info["kernel"] == "k_render_dense" and info["launches"] in (2, B + 1)
    assert eng.Q == 3 * b * b + 27 and obf.shape[1] == eng.Q
    bf, depth, opac = eng.render()
    e = max_abs(npy(bf), obf)
    ec = max_abs(npy(bf)[:, :3 * b * b], obf[:, :3 * b * b])
    ubf = npy(eng.render_unfused()[0])
    print(f"fused bundle_size {b} {Ho}x{Wo} V{V} S{S} precision {prec}: max abs err vs oracle {e:.3e} (colours {ec:.3e}); operator chain vs oracle {max_abs(ubf, obf):.3e}")
    assert e <= (FUSED_TOL if prec == 0 else FUSED_TOL_F32_SMALL)
    assert max_abs(npy(depth), od) <= 2e-3 * float(np.abs(od).max()) and max_abs(npy(opac), oo) <= 1e-5
    H, W = Ho // b, Wo // b
    packed = eng.render_packed().clone()
    assert torch.equal(packed[:, :eng.Q], bf) and torch.equal(packed[:, eng.Q], depth) and torch.equal(packed[:, eng.Q + 1], opac)
    out = torch.zeros_like(packed)
    for r0, r1 in ((0, 3), (3, 4), (4, H)):
        eng.render_packed(r0, r1, None, out)
    assert torch.equal(out, packed)


@pytest.mark.parametrize("prec", [1, 0], ids=["f32", "f16"])
@pytest.mark.parametrize("scene,fs,S,adaptive", [("dtu", None, 3, True), ("nerf", (1.0, 2.3, 0.6), 6, True), ("llff", None, 4, False)])
def test_prepare_rows_builds_only_the_strips_reach(prec, scene, fs, S, adaptive):
    """gdb_prepare_rows on a partial strip (round 6; SURVEY.md 8(e), VERDICT r05 task 4): only the pyramid tiles / image rows the strip's
    samples can reach are built (k_strip_bounds: the convex body of the strip's rays between the prior's depth extremes, projected into
    every source view, + mip / bilinear margins).  At 512 x 640 in 8 strips (world size 8): every strip's render is bit-identical to the
    same rows of the full-frame render - a tap outside the built region would read the sentinel the workspace is filled with first - and
    less than the whole pyramid is written for an inner strip (the point of it).  Zoomed source cameras and a wide-baseline scene included."""
    Ho, Wo, H = 512, 640, 256
    frame = synthetic.make_frame(Ho, Wo, V=3, scene=scene, seed=3, src_focal_scale=fs)
    w = synthetic.make_nerf_weights(seed=1)
    full = engine_for(frame, w, (0, prec), max_num_samples=S, is_adaptive=adaptive).render_packed().clone().view(H, Wo // 2, -1)
    eng = HotPathEngine(max_num_samples=S, is_adaptive=adaptive); eng.precision = prec; eng.load_weights(w)
    eng.strip_reach = True   # (GDB_PREP_STRIP_REACH: by size the library would build the whole pyramid at 512 x 640 - the bound's launch costs more than it saves there)
    fr = dev_frame(frame)
    eng.prepare(fr)                                                       # (sizes the workspace)
    lay = (C.c_size_t * 7)()
    if prec == 0:
        _lib.check(eng.lib.gdb_pyramid16_layout(C.byref(eng.cfg), C.byref(eng._frame), lay)); nbytes = int(lay[1]) * 3
    else:
        _lib.check(eng.lib.gdb_pyramid_layout(C.byref(eng.cfg), C.byref(eng._frame), lay)); nbytes = 4 * int(lay[1]) * 3
    region = eng._ws[int(lay[0]):int(lay[0]) + nbytes]
    img16 = None
    if prec == 0:   # ... and the half-precision image copy behind the pyramid blocks (gdb_internal.h IMG16_REL), built by image rows
        rel = (int(lay[1]) * 3 + 16 + 255) // 256 * 256
        img16 = eng._ws[int(lay[0]) + rel:int(lay[0]) + rel + 8 * 3 * Ho * Wo]
    written = []
    for r in range(8):
        r0, r1 = 32 * r, 32 * (r + 1)
        region.fill_(0x7F)                                                # 0x7F7F7F7F = 3.4e38 as fp32, 0x7F7F = a NaN as f16
        if img16 is not None:
            img16.fill_(0x7F)
        eng.prepare(fr, rows=(r0, r1))
        assert eng._pyr_partial
        got = eng.render_packed(r0, r1).view(H, Wo // 2, -1)
        assert torch.equal(got[r0:r1], full[r0:r1]), (r0, r1)
        written.append(float((region != 0x7F).float().mean()))
    print(f"partial pyramid, {scene} S{S} precision {prec}: share of the pyramid bytes written per strip of 32 rows: " + " ".join(f"{x:.2f}" for x in written))
    assert min(written) < 0.75
    out = eng.render_packed(0, H).view(H, Wo // 2, -1)                    # rows outside the strip: the engine prepares in full first
    assert not eng._pyr_partial and torch.equal(out, full)
