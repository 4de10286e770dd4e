"""Next row N1: the RDN decoder on the HIP library (gdb_decoder.hip) against the reference's own Decoder
(networks/gdb_nerf/decoder_rdn.py:44-81): the fixture F7 holds the reference module's weights, an input and its output."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, max_abs
from gdb_nerf_amd import _lib, synthetic
from gdb_nerf_amd.engine import HotPathEngine
from gdb_nerf_amd.networks.gdb_nerf.decoder_rdn import Decoder

KEYS = lambda n: (["in_conv.weight", "in_conv.bias"] + [f"blocks.{i}.{k}" for i in range(n) for k in (
    "conv1.weight", "conv2.weight", "conv3.weight", "se.fc.0.weight", "se.fc.2.weight")] + ["up.0.weight", "up.0.bias", "out_conv.weight", "out_conv.bias"])


def _dec_state(f7):
    return {k[len("sd.upsampler."):]: np.asarray(v, dtype=np.float32) for k, v in f7.items() if k.startswith("sd.upsampler.")}  # (the fixture stores exactly-representable tensors as float16)


def _unpack_conv16(packed, cout, cin):
    """Inverse of pack_conv16 (layers of <= 32 output channels, v_mfma_f32_16x16x4_f32 operand order): [tile mt][chunk][tap][g4 2]
    [lane][e 4] -> (cout, cin, 3, 3); element e of lane l = W[16 mt + (l & 15)][32 chunk + 4 (4 g4 + e) + (l >> 4)][tap]; a layer of
    <= 16 channels packs tile 0 only."""
    nchunk, nmt = (cin + 31) // 32, (cout + 15) // 16
    a = packed[:nmt * nchunk * 9 * 2 * 64 * 4].reshape(nmt, nchunk, 9, 2, 64, 4)
    w = np.zeros((cout, cin, 9), np.float32)
    for mt in range(nmt):
        for ch in range(nchunk):
            for g4 in range(2):
                for l in range(64):
                    for e in range(4):
                        co, ci = 16 * mt + (l & 15), 32 * ch + 4 * (4 * g4 + e) + (l >> 4)
                        if co < cout and ci < cin:
                            w[co, ci, :] = a[mt, ch, :, g4, l, e]
                        else:
                            assert np.all(a[mt, ch, :, g4, l, e] == 0)
    return w.reshape(cout, cin, 3, 3)


def _unpack_conv_x(packed, cout, cin, nt):
    """Inverse of pack_conv_x: [tile][16-channel chunk][tap][hi, lo][lane][8 halfs] -> (hi, lo) each (cout, cin, 3, 3) as float32."""
    nchunk = (cin + 31) // 32 * 2
    a = packed[:nt * nchunk * 9 * 2 * 64 * 4].view(np.float16).reshape(nt, nchunk, 9, 2, 64, 8).astype(np.float32)
    w = np.zeros((2, cout, cin, 9), np.float32)
    for t in range(nt):
        for ch in range(nchunk):
            for l in range(64):
                i, h = l & 31, l >> 5
                for e in range(8):
                    co, ci = 32 * t + i, 16 * ch + 8 * h + e
                    if co < cout and ci < cin:
                        w[:, co, ci, :] = a[t, ch, :, :, l, e].T
    return w[0].reshape(cout, cin, 3, 3), w[1].reshape(cout, cin, 3, 3)


def test_packed_decoder_weights_and_the_folded_up_stage():
    """Host logic, no GPU: the operand-order packing is a permutation of the reference's tensors, and the folded 64 -> 12
    convolution (out_conv o PixelShuffle o up-conv, summed in fp64) reproduces the reference's three modules."""
    from gdb_nerf_amd import build
    build.build()
    lib = _lib.load()
    f7 = load_golden("F7_network")
    sd = _dec_state(f7)
    cfg = _lib.GdbConfig(2, 3, 1, 0, 64, 3, 16, 8, 64, 1)
    n = C.c_size_t()
    assert lib.gdb_decoder_packed_floats(C.byref(cfg), 3, C.byref(n)) == 0
    arrs = [np.ascontiguousarray(sd[k], dtype=np.float32) for k in KEYS(3)]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    host = np.full(n.value, np.nan, np.float32)
    assert lib.gdb_pack_decoder_weights(C.byref(cfg), 3, ptrs, host.ctypes.data) == 0
    assert np.isfinite(host).all()
    conv_floats = lambda cin, nt: ((cin + 31) // 32) * 9 * 8 * 64 * 2 * nt
    o = 0
    assert np.array_equal(_unpack_conv16(host[o:], 64, 27), sd["in_conv.weight"]); o += conv_floats(27, 2)
    assert np.array_equal(host[o:o + 64], sd["in_conv.bias"]); o += 64
    for i in range(3):
        assert np.array_equal(_unpack_conv16(host[o:], 32, 64), sd[f"blocks.{i}.conv1.weight"]); o += conv_floats(64, 1)
        assert np.array_equal(_unpack_conv16(host[o:], 32, 96), sd[f"blocks.{i}.conv2.weight"]); o += conv_floats(96, 1)
        assert np.array_equal(_unpack_conv16(host[o:], 64, 128), sd[f"blocks.{i}.conv3.weight"]); o += conv_floats(128, 2)
        assert np.array_equal(host[o:o + 256].reshape(4, 64), sd[f"blocks.{i}.se.fc.0.weight"]); o += 256
        assert np.array_equal(host[o:o + 256].reshape(64, 4), sd[f"blocks.{i}.se.fc.2.weight"]); o += 256
    wf = torch.from_numpy(_unpack_conv16(host[o:], 12, 64)); o += conv_floats(64, 1)
    bf = torch.from_numpy(host[o:o + 12].copy())
    x = torch.randn(2, 64, 9, 11, generator=torch.Generator().manual_seed(0))
    want = F.conv2d(F.pixel_shuffle(F.conv2d(x, torch.from_numpy(sd["up.0.weight"]), torch.from_numpy(sd["up.0.bias"]), padding=1), 2),
                    torch.from_numpy(sd["out_conv.weight"]), torch.from_numpy(sd["out_conv.bias"]))
    y = F.conv2d(x, wf, bf, padding=1)                                # (2, 12, 9, 11), channel 3 s + o, s = dy*2 + dx
    got = y.view(2, 2, 2, 3, 9, 11).permute(0, 3, 4, 1, 5, 2).reshape(2, 3, 18, 22)
    assert max_abs(got.numpy(), want.numpy()) <= 2e-6 * float(want.abs().max())
    o += 32                                                           # folded bias (12 used)
    # the same convolutions again as split-f16 fragments: hi + lo = the fp32 weight to ~22 bits, hi = f16(w)
    fp32 = {"in": (0, 64, 27, 2)}
    q = conv_floats(27, 2) + 64
    for i in range(3):
        for name, (cout, cin, nt) in (("c1", (32, 64, 1)), ("c2", (32, 96, 1)), ("c3", (64, 128, 2))):
            fp32[f"{i}{name}"] = (q, cout, cin, nt); q += conv_floats(cin, nt)
        q += 512
    fp32["up"] = (q, 12, 64, 1)
    for key in ["in"] + [f"{i}{c}" for i in range(3) for c in ("c1", "c2", "c3")] + ["up"]:
        off, cout, cin, nt = fp32[key]
        w = _unpack_conv16(host[off:], cout, cin)
        hi, lo = _unpack_conv_x(host[o:], cout, cin, nt)
        assert np.array_equal(hi, w.astype(np.float16).astype(np.float32)), key
        assert np.all(np.abs(hi.astype(np.float64) + lo - w) <= np.abs(w) * 2.0 ** -21 + 2.0 ** -25), key
        o += conv_floats(cin, nt)
    assert o + 8 * 128 <= n.value
    with pytest.raises(ValueError, match="bundle_size 2"):   # (bundle_size 1 - no up stage - keeps the PyTorch module; 4 is taken since round 6)
        _lib.check(lib.gdb_decoder_packed_floats(C.byref(_lib.GdbConfig(1, 3, 1, 0, 64, 3, 16, 8, 64, 1)), 3, C.byref(n)))
    n4 = C.c_size_t()
    _lib.check(lib.gdb_decoder_packed_floats(C.byref(_lib.GdbConfig(4, 3, 1, 0, 64, 3, 16, 8, 64, 1)), 3, C.byref(n4)))
    assert n4.value == n.value + 4 * (2 * conv_floats(64, 2) + 64)   # upscale_factor 4: + four sub-pixel 64 -> 64 convolutions (fp32 + split-f16 forms, biases)


def _engine(B, H, W, sd, layers=3):
    frame = synthetic.make_frame(2 * H, 2 * W, V=2, B=B, seed=1)
    eng = HotPathEngine()
    eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
    eng.load_decoder_weights(sd, layers)
    return eng


# fp32 MFMA (exact fmaf chains) and the split-f16 convolutions (operand pairs of ~22 bits): both fp32-grade, one bound
PRECS = pytest.mark.parametrize("prec", [1, 2], ids=["f32", "f32x"])


@pytest.mark.gpu
@PRECS
def test_hip_decoder_matches_the_reference_fixture(prec):
    f7 = load_golden("F7_network")
    sd = _dec_state(f7)
    x = f7["dec_in"]                                                     # (1, 27, 32, 48)
    _, _, H, W = x.shape
    eng = _engine(1, H, W, sd)
    for ld in (39, 41):                                                  # the three-tensor and the packed render layouts
        bf = torch.zeros((H * W, ld))
        bf[:, 12:39] = torch.from_numpy(x[0]).permute(1, 2, 0).reshape(H * W, 27)
        bf[:, :12] = 7.0                                                 # the fine-RGB channels (and depth / opacity) are not the decoder's
        got = eng.decode(bf.cuda().contiguous(), precision=prec)
        e = max_abs(got.cpu().numpy(), f7["dec_out"])
        print(f"HIP decoder vs the reference's Decoder (F7, ld {ld}): max abs err {e:.3e} on values up to {np.abs(f7['dec_out']).max():.2f}")
        assert e <= 2e-5 * max(1.0, float(np.abs(f7["dec_out"]).max()))


@pytest.mark.gpu
@PRECS
@pytest.mark.parametrize("B,H,W,layers", [(1, 32, 48, 3), (2, 19, 45, 3), (1, 7, 33, 2), (1, 256, 320, 3), (1, 130, 70, 1), (1, 40, 72, 5), (2, 24, 40, 4)])
def test_hip_decoder_matches_torch_module(B, H, W, layers, prec):
    """Ragged bundle maps (edges inside a 32-pixel tile and inside a row group), batch 2, 1..3 blocks, and the DTU-eval size
    (two rows per wave), against the PyTorch module on the same GPU."""
    torch.manual_seed(3)
    dec = Decoder(27, 3, num_feats=64, num_layers=layers, upscale_factor=2).cuda().eval()
    with torch.no_grad():
        for p in dec.parameters():                                        # biases and gates that matter
            p.mul_(1.5)
    eng = _engine(B, H, W, {k: v.detach() for k, v in dec.state_dict().items()}, layers)
    x = torch.randn(B, 27, H, W, device="cuda")
    bf = torch.zeros((B * H * W, 39), device="cuda")
    bf[:, 12:] = x.permute(0, 2, 3, 1).reshape(B * H * W, 27)
    with torch.no_grad():
        want = dec(x)
    got = eng.decode(bf, precision=prec)
    e, scale = max_abs(got.cpu().numpy(), want.cpu().numpy()), float(want.abs().max())
    print(f"HIP decoder (precision {prec}) vs torch ({B},{H},{W}) x{layers}: max abs err {e:.3e}, output scale {scale:.2f}")
    assert e <= 3e-5 * max(1.0, scale)
    assert torch.equal(got, eng.decode(bf, precision=prec))               # deterministic (two-stage channel means, no atomics)
    with pytest.raises(ValueError, match="bundle_feat"):
        eng.decode(bf[:-1])
    with pytest.raises(ValueError, match="precision"):
        eng.decode(bf, precision=0)


@pytest.mark.gpu
@PRECS
def test_hip_decoder_does_not_depend_on_the_workspace_contents(prec):
    """The workspace is the caller's scratch and arrives uninitialised: the squeeze-excitation's arrival counters are cleared by the
    decode call itself (in_conv's launch), the channel sums and gates are written before they are read.  Garbage (all ones bits,
    then random bytes) in the workspace before the call must leave the result bit-identical, batch 2, ragged map, every block."""
    torch.manual_seed(5)
    dec = Decoder(27, 3, num_feats=64, num_layers=3, upscale_factor=2).cuda().eval()
    B, H, W = 2, 37, 70
    eng = _engine(B, H, W, {k: v.detach() for k, v in dec.state_dict().items()}, 3)
    bf = torch.zeros((B * H * W, 41), device="cuda")
    bf[:, 12:39] = torch.randn(B * H * W, 27, device="cuda")
    want = eng.decode(bf, precision=prec).clone()
    for fill in ("ones", "random"):
        if fill == "ones":
            eng._dec_ws.fill_(0xFF)
        else:
            eng._dec_ws.copy_(torch.randint(0, 256, (eng._dec_ws.numel(),), dtype=torch.uint8, device="cuda"))
        got = eng.decode(bf, precision=prec)
        assert torch.equal(got, want), fill
    with torch.no_grad():
        ref = dec(bf[:, 12:39].view(B, H, W, 27).permute(0, 3, 1, 2).contiguous())
    assert max_abs(want.cpu().numpy(), ref.cpu().numpy()) <= 3e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,layers,prec", [(1, 16, 24, 3, 1), (2, 13, 37, 2, 1), (1, 32, 40, 3, 2)])
def test_hip_decoder_upscale_factor_4_matches_the_module(B, H, W, layers, prec):
    """Round 6 (VERDICT r05 "missing 3"): `Decoder(..., upscale_factor=4)` - the bundle_size 4 network's decoder, two up stages
    (decoder_rdn.py:59-62) - on the HIP library: the first stage as four sub-pixel 64 -> 64 convolutions into a (2H, 2W, 64) map, the
    second folded with out_conv (no non-linearity anywhere in `up`), against the package's PyTorch module (the reference's own class,
    checkpoint-compatible) on random weights and a random (N_b, 77) bundle tensor read in place."""
    torch.manual_seed(3)
    dec = Decoder(27, 3, num_feats=64, num_layers=layers, upscale_factor=4).eval()
    with torch.no_grad():
        for prm in dec.parameters():
            prm.mul_(1.5)
    sd = {k: v.detach().numpy() for k, v in dec.state_dict().items()}
    frame = synthetic.make_frame(4 * H, 4 * W, V=2, B=B, bundle_size=4, seed=1)
    eng = HotPathEngine(bundle_size=4)
    eng.prepare({k: torch.from_numpy(v).cuda() for k, v in frame.items()})
    eng.load_decoder_weights(sd, layers)
    bf = torch.randn(B * H * W, 77)
    with torch.no_grad():
        want = dec(bf[:, 48:75].view(B, H, W, 27).permute(0, 3, 1, 2).contiguous()).numpy()
    got = eng.decode(bf.cuda(), precision=prec).cpu().numpy()
    assert got.shape == want.shape == (B, 3, 4 * H, 4 * W)
    e = max_abs(got, want) / float(np.abs(want).max())
    print(f"HIP decoder, upscale_factor 4, {B}x{H}x{W}, {layers} blocks, precision {prec}: max |delta| / scale = {e:.2e}")
    assert e <= (2e-5 if prec == 1 else 5e-5)
