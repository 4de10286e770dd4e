"""Host logic of the f16-operand paths (GDB_PREC_F16 and the split-f16 GDB_PREC_F32X): the packed A-operand fragments of
`gdb_pack_weights` — the f16 section and the low-half section behind it — driven by a numpy emulation of
v_mfma_f32_32x32x16_f16 with the device's register conventions (gdb_fused.hip: "MFMA section", slot_mlp_core<X>), must
reproduce the oracle's NeRF MLP (nerf.py:84-115): to f16-operand accuracy with the hi fragments alone, to fp32-grade accuracy
with hi + lo pairs and the three-product form lo·hi + hi·lo + hi·hi.  Runs on CPU: it pins which fragment element carries
which weight column / bias, and that hi + lo is the fp32 weight to ~22 bits, before any GPU run.

Conventions emulated: lane l = (r = l & 31, h = l >> 5) supplies A[r][k = 8h + i] and B[k = 8h + i][j = r], i = 0..7; the
accumulator register q of lane (j, h) holds row acc_row(q, h) of column j, so element i of k-step s of a finished tile used as
the next B operand is row vrow(s, h, i) = 16 s + 8 (i >> 2) + 4 h + (i & 3) (the host pre-permutes K to that order).
"""
import ctypes as C

import numpy as np
import pytest

import gdb_oracle as oracle
from gdb_nerf_amd import _lib, synthetic
from gdb_nerf_amd.engine import NERF_KEYS

# enum values of gdb_fused.hip (f16 section, then the low-half section behind the f32 section)
F_VIEW, F_GVAR, F_GMEAN, F_GA, F_FC, F_LR0, F_FH, F_W0A, F_W0B, F_W0C, F_FHB, F_FCA, F_W2R, N_FRAGS = 0, 1, 3, 5, 7, 9, 13, 17, 25, 29, 33, 34, 36, 40
TB_FC = N_FRAGS * 256
TD_AGG, TD_W2 = TB_FC + 32, TB_FC + 64
TS_BAGG, TS_BW2 = TD_W2 + 64, TD_W2 + 65
MFMA_FLOATS = (TS_BW2 + 1 + 63) // 64 * 64
XLO_FLOATS = N_FRAGS * 256


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def vrow(s, h, i):
    return 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def split(x, on):
    """An operand as the device holds it: (hi, lo) with lo = 0 for the plain f16 path."""
    x = np.asarray(x, np.float32)
    hi = f16(x)
    return hi, (f16(x - hi) if on else np.zeros_like(hi))


def a_matrix(sec, idx):
    """Fragment idx as the dense A[32 rows][16 k] it is on the matrix pipe (k = 8h + i)."""
    f = sec[idx * 256:(idx + 1) * 256].view(np.float16).reshape(64, 8).astype(np.float32)
    return np.concatenate([f[:32], f[32:]], axis=1)


def b_step(tile, s):
    """k-step s of a (32, n) tile as a B operand: (16, n) with k = 8h + i."""
    return np.stack([tile[vrow(s, h, i)] for h in range(2) for i in range(8)])


def mm(secs, idx, b, acc, on):
    """acc += A·B for one k-step, operands split; products exact, one fp32 rounding at the end of each MFMA."""
    hi, lo = secs
    a_hi, a_lo = a_matrix(hi, idx), a_matrix(lo, idx)
    b_hi, b_lo = split(b, on)
    acc = acc.astype(np.float64)
    if on:
        acc = (acc + a_lo.astype(np.float64) @ b_hi).astype(np.float32).astype(np.float64)
        acc = (acc + a_hi.astype(np.float64) @ b_lo).astype(np.float32).astype(np.float64)
    return (acc + a_hi.astype(np.float64) @ b_hi).astype(np.float32)


def table(sec, off):
    out = np.zeros(32, np.float32)
    for h in range(2):
        for r in range(16):
            out[acc_row(r, h)] = sec[off + h * 16 + r]
    return out


def relu(x):
    return np.maximum(x, np.float32(0))


def emulate(secs, vox, x_in, on, mfma_sums=False):
    """slot_mlp_core<X = on> for n samples: x_in (V, n, 35) per-view vectors, vox (n, 8) -> sigma pre-activation, feat (n, 39).
    mfma_sums: the GDB_PREC_F16 form of round 5 - the two one-row layers (agg_w_fc, weight.2) and fc run as MFMAs on the ReLU'd
    accumulator of the layer before (fragments F_FCA, F_W2R: the logit comes out in register 8 / 0 of BOTH lane halves, i.e. rows
    16 and 20 / 0 and 4), the softmax-weighted sum over views moves behind fc, g_v is formed on packed halves."""
    hi = secs[0]
    V, n, _ = x_in.shape
    z = np.zeros((32, n), np.float32)
    tv = np.zeros((V, 32, n), np.float32)
    tv[:, :19] = np.transpose(x_in[:, :, 12:31], (0, 2, 1))
    tv[:, 19] = 1.0                                                     # carries view_fc's bias
    tv[:, 24:28] = np.transpose(x_in[:, :, 31:35], (0, 2, 1))
    g = np.zeros((V, 32, n), np.float32)
    for v in range(V):
        a = mm(secs, F_VIEW, b_step(tv[v], 1), z, on)
        g[v, :24] = relu(a[:24])
        g[v, :19] += tv[v, :19]
    mean = g.mean(axis=0, dtype=np.float32)
    var = (np.sum((g - mean) ** 2, axis=0, dtype=np.float32) / np.float32(V - 1)).astype(np.float32)
    var[24] = 1.0                                                       # carries global_fc's bias
    base = z
    for s in range(2):
        base = mm(secs, F_GVAR + s, b_step(var, s), base, on)
    for s in range(2):
        base = mm(secs, F_GMEAN + s, b_step(mean, s), base, on)
    w_agg, b_agg = table(hi, TD_AGG), hi[TS_BAGG]
    G = []
    for v in range(V):
        t = base
        # (mfma_sums: g_v = staged halves + f16(ReLU(view_fc)), summed in f16)
        gv = f16(f16(tv[v]) * (np.arange(32) < 19)[:, None] + f16(relu(mm(secs, F_VIEW, b_step(tv[v], 1), z, on)))) if mfma_sums else g[v]
        for s in range(2):
            t = mm(secs, F_GA + s, b_step(gv, s), t, on)
        G.append(relu(t))
    G = np.stack(G)
    if mfma_sums:
        assert not on
        T = np.stack([mm(secs, F_FCA + 1, b_step(G[v], 1), mm(secs, F_FCA, b_step(G[v], 0), z, on), on) for v in range(V)])
        assert np.array_equal(T[:, 16], T[:, 20]) and not T[:, 17:20].any() and not T[:, 21:].any()   # the logit, in both lane halves
        sc = relu(T[:, 16] + b_agg)
        e = np.exp(sc - sc.max(axis=0)); a_w = (e / e.sum(axis=0)).astype(np.float32)
        im = (table(hi, TB_FC)[:, None] + np.einsum("vrn,vn->rn", T, a_w)).astype(np.float32)
        im[16:] = 0
    else:
        sc = relu(np.einsum("vrn,r->vn", G, w_agg) + b_agg)
        e = np.exp(sc - sc.max(axis=0)); a_w = (e / e.sum(axis=0)).astype(np.float32)
        agg = np.einsum("vrn,vn->rn", G, a_w).astype(np.float32)
        im = np.broadcast_to(table(hi, TB_FC)[:, None], (32, n)).astype(np.float32)
        for s in range(2):
            im = mm(secs, F_FC + s, b_step(agg, s), im, on)
    H0 = b_step(relu(im), 0)
    H1 = np.zeros((16, n), np.float32)
    for h in range(2):
        for i in range(4):
            H1[8 * h + i] = vox[:, 4 * h + i]
    H1[4] = 1.0                                                         # carries the biases of lr0, weight.0, feat_head, sigma
    x = [relu(mm(secs, F_LR0 + 2 * t + 1, H1, mm(secs, F_LR0 + 2 * t, H0, z, on), on)) for t in range(2)]
    X = [b_step(x[t], s) for t in range(2) for s in range(2)]          # X00 X01 X10 X11
    fh = mm(secs, F_FHB, H1, z, on)
    for k in range(4):
        fh = mm(secs, F_FH + k, X[k], fh, on)
    hs = []
    for t in range(2):
        a = z
        for k in range(4):
            a = mm(secs, F_W0A + 4 * t + k, X[k], a, on)
        a = mm(secs, F_W0B + 2 * t, H0, a, on)
        hs.append(mm(secs, F_W0B + 2 * t + 1, H1, a, on))
    w2 = [table(hi, TD_W2 + 32 * t) for t in range(2)]
    up = np.zeros((V, n), np.float32)
    for v in range(V):
        u = z
        for t in range(2):
            hv = mm(secs, F_W0C + 2 * t + 1, b_step(tv[v], 1), mm(secs, F_W0C + 2 * t, b_step(tv[v], 0), hs[t], on), on)
            up[v] += np.einsum("rn,r->n", relu(hv), w2[t])
            for s in range(2):
                u = mm(secs, F_W2R + 2 * t + s, b_step(relu(hv), s), u, on)
        if mfma_sums:
            assert np.array_equal(u[0], u[4]) and not u[1:4].any() and not u[5:].any()   # the logit, in both lane halves
            up[v] = u[0]
    up = relu(up + hi[TS_BW2])
    e = np.exp(up - up.max(axis=0)); bw = (e / e.sum(axis=0)).astype(np.float32)
    blended = np.einsum("vnc,vn->nc", x_in[:, :, :31], bw)
    return fh[8], np.concatenate([blended, relu(fh[:8]).T], axis=1)


def packed(viewdir):
    from gdb_nerf_amd import build
    build.build()
    lib = _lib.load()
    w = synthetic.make_nerf_weights(seed=3)
    cfg = _lib.GdbConfig(2, 3, 1, 0, 64, 3, 16, 8, 64, int(viewdir))
    n = C.c_size_t()
    assert lib.gdb_packed_weight_floats(C.byref(cfg), C.byref(n)) == 0
    host = np.zeros(n.value, np.float32)
    arrs = [np.ascontiguousarray(w[k + s]) for k in NERF_KEYS for s in (".weight", ".bias")]
    ptrs = (C.c_void_p * 18)(*[a.ctypes.data for a in arrs])
    assert lib.gdb_pack_weights(C.byref(cfg), ptrs, host.ctypes.data) == 0
    return w, host


@pytest.mark.parametrize("V,viewdir", [(3, True), (2, True), (5, True), (3, False)])
def test_f16_sections_reproduce_the_mlp(V, viewdir):
    w, host = packed(viewdir)
    n = host.size
    lo = host[n - XLO_FLOATS:]
    # layout: [fp32 section][f16 section MFMA_FLOATS][f32 section][low halves XLO_FLOATS]; the f32 section's size comes from
    # tests/test_pack_f32_section.py
    from test_pack_f32_section import F32SEC_FLOATS
    hi = host[n - XLO_FLOATS - F32SEC_FLOATS - MFMA_FLOATS:n - XLO_FLOATS - F32SEC_FLOATS]
    rng = np.random.default_rng(11)
    ns = 96
    x_in = rng.standard_normal((V, ns, 35)).astype(np.float32)
    x_in[:, :, :12] = rng.uniform(0, 1, (V, ns, 12))
    vox = rng.standard_normal((ns, 8)).astype(np.float32)
    osig, ofeat = oracle.nerf_mlp(w, vox, x_in, viewdir_agg=viewdir)
    err = {}
    for on in (False, True, "mfma_sums"):
        sig, feat = emulate((hi, lo), vox, x_in, on is True, mfma_sums=on == "mfma_sums")
        err[on] = (np.abs(feat - ofeat).max() / max(1.0, np.abs(ofeat).max()),
                   np.abs(oracle._softplus(sig[:, None])[:, 0] - osig).max() / max(1.0, np.abs(osig).max()))
    print(f"V={V} viewdir={viewdir}: f16 operands {err[False]}, with the MFMA sums of round 5 {err['mfma_sums']}, split-f16 pairs {err[True]}")
    assert max(err[False]) <= 1e-3          # f16 operands: the layout is right, the precision is f16's
    assert max(err["mfma_sums"]) <= 1e-3    # ... and so is the form GDB_PREC_F16 runs since round 5
    assert max(err[True]) <= 1e-6           # hi + lo pairs: fp32-grade, the bound the f32-MFMA section meets


def test_low_halves_complete_the_f16_fragments():
    """Element by element: lo is the f16 of what hi = f16(w) left over (|lo| <= half an ulp of hi, lo = 0 where hi is exact), and
    for a fragment whose map is simple (fc: row r, column vrow(s, h, i)) hi + lo is the fp32 weight to 2^-21."""
    from test_pack_f32_section import F32SEC_FLOATS
    w, host = packed(True)
    n = host.size
    lo = host[n - XLO_FLOATS:].view(np.float16).astype(np.float64)
    hi = host[n - XLO_FLOATS - F32SEC_FLOATS - MFMA_FLOATS:n - XLO_FLOATS - F32SEC_FLOATS][:N_FRAGS * 256].view(np.float16).astype(np.float64)
    assert hi.size == lo.size == N_FRAGS * 512
    ulp = np.spacing(np.abs(hi).astype(np.float16)).astype(np.float64)
    assert np.all(np.abs(lo) <= 0.5 * ulp + 1e-30) and np.all(lo[hi == 0] == 0)
    assert np.count_nonzero(lo) > 0.9 * np.count_nonzero(hi)          # random fp32 weights are (almost) never exact in f16
    W = w["fc.0.weight"]
    for s in range(2):
        fh = hi[(F_FC + s) * 512:(F_FC + s + 1) * 512].reshape(64, 8)
        fl = lo[(F_FC + s) * 512:(F_FC + s + 1) * 512].reshape(64, 8)
        for l in range(64):
            r, h = l & 31, l >> 5
            for i in range(8):
                want = float(W[r, vrow(s, h, i)]) if r < W.shape[0] else 0.0
                assert abs(fh[l, i] + fl[l, i] - want) <= abs(want) * 2.0 ** -21 + 2.0 ** -25
