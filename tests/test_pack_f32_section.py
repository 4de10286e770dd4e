"""Host logic of the fp32-MFMA path (GDB_PREC_F32): the packed A-operand steps of `gdb_pack_weights`, driven by a numpy
emulation of v_mfma_f32_32x32x2_f32 with the device's register conventions (gdb_fused.hip: "f32-MFMA section",
slot_mlp_core_f32), must reproduce the oracle's NeRF MLP (nerf.py:84-115).  Runs on CPU: it pins the fragment
layout — which accumulator register is which feature, which step carries which weight column — before any GPU run.

Conventions emulated (MI355X guide, FP32-input MFMA): lane l = (i = l & 31, h = l >> 5) supplies A[i][k = h] and
B[k = h][j = i]; accumulator register r of lane (j, h) holds row acc_row(r, h) = (r & 3) + 8 (r >> 2) + 4 h of column j.
The two 16-row layers (fc, feat_head + sigma) run on v_mfma_f32_16x16x4_f32 behind v_permlane16_swap_b32 (lane l supplies
A[l & 15][k = l >> 4], B[k = l >> 4][l & 15]; D register r of lane l = row 4 (l >> 4) + r of column l & 15); the emulation is
lane by lane, one wave of 32 samples at a time, so the swaps and the row permutations of the packed section are pinned too.
"""
import ctypes as C

import numpy as np
import pytest

import gdb_oracle as oracle
from gdb_nerf_amd import _lib, synthetic
from gdb_nerf_amd.engine import NERF_KEYS

# enum values of gdb_fused.hip (f32-MFMA section)
Q_VIEW, Q_GVAR, Q_GMEAN, Q_GA, Q_FC, Q_LR0, Q_FH, Q_W0A, Q_W0B, Q_W0C, NQUADS = 0, 1, 4, 6, 9, 11, 17, 21, 37, 43, 49
T32_VIEW = NQUADS * 256
T32_GLOB, T16_FC = T32_VIEW + 32, T32_VIEW + 64
T32_LR0 = T16_FC + 256
T16_FH = T32_LR0 + 64
T32_W0 = T16_FH + 256
T32_AGG, T32_W2 = T32_W0 + 64, T32_W0 + 96
S32_BAGG, S32_BW2 = T32_W2 + 64, T32_W2 + 65
F32SEC_FLOATS = (S32_BW2 + 1 + 63) // 64 * 64


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


# ---- lane-level emulation of one wave: a "register" is a (64,) float32 array, lane l = (j = l & 31, h = l >> 5) --------------
def regs_of(m, nregs):
    """B-operand registers of a finished 32-row accumulator tile m (32 rows, 32 samples): register r of lane (j, h) = m[acc_row(r, h)][j]."""
    return [np.concatenate([m[acc_row(r, 0)], m[acc_row(r, 1)]]).astype(np.float32) for r in range(nregs)]


def tile_of(regs):
    """Inverse of regs_of for a full 16-register accumulator: (32 rows, 32 samples)."""
    m = np.zeros((32, 32), np.float32)
    for r, v in enumerate(regs):
        m[acc_row(r, 0)], m[acc_row(r, 1)] = v[:32], v[32:]
    return m


def table(sec, off):
    """[h][16] bias table -> the 16 accumulator registers it initialises."""
    return [np.concatenate([np.full(32, sec[off + r], np.float32), np.full(32, sec[off + 16 + r], np.float32)]) for r in range(16)]


def quad_elem(sec, q0, st):
    """A operand of step / k-group `st` counted from quad q0: one float per lane."""
    return sec[(q0 + st // 4) * 256:(q0 + st // 4 + 1) * 256].reshape(64, 4)[:, st % 4].astype(np.float32)


def mfma32(a, b, acc):
    """v_mfma_f32_32x32x2_f32: lane (i, h) supplies A[i][k = h] and B[k = h][j = i]; acc = list of 16 registers; k-ordered fmaf chain."""
    m = tile_of(acc)
    for k in range(2):
        A, B = a[32 * k:32 * k + 32], b[32 * k:32 * k + 32]
        m = (m.astype(np.float64) + A[:, None].astype(np.float64) * B[None, :].astype(np.float64)).astype(np.float32)
    return regs_of(m, 16)


def chain32(sec, q0, bregs, acc):
    for s, b in enumerate(bregs):
        acc = mfma32(quad_elem(sec, q0, s), b, acc)
    return acc


def mfma16(a, b, d):
    """v_mfma_f32_16x16x4_f32: lane l supplies A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]; D register r of lane l is
    row 4 (l >> 4) + r of column l & 15.  d = list of 4 registers."""
    D = np.zeros((16, 16), np.float32)
    for r in range(4):
        for l in range(64):
            D[4 * (l >> 4) + r, l & 15] = d[r][l]
    for k in range(4):
        A, B = a[16 * k:16 * k + 16], b[16 * k:16 * k + 16]
        D = (D.astype(np.float64) + A[:, None].astype(np.float64) * B[None, :].astype(np.float64)).astype(np.float32)
    return [np.array([D[4 * (l >> 4) + r, l & 15] for l in range(64)], np.float32) for r in range(4)]


def swap16(a, b):
    """v_permlane16_swap_b32 a, b: odd 16-lane rows of a <-> even rows of b."""
    a, b = a.copy(), b.copy()
    for row in (0, 2):
        t = a[16 * (row + 1):16 * (row + 2)].copy()
        a[16 * (row + 1):16 * (row + 2)] = b[16 * row:16 * (row + 1)]
        b[16 * row:16 * (row + 1)] = t
    return a, b


def swap32(a, b):
    """v_permlane32_swap_b32 a, b: upper half of a <-> lower half of b."""
    a, b = a.copy(), b.copy()
    t = a[32:].copy(); a[32:] = b[:32]; b[:32] = t
    return a, b


def chain16(sec, q0, x, toff):
    """One 16-row layer: register pairs of x swapped into the B operands of samples 0..15 / 16..31, ReLU, D swap.
    Returns (e0, e1): e0[r] / e1[r] of lane (j, h) = ReLU(output row 8h + r / 8h + 4 + r) of sample j."""
    bias = sec[toff:toff + 256].reshape(64, 4)
    d0 = [bias[:, r].astype(np.float32) for r in range(4)]
    d1 = [v.copy() for v in d0]
    for g in range(len(x) // 2):
        xa, xb = swap16(x[2 * g], x[2 * g + 1])
        d0 = mfma16(quad_elem(sec, q0, g), xa, d0)
        d1 = mfma16(quad_elem(sec, q0, g), xb, d1)
    e0, e1 = [], []
    for r in range(4):
        a, b = swap16(relu(d0[r]), relu(d1[r]))
        e0.append(a); e1.append(b)
    return e0, e1


def relu(x):
    return np.maximum(x, np.float32(0))


def emulate_wave(sec, vox, x_in):
    """slot_mlp_core_f32 for the 32 samples of one wave: x_in (V, 32, 35) per-view vectors, vox (32, 8).  Returns the sigma
    pre-activation (32,) and feat (32, 39)."""
    V = x_in.shape[0]
    lane_h = np.arange(64) >> 5
    feat = np.zeros((V, 32, 32), np.float32)                            # rows = staged channel (feat (+) rgb), zeros above 19
    feat[:, :19] = np.transpose(x_in[:, :, 12:31], (0, 2, 1))
    dirs = np.transpose(x_in[:, :, 31:35], (0, 2, 1))                   # (V, 4, 32)

    def tail(v):  # load_tail32: registers 8..11 of half 1 carry dir 0..3, register 11 of half 0 reads dir 0 (meets zero weights)
        fv = regs_of(feat[v], 12)
        for e in range(4):
            fv[8 + e][32:] = dirs[v, e]
        fv[11][:32] = dirs[v, 0]
        return fv
    dregs = lambda v: [np.concatenate([dirs[v, 2 * s], dirs[v, 2 * s + 1]]) for s in range(2)]
    b_view = table(sec, T32_VIEW)

    def view_g(v):
        a = chain32(sec, Q_VIEW, dregs(v), b_view)
        return [tail(v)[i] + relu(a[i]) for i in range(12)]
    g = [view_g(v) for v in range(V)]
    s1 = [sum((g[v][i] for v in range(V)), np.zeros(64, np.float32)) for i in range(12)]
    s2 = [sum((g[v][i] * g[v][i] for v in range(V)), np.zeros(64, np.float32)) for i in range(12)]
    mean = [(s1[i] * np.float32(1.0 / V)).astype(np.float32) for i in range(12)]
    m2 = [((s2[i] - s1[i] * mean[i]) * np.float32(1.0 / (V - 1))).astype(np.float32) for i in range(12)]
    for i in (8, 9, 10):
        m2[i], mean[i] = swap32(m2[i], mean[i])
    base = chain32(sec, Q_GVAR, m2, table(sec, T32_GLOB))
    base = chain32(sec, Q_GMEAN, mean[:8], base)
    w_agg, b_agg = table(sec, T32_AGG), sec[S32_BAGG]
    G = [[relu(r) for r in chain32(sec, Q_GA, g[v], base)] for v in range(V)]
    sc = []
    for v in range(V):
        sp = sum((G[v][i] * w_agg[i] for i in range(16)), np.zeros(64, np.float32))
        sc.append(relu(sp + np.concatenate([sp[32:], sp[:32]]) + b_agg))
    sc = np.stack(sc)
    e = np.exp(sc - sc.max(axis=0)); a_w = (e / e.sum(axis=0)).astype(np.float32)
    agg = [sum((G[v][i] * a_w[v] for v in range(V)), np.zeros(64, np.float32)).astype(np.float32) for i in range(16)]
    e0, e1 = chain16(sec, Q_FC, agg, T16_FC)
    voxT = vox.T.astype(np.float32)                                      # (8, 32)
    hb = e0 + e1 + [np.concatenate([voxT[i], voxT[4 + i]]) for i in range(4)]
    x0 = [relu(r) for r in chain32(sec, Q_LR0, hb, table(sec, T32_LR0))]
    x1 = [relu(r) for r in chain32(sec, Q_LR0 + 3, hb, table(sec, T32_LR0 + 32))]
    X = x0 + x1
    hs = [chain32(sec, Q_W0B + 3 * t, hb, chain32(sec, Q_W0A + 8 * t, X, table(sec, T32_W0 + 32 * t))) for t in range(2)]
    fhv, e1 = chain16(sec, Q_FH, X, T16_FH)
    sig = (e1[0] - e1[1])[:32]
    w2 = [table(sec, T32_W2 + 32 * t) for t in range(2)]
    up = []
    for v in range(V):
        u = np.zeros(64, np.float32)
        for t in range(2):
            hv = chain32(sec, Q_W0C + 3 * t, tail(v), hs[t])
            u = u + sum((relu(hv[i]) * w2[t][i] for i in range(16)), np.zeros(64, np.float32))
        up.append(relu(u + np.concatenate([u[32:], u[:32]]) + sec[S32_BW2]))
    up = np.stack(up)[:, :32]
    e = np.exp(up - up.max(axis=0)); bw = (e / e.sum(axis=0)).astype(np.float32)
    blended = np.einsum("vnc,vn->nc", x_in[:, :, :31], bw)
    fh = np.stack([np.concatenate([fhv[r][:32][None], fhv[r][32:][None]]) for r in range(4)])  # [r][h][j] = channel 4h + r
    fh8 = np.stack([fh[r, h] for h in range(2) for r in range(4)], axis=1)                        # (32, 8)
    return sig, np.concatenate([blended, fh8], axis=1)


def emulate(sec, vox, x_in):
    outs = [emulate_wave(sec, vox[i:i + 32], x_in[:, i:i + 32]) for i in range(0, x_in.shape[1], 32)]
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


@pytest.mark.parametrize("V,viewdir", [(3, True), (2, True), (5, True), (3, False)])
def test_f32_mfma_section_reproduces_the_mlp(V, viewdir):
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
    XLO_FLOATS = 40 * 256  # (N_FRAGS x 256) the low-half fragments of GDB_PREC_F32X follow the f32 section (tests/test_pack_f16_sections.py)
    sec = host[n.value - XLO_FLOATS - F32SEC_FLOATS:n.value - XLO_FLOATS]
    rng = np.random.default_rng(11)
    ns = 64
    x_in = rng.standard_normal((V, ns, 35)).astype(np.float32)
    x_in[:, :, :12] = rng.uniform(0, 1, (V, ns, 12))
    vox = rng.standard_normal((ns, 8)).astype(np.float32)
    sig, feat = emulate(sec, vox, x_in)
    osig, ofeat = oracle.nerf_mlp(w, vox, x_in, viewdir_agg=viewdir)
    assert np.abs(feat - ofeat).max() <= 5e-6 * max(1.0, np.abs(ofeat).max())
    assert np.abs(oracle._softplus(sig[:, None])[:, 0] - osig).max() <= 5e-6 * max(1.0, np.abs(osig).max())
