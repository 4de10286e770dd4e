"""Host logic of the fp32-MFMA path (GDB_PREC_F32): the packed A-operand steps of `gdb_pack_weights`, driven by a numpy
emulation of v_mfma_f32_32x32x2_f32 with the device's register conventions (gdb_fused.hip: "f32-MFMA section",
slot_mlp_core_f32), must reproduce the oracle's NeRF MLP (nerf.py:84-115).  Runs on CPU: it pins the fragment
layout — which accumulator register is which feature, which step carries which weight column — before any GPU run.

Conventions emulated (MI355X guide, FP32-input MFMA): lane l = (i = l & 31, h = l >> 5) supplies A[i][k = h] and
B[k = h][j = i]; accumulator register r of lane (j, h) holds row acc_row(r, h) = (r & 3) + 8 (r >> 2) + 4 h of column j.
"""
import ctypes as C

import numpy as np
import pytest

import gdb_oracle as oracle
from gdb_nerf_amd import _lib, synthetic
from gdb_nerf_amd.engine import NERF_KEYS

# enum values of gdb_fused.hip (f32-MFMA section)
Q_VIEW, Q_GVAR, Q_GMEAN, Q_GA, Q_FC, Q_LR0, Q_FH, Q_W0A, Q_W0B, Q_W0C, NQUADS = 0, 1, 4, 7, 10, 14, 20, 28, 44, 50, 58
T32_VIEW = NQUADS * 256
T32_GLOB, T32_FC, T32_LR0 = T32_VIEW + 32, T32_VIEW + 64, T32_VIEW + 96
T32_FH, T32_W0, T32_AGG, T32_W2 = T32_LR0 + 64, T32_LR0 + 96, T32_LR0 + 160, T32_LR0 + 192
S32_BAGG, S32_BW2 = T32_W2 + 64, T32_W2 + 65
F32SEC_FLOATS = (S32_BW2 + 1 + 63) // 64 * 64


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def regs_of(m, nregs):
    """B-operand registers of a finished 32-row accumulator tile m (32, n): register r = [h][j]."""
    return [np.stack([m[acc_row(r, 0)], m[acc_row(r, 1)]]) for r in range(nregs)]


def table(sec, off):
    out = np.zeros(32, np.float32)
    for h in range(2):
        for r in range(16):
            out[acc_row(r, h)] = sec[off + h * 16 + r]
    return out


def chain(sec, q0, bregs, acc):
    """acc (32, n) += sum over steps of A_step · B_step, each product rounded to fp32 like the MFMA's fmaf chain."""
    acc = acc.astype(np.float32).copy()
    for s, b in enumerate(bregs):
        a = sec[(q0 + s // 4) * 256:(q0 + s // 4 + 1) * 256].reshape(64, 4)[:, s % 4].reshape(2, 32)  # [h][i]
        for k in range(2):
            acc = (acc.astype(np.float64) + a[k][:, None].astype(np.float64) * b[k][None, :].astype(np.float64)).astype(np.float32)
    return acc


def relu(x):
    return np.maximum(x, np.float32(0))


def emulate(sec, vox, x_in):
    """slot_mlp_core_f32 for n samples: x_in (V, n, 35) per-view vectors, vox (n, 8).  Returns sigma pre-activation, feat (n, 39)."""
    V, n, _ = x_in.shape
    col = lambda t: np.broadcast_to(table(sec, t)[:, None], (32, n)).astype(np.float32)
    feat = np.zeros((V, 32, n), np.float32)
    feat[:, :19] = np.transpose(x_in[:, :, 12:31], (0, 2, 1))          # rows = feat ⊕ rgb channel, zeros above 19
    dirs = np.transpose(x_in[:, :, 31:35], (0, 2, 1))                  # (V, 4, n)
    dregs = lambda v: [np.stack([dirs[v, 2 * s], dirs[v, 2 * s + 1]]) for s in range(2)]
    g = np.zeros((V, 32, n), np.float32)
    for v in range(V):
        a = chain(sec, Q_VIEW, dregs(v), col(T32_VIEW))
        g[v, :24] = feat[v, :24] + relu(a[:24])                        # registers 0..11 cover rows 0..23
    mean = g.mean(axis=0, dtype=np.float32)
    var = (np.sum((g - mean) ** 2, axis=0, dtype=np.float32) / np.float32(V - 1)).astype(np.float32)
    base = chain(sec, Q_GVAR, regs_of(var, 12), col(T32_GLOB))
    base = chain(sec, Q_GMEAN, regs_of(mean, 12), base)
    w_agg, b_agg = table(sec, T32_AGG), sec[S32_BAGG]
    G = np.stack([relu(chain(sec, Q_GA, regs_of(g[v], 12), base)) for v in range(V)])
    sc = relu(np.einsum("vrn,r->vn", G, w_agg) + b_agg)
    e = np.exp(sc - sc.max(axis=0)); a_w = (e / e.sum(axis=0)).astype(np.float32)
    agg = np.einsum("vrn,vn->rn", G, a_w).astype(np.float32)
    im = relu(chain(sec, Q_FC, regs_of(agg, 16), col(T32_FC)))
    voxT = vox.T.astype(np.float32)                                      # (8, n)
    hb = regs_of(im, 8) + [np.stack([voxT[i], voxT[4 + i]]) for i in range(4)]
    x0 = relu(chain(sec, Q_LR0, hb, col(T32_LR0)))
    x1 = relu(chain(sec, Q_LR0 + 3, hb, col(T32_LR0 + 32)))
    X = regs_of(x0, 16) + regs_of(x1, 16)
    fh = chain(sec, Q_FH, X, col(T32_FH))
    hs = [chain(sec, Q_W0B + 3 * t, hb, chain(sec, Q_W0A + 8 * t, X, col(T32_W0 + 32 * t))) for t in range(2)]
    w2 = [table(sec, T32_W2 + 32 * t) for t in range(2)]
    up = np.zeros((V, n), np.float32)
    for v in range(V):
        tb = regs_of(feat[v], 12) + dregs(v)
        for t in range(2):
            up[v] += np.einsum("rn,r->n", relu(chain(sec, Q_W0C + 4 * t, tb, hs[t])), w2[t])
    up = relu(up + sec[S32_BW2])
    e = np.exp(up - up.max(axis=0)); bw = (e / e.sum(axis=0)).astype(np.float32)
    blended = np.einsum("vnc,vn->nc", x_in[:, :, :31], bw)
    return fh[8], np.concatenate([blended, relu(fh[:8]).T], axis=1)


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
    XLO_FLOATS = 34 * 256  # the low-half fragments of GDB_PREC_F32X follow the f32 section (tests/test_pack_f16_sections.py)
    sec = host[n.value - XLO_FLOATS - F32SEC_FLOATS:n.value - XLO_FLOATS]
    rng = np.random.default_rng(11)
    ns = 96
    x_in = rng.standard_normal((V, ns, 35)).astype(np.float32)
    x_in[:, :, :12] = rng.uniform(0, 1, (V, ns, 12))
    vox = rng.standard_normal((ns, 8)).astype(np.float32)
    sig, feat = emulate(sec, vox, x_in)
    osig, ofeat = oracle.nerf_mlp(w, vox, x_in, viewdir_agg=viewdir)
    assert np.abs(feat - ofeat).max() <= 5e-6 * max(1.0, np.abs(ofeat).max())
    assert np.abs(oracle._softplus(sig[:, None])[:, 0] - osig).max() <= 5e-6 * max(1.0, np.abs(osig).max())
