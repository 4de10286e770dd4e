// Production kernel: the whole hot-path section of Network.forward (network.py:145-169 of the
// reference: build_rays → sample → encode → NeRF MLP → normalised alpha composite) in ONE
// launch, with no per-sample intermediate in HBM.
//
// Mapping (gfx950, wave64):
//   workgroup  = one 32-bundle segment of a bundle-map row; wave w handles sample slots
//                k = w, w+nw, ... of those bundles (one slot per wave when S_max fits).
//   lane       = (j = lane&31: bundle of the segment, h = lane>>5: "half").  The two halves
//                of a wave split every per-sample vector exactly the way the 32x32 MFMA
//                accumulator splits its rows: half h owns rows 4h..4h+3 of every 8.
//   gather     = coalesced along the row: the NCDHW cost volume and the NCHW images are read
//                along x by consecutive lanes; feature texels (20 floats, channel-last
//                pyramid) are read as 16-B chunks, chunk parity = h.
//   MLP        = every layer is D = W · Xᵀ on v_mfma_f32_32x32x16_f16 (samples on the MFMA
//                column = lane, features on the accumulator rows).  A layer's accumulator is
//                converted in place to the next layer's B operand (weights are pre-permuted
//                on the host to the accumulator's row order), so activations never leave
//                registers; views are separate accumulators of the same 32 samples, which
//                makes variance/mean/softmax over views per-lane arithmetic.
//   composite  = per-slot results meet in LDS; transmittance weights, normalisation and the
//                weighted sum run per bundle; the (N_b, 39) output rows are written as one
//                contiguous run per segment.
#include "gdb_internal.h"
#if !defined(GDB_DIAG) && (defined(GDB_XP_NOW) || defined(GDB_XP_PK))
#error "GDB_XP_* timing experiments produce wrong results by design: they exist in the diagnostic build (-DGDB_DIAG) only"
#endif
#include <cstdlib>
#include <type_traits>
#include <cstring>

int gdb_fail(int code, const char* fmt, ...);
int gdb_check_cfg(const GdbConfig* c);
int gdb_check_frame(const GdbConfig* c, const GdbFrame* f, bool need_ptrs);
int gdb_build_dense_plan(const GdbConfig* cfg, const GdbFrame* f, void* ws, hipStream_t st);
int gdb_build_pyr16(const GdbConfig* cfg, const GdbFrame* f, void* ws, hipStream_t st);

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- MFMA section of the packed weights ---------------------------------------------------------
// 33 A-operand fragments (64 lanes x 8 halfs = 256 floats each), then fp32 tables.
enum {
    F_VIEW = 0,   // view_fc, k-step 1 of the tail vector (dir)
    F_GVAR = 1,   // global_fc columns [19,38) on var, k-steps 0,1
    F_GMEAN = 3,  // global_fc columns [38,57) on mean
    F_GA = 5,     // global_fc columns [0,19) on g_v
    F_FC = 7,     // fc on the aggregated 32-vector
    F_LR0 = 9,    // lr0: [out tile][k-step: 0 = im, 1 = vox]
    F_FH = 13,    // rows 0..7 feat_head, row 8 sigma, on x: [x tile][k-step]
    F_W0A = 17,   // weight.0 columns [0,64) on x: [out tile][x tile][k-step]
    F_W0B = 25,   // weight.0 columns [64,88) on [vox|im]: [out tile][k-step]
    F_W0C = 29,   // weight.0 columns [88,111) on the per-view tail: [out tile][k-step]
    F_FHB = 33,   // biases of feat_head / sigma, on the constant-one slot of the [vox | im] operand
    // GDB_PREC_F16 only (round 5): the two one-row layers as MFMAs on the ReLU'd accumulator of the layer before (the f16 kernels are
    // bound by vector instructions, the matrix pipe idles: a 32-term dot product per lane is 32 of them, the operand conversion 16)
    F_FCA = 34,   // rows 0..15 fc, rows 16 and 20 agg_w_fc (register 8 of BOTH lane halves), on ReLU(G_v): k-steps 0, 1
    F_W2R = 36,   // rows 0 and 4 weight.2 (register 0 of both lane halves), on ReLU(weight.0's output): [tile][k-step]
    N_FRAGS = 40
};
// Biases ride inside the MFMAs wherever an operand has a spare K slot: that slot of the B operand is set to 1.0
// and the matching column of the weight fragment holds the bias (view_fc: tail slot 19; global_fc: slot 24 of the
// variance operand; lr0 / weight.0 / feat_head / sigma: element 4 of the [vox | im] k-step-1 operand).  Only fc,
// whose 32-wide input has no spare slot, keeps a bias table.
enum {
    TB_FC = N_FRAGS * 256,    // fc bias in accumulator layout [h][16]
    TD_AGG = TB_FC + 32,      // agg_w_fc weights in accumulator layout
    TD_W2 = TD_AGG + 32,      // weight.2 weights [tile][h][16]
    TS_BAGG = TD_W2 + 64,
    TS_BW2 = TS_BAGG + 1,
    MFMA_FLOATS = (TS_BW2 + 1 + 63) / 64 * 64
};

// ---- f32-MFMA section of the packed weights (precision GDB_PREC_F32), appended to the f16 section ---------------
// v_mfma_f32_32x32x2_f32 takes ONE f32 per lane for A and for B: lane (i = l & 31, h = l >> 5) supplies A[i][k = h]
// and B[k = h][j = i]; C/D use the 32x32 accumulator map.  So an accumulator register r of a finished layer IS a B
// operand: it carries features acc_row(r, 0) (half 0) and acc_row(r, 1) (half 1) of the lane's sample, and the next
// layer runs one MFMA per register with no conversion and no lane movement.  A "step" is one MFMA's A operand (64
// floats, W[out row i][k of (step, h)]); four consecutive steps are stored as one float4 per lane (a "quad", 1 KiB).
//
// The two layers with <= 16 output rows (fc 32 -> 16; feat_head + sigma 64 -> 9) run on v_mfma_f32_16x16x4_f32 instead
// (same FLOP rate, half the padding: 32 cycles per 16 rows x 16 samples x 4 k).  Lane l supplies A[row l & 15][k = l >> 4]
// and B[k = l >> 4][sample l & 15]; D register r of lane l is row 4 (l >> 4) + r of sample l & 15.  A "k-group" g of such
// a layer takes the input register PAIR (2g, 2g + 1): v_permlane16_swap turns the pair into the B operands of samples
// 0..15 and 16..31, whose k order is [reg 2g half 0, reg 2g+1 half 0, reg 2g half 1, reg 2g+1 half 1]; a second swap of
// the two D tiles leaves lane (j, h) with output rows 8h + r and 8h + 4 + r of its own sample, and the rows are permuted
// on the host so that those are the features the next layer's operand wants there.  Four k-groups = one quad.
enum {
    Q_VIEW = 0,    // view_fc on dir: 2 steps (step s carries dir 2s + h)
    Q_GVAR = 1,    // global_fc columns [19,38) on var: 12 steps (register r of the 19-vector in accumulator layout); the slots
                   // (r = 8..10, half 1) - channels 20..22, which do not exist - carry the MEAN columns of channels 16..18
    Q_GMEAN = 4,   // global_fc columns [38,54) on mean channels 0..15: 8 steps
    Q_GA = 6,      // global_fc columns [0,19) on g_v: 12 steps
    Q_FC = 9,      // fc on the aggregated 32-vector: 8 k-groups (16x16x4 form)
    Q_LR0 = 11,    // [out tile 2][3 quads]: 8 steps on im (registers 0..7), then 4 on vox (step i carries channel i + 4h)
    Q_FH = 17,     // rows feat_head 0..7, +sigma, -sigma on x: 16 k-groups (16x16x4 form)
    Q_W0A = 21,    // weight.0 columns [0,64) on x: [out tile 2][8 quads]
    Q_W0B = 37,    // weight.0 columns [64,88) on [vox | im]: [out tile 2][3 quads], step order as Q_LR0
    Q_W0C = 43,    // weight.0 columns [88,111) on the per-view tail: [out tile 2][3 quads]: 12 steps on the tail registers
                   // (feat (+) rgb in accumulator layout, with dir 0..3 in the slots (r = 8..11, half 1))
    NQUADS = 49
};
enum {  // fp32 tables: [h][16] in 32x32 accumulator layout (bias tables initialise a layer's accumulator), [lane][4] for the
        // 16x16x4 layers (T16_*), then scalars
    T32_VIEW = NQUADS * 256, T32_GLOB = T32_VIEW + 32, T16_FC = T32_GLOB + 32, T32_LR0 = T16_FC + 256 /* 2 tiles */,
    T16_FH = T32_LR0 + 64, T32_W0 = T16_FH + 256 /* 2 tiles */, T32_AGG = T32_W0 + 64, T32_W2 = T32_AGG + 32 /* 2 tiles */,
    S32_BAGG = T32_W2 + 64, S32_BW2 = S32_BAGG + 1,
    F32SEC_FLOATS = (S32_BW2 + 1 + 63) / 64 * 64
};

// ---- low halves of the f16 fragments (precision GDB_PREC_F32X), appended to the f32 section --------------------
// The split-f16 path represents every MFMA operand as hi + lo with hi = f16(x), lo = f16(x - hi) (about 22 bits) and forms
// a product as lo·hi + hi·lo + hi·hi on the f16 matrix pipe with fp32 accumulation.  Its hi weight fragments ARE the f16
// section's fragments; this section holds the N_FRAGS matching lo fragments (same indices, same element order).  The fp32
// tables (TB_FC, TD_AGG, TD_W2, scalars) of the f16 section are shared.
enum { XLO_OFF = MFMA_FLOATS + F32SEC_FLOATS, XLO_FLOATS = N_FRAGS * 256 };

size_t gdb_mfma_section_floats() { return (size_t)MFMA_FLOATS + F32SEC_FLOATS + XLO_FLOATS; }

// Accumulator row of (register r, half h) in a 32x32 MFMA tile; also the k index that element
// (r & 7) of k-step (r >> 3) carries when the tile is reused as a B operand.
static inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

namespace {
struct Packer {
    const float* w;  // fp32 section
    float* out;      // MFMA section
    float* lo = nullptr;  // low-half fragments (XLO section) or null
    // element e of fragment idx: hi = f16(v) into the f16 section, lo = f16(v - hi) into the low-half section
    void put(int idx, int e, float v) {
        const _Float16 hi = (_Float16)v;
        ((_Float16*)(out + (size_t)idx * 256))[e] = hi;
        if (lo) ((_Float16*)(lo + (size_t)idx * 256))[e] = (_Float16)(v - (float)hi);
    }
    // kmap(h, i) -> column of W (or -1 for a zero), rows r -> output feature rowmap(r) (or -1)
    // biasOff >= 0: element (bh, bi) of every row carries that row's bias (the B operand has 1.0 there)
    template <class RowMap, class KMap>
    void frag(int idx, int wOff, int ld, RowMap rowmap, KMap kmap, int biasOff = -1, int bh = 0, int bi = 0) {
        for (int l = 0; l < 64; ++l) {
            int r = l & 31, h = l >> 5, orow = rowmap(r);
            for (int i = 0; i < 8; ++i) {
                int col = kmap(h, i);
                float v = (orow >= 0 && col >= 0) ? w[wOff + orow * ld + col] : 0.f;
                if (biasOff >= 0 && h == bh && i == bi) v = orow >= 0 ? w[biasOff + orow] : 0.f;
                put(idx, l * 8 + i, v);
            }
        }
    }
    template <class RowMap>
    void table(int off, int srcOff, int stride, RowMap rowmap) {
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                int o = rowmap(acc_row(r, h));
                out[off + h * 16 + r] = o >= 0 ? w[srcOff + o * stride] : 0.f;
            }
    }
};
}  // namespace

void gdb_pack_mfma_section(const float* fp32, float* out) {
    Packer p{fp32, out, out + XLO_OFF};
    memset(out, 0, sizeof(float) * MFMA_FLOATS);
    memset(out + XLO_OFF, 0, sizeof(float) * XLO_FLOATS);
    auto vrow = [](int s, int h, int i) { return 16 * s + 8 * (i >> 2) + 4 * h + (i & 3); };
    auto lt = [](int n) { return [n](int r) { return r < n ? r : -1; }; };
    auto tile = [](int t, int n) { return [t, n](int r) { return 32 * t + r < n ? 32 * t + r : -1; }; };
    // tail vector tv[32]: [0,19) feat ⊕ rgb, [24,28) dir  -> tail column
    auto tvcol = [&](int s) { return [=](int h, int i) { int k = vrow(s, h, i); return k < GDB_CFR ? k : (k >= 24 && k < 28 ? GDB_CFR + k - 24 : -1); }; };
    auto c19 = [&](int s, int base) { return [=](int h, int i) { int k = vrow(s, h, i); return k < GDB_CFR ? base + k : -1; }; };
    // view_fc reads only dir: W_view column d sits at tv[24+d]
    p.frag(F_VIEW, PW_VIEW_W, 4, lt(GDB_CFR), [&](int h, int i) { int k = vrow(1, h, i); return (k >= 24 && k < 28) ? k - 24 : -1; },
           PW_VIEW_B, 0, 3);  // tv[19] = vrow(1,0,3) carries 1.0
    for (int s = 0; s < 2; ++s) {
        if (s == 0) p.frag(F_GVAR + s, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19(s, GDB_CFR));
        else p.frag(F_GVAR + s, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19(s, GDB_CFR), PW_GLOB_B, 0, 4);  // var slot 24 = vrow(1,0,4) carries 1.0
        p.frag(F_GMEAN + s, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19(s, 2 * GDB_CFR));
        p.frag(F_GA + s, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19(s, 0));
        p.frag(F_FC + s, PW_FC_W, GDB_GF, lt(GDB_IM), [=](int h, int i) { return vrow(s, h, i); });
    }
    // [vox | im] vector: k-step 0 = im (16, accumulator order), k-step 1 = vox (element i<4 of half h = channel 4h+i)
    auto hcol = [&](int s, int base) {
        return [=](int h, int i) { return s == 0 ? base + GDB_CV + vrow(0, h, i) : (i < 4 ? base + 4 * h + i : -1); };
    };
    for (int ot = 0; ot < 2; ++ot)
        for (int s = 0; s < 2; ++s) {
            // element 4 of half 0 of the [vox | im] k-step-1 operand carries 1.0
            p.frag(F_LR0 + 2 * ot + s, PW_LR0_W, GDB_HD, tile(ot, GDB_HID), hcol(s, 0), s == 1 ? PW_LR0_B : -1, 0, 4);
            p.frag(F_W0B + 2 * ot + s, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID), hcol(s, GDB_HID), s == 1 ? PW_W0_B : -1, 0, 4);
            p.frag(F_W0C + 2 * ot + s, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID),
                   [&, s](int h, int i) { int c = tvcol(s)(h, i); return c >= 0 ? GDB_HID + GDB_HD + c : -1; });
            for (int xt = 0; xt < 2; ++xt)
                p.frag(F_W0A + 4 * ot + 2 * xt + s, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID), [=](int h, int i) { return 32 * xt + vrow(s, h, i); });
        }
    // rows 0..7 = feat_head, row 8 = sigma; both read x
    for (int xt = 0; xt < 2; ++xt)
        for (int s = 0; s < 2; ++s) {
            for (int l = 0; l < 64; ++l) {
                int r = l & 31, h = l >> 5;
                for (int i = 0; i < 8; ++i) {
                    int col = 32 * xt + vrow(s, h, i);
                    float v = r < GDB_CV ? fp32[PW_FH_W + r * GDB_HID + col] : (r == GDB_CV ? fp32[PW_SIG_W + col] : 0.f);
                    p.put(F_FH + 2 * xt + s, l * 8 + i, v);
                }
            }
        }
    p.table(TB_FC, PW_FC_B, 1, lt(GDB_IM));
    for (int ot = 0; ot < 2; ++ot) p.table(TD_W2 + 32 * ot, PW_W2_W, 1, tile(ot, GDB_HID));
    {   // F_FHB: rows 0..7 feat_head bias, row 8 sigma bias, at element 4 of half 0
        for (int l = 0; l < 64; ++l) {
            int r = l & 31, h = l >> 5;
            for (int i = 0; i < 8; ++i) {
                float v = (h == 0 && i == 4) ? (r < GDB_CV ? fp32[PW_FH_B + r] : (r == GDB_CV ? fp32[PW_SIG_B] : 0.f)) : 0.f;
                p.put(F_FHB, l * 8 + i, v);
            }
        }
    }
    p.table(TD_AGG, PW_AGG_W, 1, lt(GDB_GF));
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l) {
            const int r = l & 31, h = l >> 5;
            for (int i = 0; i < 8; ++i) {
                const int col = vrow(s, h, i);
                p.put(F_FCA + s, l * 8 + i, r < GDB_IM ? fp32[PW_FC_W + r * GDB_GF + col] : ((r == 16 || r == 20) ? fp32[PW_AGG_W + col] : 0.f));
                for (int t = 0; t < 2; ++t) p.put(F_W2R + 2 * t + s, l * 8 + i, (r == 0 || r == 4) ? fp32[PW_W2_W + 32 * t + col] : 0.f);
            }
        }
    out[TS_BAGG] = fp32[PW_AGG_B];
    out[TS_BW2] = fp32[PW_W2_B];

    // ---- f32-MFMA section ----------------------------------------------------------------------------------------
    float* sec = out + MFMA_FLOATS;
    memset(sec, 0, sizeof(float) * F32SEC_FLOATS);
    Packer p32{fp32, sec, nullptr};
    // step `st` counted from quad q0: element (st & 3) of quad q0 + (st >> 2); kcol(h) -> column of W or -1
    auto step = [&](int q0, int st, int wOff, int ld, auto rowmap, auto kcol) {
        for (int l = 0; l < 64; ++l) {
            int i = l & 31, h = l >> 5, orow = rowmap(i), col = kcol(h);
            sec[(size_t)(q0 + (st >> 2)) * 256 + l * 4 + (st & 3)] = (orow >= 0 && col >= 0) ? fp32[wOff + orow * ld + col] : 0.f;
        }
    };
    // k-group g of a 16x16x4 layer, counted from quad q0: lane l carries value(row l & 15, k = l >> 4); the k order of the
    // swapped register pair (2g, 2g + 1) is [2g half 0, 2g+1 half 0, 2g half 1, 2g+1 half 1]
    auto group = [&](int q0, int g, auto value) {
        for (int l = 0; l < 64; ++l) {
            const int k = l >> 4, reg = 2 * g + (k & 1), h = k >> 1;
            sec[(size_t)(q0 + (g >> 2)) * 256 + l * 4 + (g & 3)] = value(l & 15, reg, h);
        }
    };
    auto table16 = [&](int off, auto value) {  // D layout of a 16x16x4 tile: register r of lane l = row 4 (l >> 4) + r
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) sec[off + l * 4 + r] = value(4 * (l >> 4) + r);
    };
    auto c19k = [](int r, int base) { return [=](int h) { int k = acc_row(r, h); return k < GDB_CFR ? base + k : -1; }; };
    for (int s = 0; s < 2; ++s) step(Q_VIEW, s, PW_VIEW_W, 4, lt(GDB_CFR), [=](int h) { return 2 * s + h; });
    for (int r = 0; r < 12; ++r) {
        // var registers; half 1 of registers 8..10 (no such channel) carries mean channels 16..18, moved there by the kernel
        step(Q_GVAR, r, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), [=](int h) {
            if (h == 1 && r >= 8 && r <= 10) return 2 * GDB_CFR + 16 + (r - 8);
            int k = acc_row(r, h); return k < GDB_CFR ? GDB_CFR + k : -1; });
        if (r < 8) step(Q_GMEAN, r, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19k(r, 2 * GDB_CFR));
        step(Q_GA, r, PW_GLOB_W, 3 * GDB_CFR, lt(GDB_GF), c19k(r, 0));
    }
    // fc: output row m of the 16x16x4 tile computes feature pi(m), chosen so that after the D swap lane (j, h) holds im
    // registers 0..7 (features acc_row(r, h)): rows 0..3 -> 0..3, 4..7 -> 8..11, 8..11 -> 4..7, 12..15 -> 12..15
    auto pi_fc = [](int m) { return (m & 3) + ((m >> 2) == 1 ? 8 : (m >> 2) == 2 ? 4 : (m >> 2) * 4); };
    for (int g = 0; g < 8; ++g)
        group(Q_FC, g, [&](int m, int reg, int h) { return fp32[PW_FC_W + pi_fc(m) * GDB_GF + acc_row(reg, h)]; });
    table16(T16_FC, [&](int m) { return fp32[PW_FC_B + pi_fc(m)]; });
    // [vox | im] operand: steps 0..7 = im registers (features acc_row(r, h) < 16), steps 8..11 = vox channel i + 4h
    auto hk = [](int st, int base) {
        return [=](int h) { return st < 8 ? base + GDB_CV + acc_row(st, h) : base + (st - 8) + 4 * h; };
    };
    for (int ot = 0; ot < 2; ++ot) {
        for (int st = 0; st < 12; ++st) {
            step(Q_LR0 + 3 * ot, st, PW_LR0_W, GDB_HD, tile(ot, GDB_HID), hk(st, 0));
            step(Q_W0B + 3 * ot, st, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID), hk(st, GDB_HID));
        }
        for (int st = 0; st < 32; ++st)
            step(Q_W0A + 8 * ot, st, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID), [=](int h) { return 32 * (st >> 4) + acc_row(st & 15, h); });
        // tail registers: feat (+) rgb in accumulator layout; half 1 of registers 8..11 (channels 20..23: none) carries dir 0..3
        for (int r = 0; r < 12; ++r)
            step(Q_W0C + 3 * ot, r, PW_W0_W, GDB_W0IN, tile(ot, GDB_HID), [=](int h) {
                if (h == 1 && r >= 8) return GDB_HID + GDB_HD + GDB_CFR + (r - 8);
                int k = acc_row(r, h); return k < GDB_CFR ? GDB_HID + GDB_HD + k : -1; });
    }
    // feat_head + sigma on x (register xi of X carries feature 32 (xi >> 4) + acc_row(xi & 15, h)).  Output rows: 0..3 feat_head
    // 0..3, 8..11 feat_head 4..7 (after the D swap lane (j, h) holds channels 4h..4h+3), row 4 = +sigma, row 5 = -sigma: both are
    // ReLU'd with the rest before the swap and sigma = ReLU(s) - ReLU(-s) (exact), so that the swap reads VALU results only.
    auto fh_w = [&](int m, int col) {
        if (m < 4) return fp32[PW_FH_W + m * GDB_HID + col];
        if (m >= 8 && m < 12) return fp32[PW_FH_W + (m - 4) * GDB_HID + col];
        if (m == 4) return fp32[PW_SIG_W + col];
        if (m == 5) return -fp32[PW_SIG_W + col];
        return 0.f;
    };
    for (int g = 0; g < 16; ++g)
        group(Q_FH, g, [&](int m, int reg, int h) { return fh_w(m, 32 * (reg >> 4) + acc_row(reg & 15, h)); });
    table16(T16_FH, [&](int m) {
        if (m < 4) return fp32[PW_FH_B + m];
        if (m >= 8 && m < 12) return fp32[PW_FH_B + m - 4];
        if (m == 4) return fp32[PW_SIG_B];
        if (m == 5) return -fp32[PW_SIG_B];
        return 0.f; });
    p32.table(T32_VIEW, PW_VIEW_B, 1, lt(GDB_CFR));
    p32.table(T32_GLOB, PW_GLOB_B, 1, lt(GDB_GF));
    for (int ot = 0; ot < 2; ++ot) {
        p32.table(T32_LR0 + 32 * ot, PW_LR0_B, 1, tile(ot, GDB_HID));
        p32.table(T32_W0 + 32 * ot, PW_W0_B, 1, tile(ot, GDB_HID));
        p32.table(T32_W2 + 32 * ot, PW_W2_W, 1, tile(ot, GDB_HID));
    }
    p32.table(T32_AGG, PW_AGG_W, 1, lt(GDB_GF));
    sec[S32_BAGG] = fp32[PW_AGG_B];
    sec[S32_BW2] = fp32[PW_W2_B];
}

// ---- device side ----------------------------------------------------------------------------
// Compiler-only fence: weight fragments are loop-invariant loads, and without it hipcc hoists all
// 33 of them (132 VGPRs) to kernel entry.
#define PHASE_FENCE() asm volatile("" ::: "memory")
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#ifdef GDB_DEBUG_STAMPS   // diagnostic build only: per-wave s_memtime stamps (tools/stamps.py)
// slot i: s_memtime (shader-clock ticks, a per-CU counter); stamps 0 and 9 also leave s_memrealtime (100 MHz, chip-wide) in slots
// 10 and 11: launch span and the clock the wave really ran at (MI355X guide, DVFS give-back item 6)
#define STAMP(i) do { if (dbg) { unsigned long long t_, r_ = 0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if ((i) == 0 || (i) == 9) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); \
    if (lane == 0) { unsigned long long* d_ = (unsigned long long*)dbg + (size_t)(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16; d_[(i)] = t_; \
        if ((i) == 0) d_[10] = r_; if ((i) == 9) d_[11] = r_; } } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

constexpr int NBLEND = 3 * 4 + GDB_CFR;      // 31 blended channels per view: 12 rgbs + 19 feat ⊕ rgb
constexpr int NOUT = NBLEND + GDB_CV;        // 39
// Per (wave, view) staging in LDS: rows of 32 bundles holding the reference's per-view vector [rgbs 12 | feat ⊕ rgb 19 | dir 4]
// (bundle_sampler.py:369) - 35 fp32 rows, one per channel, at GDB_PREC_F32 / F32X; 18 rows of packed halves at GDB_PREC_F16 (see row_feat<> below).
// Timing-only ablation bits (GDB_FUSED_SKIP; 1 colours, 2 features, 4 volume, 8 MLP) exist only in the diagnostic build
// -DGDB_DEBUG_SKIP: as runtime branches they split the gather into basic blocks and defeat its load scheduling.
#ifdef GDB_DEBUG_SKIP
#define SKIPPED(skip, bit) (((skip) & (bit)) != 0)
#else
#define SKIPPED(skip, bit) false
#endif
// At GDB_PREC_F16 the 4 direction values are only ever an f16 MFMA operand: they are staged as two rows of packed halves
// (dir0|dir1, dir2|dir3) - the same rounding, just earlier.  (The LDS allocation granule is 1280 B: tools/ubench/simd_map.hip.)
constexpr int COMP_LD = 33;                  // padded bundle stride of the composite record
// GDB_PREC_F16 goes further: the 12 sub-ray colours (values in [0, 1]) are staged as 6 rows of packed halves (round to nearest:
// 2.4e-4 at most, inside that path's 2e-3 bound), and so are the 19 channels of feat (+) rgb - sums of taps of the HALF-precision
// pyramid that are about to become f16 MFMA operands anyway - as 10 rows of channel pairs (2c, 2c + 1), channel 19 = 0:
// 18 rows = 2,304 B per view.  Five views (c5) are 11,520 B per wave: LDS would hold 14 waves on a CU, so the segment-wave kernel
// runs the 12 its registers allow (27 rows with fp32 features: 9 waves; 35 fp32 rows: 7).
// Rows of a staged view: [colours | feat (+) rgb | dir]
template <int PREC> constexpr int row_feat() { return PREC == GDB_PREC_F16 ? 6 : 12; }
template <int PREC> constexpr int feat_rows() { return PREC == GDB_PREC_F16 ? 10 : GDB_CFR; }
template <int PREC> constexpr int row_dir() { return row_feat<PREC>() + feat_rows<PREC>(); }
template <int PREC> constexpr int stage_v() { return (row_dir<PREC>() + (PREC != GDB_PREC_F16 ? 4 : 2)) * 32; }  // floats per (wave, view): 2304 / 4480 B
typedef _Float16 hpair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h2(float a, float b) { const hpair p = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, p); }
__device__ __forceinline__ hpair unpack_h2(unsigned u) { return __builtin_bit_cast(hpair, u); }
// Blended output channels of lane (j, h), register i of 16: the 6 colours of ITS OWN two sub-rays (channel c b^2 + 2h + e, the
// ones its half gathered), then 10 (half 0) / 9 (half 1) channels of feat (+) rgb.  -1: no channel (half 1's last register).
__device__ __forceinline__ constexpr int own_chan(int h, int i) {
    return i < 6 ? (i >> 1) * 4 + 2 * h + (i & 1) : (10 * h + i - 6 < GDB_CFR ? 12 + 10 * h + i - 6 : -1);
}
// Store the lane's 16 blended values (val(i)) into a channel-major record rec[channel * COMP_LD + col]: the half only moves two
// base addresses, every row offset is a compile-time constant.
template <class F>
__device__ __forceinline__ void store_own16(float* rec, int col, int h, F val) {
    float* ra = rec + h * (2 * COMP_LD) + col;    // colours: channel c b^2 + 2h + e
    float* rb = rec + h * (10 * COMP_LD) + col;   // feat (+) rgb: channel 12 + 10h + (i - 6)
#pragma unroll
    for (int i = 0; i < 6; ++i) ra[((i >> 1) * 4 + (i & 1)) * COMP_LD] = val(i);
#pragma unroll
    for (int i = 6; i < 15; ++i) rb[(12 + i - 6) * COMP_LD] = val(i);
    if (h == 0) rb[(12 + 9) * COMP_LD] = val(15);  // half 1's register 15 carries no channel
}
// ---- the output record of the one-wave schedules (segment wave, dense): BUNDLE-major, exactly as the rows lie in memory ----------
// rec[col * ld + channel], ld = floats per output row (NOUT, or NOUT + 2 = [feat | depth | opacity] for the packed layout), so that
// the window's output is one LINEAR copy LDS -> global, 16 bytes per lane and step with no index arithmetic.  (Round 3 kept the
// record channel-major and every output element paid a division by 39 / 41 and a transposed LDS address: ~12 VALU instructions
// per element, 20 elements per lane - 7 % of the wave's vector instructions, read off the ISA.)  ld is odd: lanes = columns
// write conflict-free.  Three-tensor form: depth and opacity sit behind the 32 rows, [32 ld + col] and [32 ld + 32 + col].
template <class F>
__device__ __forceinline__ void out_store_own16(float* rec, int ld, int col, int h, F val) {
    float* r = rec + col * ld;
    float* ra = r + 2 * h;    // colours: channel c b^2 + 2h + e
    float* rb = r + 10 * h;   // feat (+) rgb: channel 12 + 10h + (i - 6)
#pragma unroll
    for (int i = 0; i < 6; ++i) ra[(i >> 1) * 4 + (i & 1)] = val(i);
#pragma unroll
    for (int i = 6; i < 15; ++i) rb[12 + i - 6] = val(i);
    if (h == 0) rb[12 + 9] = val(15);  // half 1's register 15 carries no channel
}
__device__ __forceinline__ float* out_depth_slot(float* rec, int ld, int col) { return ld == NOUT ? rec + 32 * NOUT + col : rec + col * ld + NOUT; }
__device__ __forceinline__ float* out_opac_slot(float* rec, int ld, int col) { return ld == NOUT ? rec + 32 * NOUT + 32 + col : rec + col * ld + NOUT + 1; }
// This lane's 16 staged values of one view in that order (what the softmax blend of nerf.py:110 accumulates)
template <int PREC>
__device__ __forceinline__ void load_blend16(const float* __restrict__ st, int j, int h, float val[16]) {
    if constexpr (PREC == GDB_PREC_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned* su = (const unsigned*)st + h * 32;   // packed colour rows: row 2c + h = (sub-ray 2h, sub-ray 2h + 1) of colour c
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const h2 p = __builtin_bit_cast(h2, su[(2 * c) * 32 + j]);
            val[2 * c] = (float)p.x; val[2 * c + 1] = (float)p.y;
        }
    } else {
        const float* sa = st + h * (2 * 32);
#pragma unroll
        for (int i = 0; i < 6; ++i) val[i] = sa[((i >> 1) * 4 + (i & 1)) * 32 + j];
    }
    if constexpr (PREC == GDB_PREC_F16) {   // channels 10h .. 10h + 9 = packed rows 5h .. 5h + 4 (channel 19 is stored as 0)
        const unsigned* sp = (const unsigned*)st + (row_feat<PREC>() + 5 * h) * 32 + j;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const hpair p = unpack_h2(sp[r * 32]);
            val[6 + 2 * r] = (float)p.x; val[7 + 2 * r] = (float)p.y;
        }
        if (h) val[15] = 0.f;
        return;
    }
    const float* sb = st + h * (10 * 32);
#pragma unroll
    for (int i = 6; i < 15; ++i) val[i] = sb[(row_feat<PREC>() + i - 6) * 32 + j];
    const float last = sb[(row_feat<PREC>() + 9) * 32 + j];   // half 0: channel 9 of feat (+) rgb; half 1: past the 19 channels (a dir row)
    val[15] = h ? 0.f : last;
}
constexpr int COMP_CH = NOUT + 1;            // 39 channels + z
constexpr int COMP_ALPHA = (COMP_CH * COMP_LD + 3) / 4 * 4;  // alpha [32] of the slot
constexpr int COMP_WN = COMP_ALPHA + 32;                     // normalised weight [32]
constexpr int COMP_REC = COMP_WN + 32;                       // floats per slot record

struct FusedArgs {
    DevFrame f;
    const float* pw;  // packed weights (fp32 section, then MFMA sections)
    int row_begin, nrows, nseg, nsegs, ntiles, alias, skip;  // skip: timing-only ablation bits (diagnostic build)
    int ldo;          // floats per output row of bf: NOUT, or NOUT + 2 for the packed layout [feat | depth | opacity]
    int wave_floats;  // dense schedule: floats of LDS per wave (a workgroup may hold two waves, each with its own area)
    int row_lo, row_hi, tile_stride;  // dense schedule: global rows (batch item x H + row) of this launch; tiles a wave skips per step
    int flat_base;    // flat schedule: index of this launch's first window boundary in the side buffers (one launch per batch item)
    float* bf; float* depth; float* opac;  // depth / opac unused (NULL) in the packed layout
    float* wv;        // bundle_size 1 / 4 only: per (bundle, sample slot, view) the colour weight k_bundle_colours applies (else NULL)
    unsigned* dbg;    // diagnostic build only
    int xp_stagger, xp_per_rank;   // diagnostic build only (GDB_XP_STAGGER): start-of-launch stagger of the list kernels' waves by their rank on the CU
};
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte store to a 4-byte aligned address (gfx950: dword alignment suffices)
// nb rows of the record -> bf[b0 ...] (and depth / opac in the three-tensor form); the whole wave calls it
__device__ __forceinline__ void out_copy(const FusedArgs& a, const float* rec, size_t b0, int nb, int lane) {
    const int ld = a.ldo, n = nb * ld, nv = n >> 2;
    float* dst = a.bf + b0 * (size_t)ld;   // wave-uniform base; per-lane 32-bit offsets below
    for (int i = lane; i < nv; i += 64)
        *(f4u*)((char*)dst + 16u * (unsigned)i) = *(const f4u*)(rec + 4 * i);
    if (lane < (n & 3)) dst[4 * nv + lane] = rec[4 * nv + lane];
    if (ld == NOUT) {
        const int j = lane & 31;
        if (j < nb) {
            if (lane >> 5) a.opac[b0 + j] = rec[32 * NOUT + 32 + j];
            else a.depth[b0 + j] = rec[32 * NOUT + j];
        }
    }
}

// The product library keeps NO process-global mutable state: schedule and precision are arguments of
// the entry point.  Only the diagnostic build (-DGDB_DIAG: tools/stamps.py, tools/valu_split.sh) has a debug buffer and
// reads GDB_FUSED_SKIP from the environment.
#ifdef GDB_DIAG
static unsigned* g_dbg = nullptr;
extern "C" void gdb_debug_set_buffer(void* p) { g_dbg = (unsigned*)p; }
#endif

// k-step S (0/1) of a 32x32 accumulator tile as the next layer's B operand (optionally through ReLU).
template <int S, bool RELU>
__device__ __forceinline__ half8 acc_frag(const f32x16& a) {
    half8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (_Float16)a[8 * S + i];  // |v| > 65504 becomes inf, as any f16 conversion
    if (RELU) {  // on the packed halves: one v_pk_max_f16 per two values (ReLU commutes with the rounding)
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        r = __builtin_elementwise_max(r, z);
    }
    return r;
}

// Weight fragments / tables are addressed as (uniform base + constant) + per-lane offset.  The
// per-lane part goes through an opaque asm once per phase: otherwise hipcc materialises all ~45
// per-lane addresses at kernel entry (loop-invariant) and spills them.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
// Load through (wave-uniform base, 32-bit byte offset): the base stays in SGPRs and the offset is one VGPR
// (global_load ... v_off, s[base:base+1]) instead of a 64-bit per-lane address built for every load — in the
// per-view gather a fifth of all VALU instructions had been 64-bit address arithmetic.  The entry point checks
// that every addressed object is smaller than 4 GiB.
template <typename T>
__device__ __forceinline__ T ldu(const void* __restrict__ base, unsigned byte_off) {
    return *(const T*)((const char*)base + byte_off);
}
// The same with the uniform part pinned in an SGPR pair: without the pin the compiler folds a constant part of the
// base into the per-lane offset and is back to 64-bit VGPR address arithmetic (weight fragments: base + idx KiB + lane*16).
// The pointer is typed as global memory across the pin (an untyped one comes back as a flat pointer: flat_load, 64-bit).
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T ldu_pin(const void* __restrict__ base, unsigned byte_off) {
    const char GLOBAL_AS* b = (const char GLOBAL_AS*)base;
    asm volatile("" : "+s"(b));
    return *(const T GLOBAL_AS*)(b + byte_off);
}
__device__ __forceinline__ f32x16 load_tab(const float* __restrict__ mf, int off, int h) {
    return ldu_pin<f32x16>(mf + off, (unsigned)h * 64u);
}
__device__ __forceinline__ half8 load_frag(const float* __restrict__ mf, int idx, int lane) {
#if defined(GDB_DIAG) && defined(GDB_XP_NOW) && GDB_XP_NOW == 1   // diagnostic build only - timing experiment (wrong results): no weight loads at all
    half8 r; asm volatile("" : "=v"(r)); return r;
#else
    return ldu_pin<half8>(mf + (size_t)idx * 256, (unsigned)lane * 16u);
#endif
}
#define LANE_KEYS() const int lane_o = opaque(lane), h_o = lane_o >> 5; (void)h_o

// ---- split-f16 operands (GDB_PREC_F32X) --------------------------------------------------------------------------------
// An MFMA operand fragment is one half8 (X = false: GDB_PREC_F16) or a hi/lo pair of them (X = true) with hi + lo ≈ the fp32
// value to about 22 bits; a product is then lo·hi + hi·lo + hi·hi, three v_mfma_f32_32x32x16_f16 into the same fp32
// accumulator (the lo·lo term, 2^-22 of the product, is dropped).  Low parts of values below 2^-4 are f16 subnormals: the
// matrix pipe and v_cvt_pkrtz honour them (tools/ubench/mfma_denorm.hip).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float relu1(float x);
template <bool X> struct Frag;
template <> struct Frag<false> { half8 hi; };
template <> struct Frag<true> { half8 hi, lo; };
template <bool X>
__device__ __forceinline__ Frag<X> load_fragx(const float* __restrict__ mf, int idx, int lane) {
    Frag<X> r;
    r.hi = load_frag(mf, idx, lane);
    if constexpr (X) r.lo = load_frag(mf + XLO_OFF, idx, lane);
    return r;
}
template <bool X>
__device__ __forceinline__ f32x16 mm(const Frag<X>& a, const Frag<X>& b, f32x16 c) {
    if constexpr (X) {  // small terms first
        c = MFMA(a.lo, b.hi, c);
        c = MFMA(a.hi, b.lo, c);
    }
    return MFMA(a.hi, b.hi, c);
}
// hi = f16(v) (towards zero), lo = f16(v - hi): two v_cvt_pkrtz and two subtractions per pair of values
__device__ __forceinline__ Frag<true> split8(const float v[8]) {
    Frag<true> r;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const half2v hp = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]));
        const half2v lp = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[2 * p] - (float)hp.x, v[2 * p + 1] - (float)hp.y));
        r.hi[2 * p] = hp.x; r.hi[2 * p + 1] = hp.y;
        r.lo[2 * p] = lp.x; r.lo[2 * p + 1] = lp.y;
    }
    return r;
}
// k-step S of an accumulator tile as the next layer's B operand, either form
template <int S, bool RELU, bool X>
__device__ __forceinline__ Frag<X> accf(const f32x16& a) {
    if constexpr (X) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = RELU ? relu1(a[8 * S + i]) : a[8 * S + i];
        return split8(v);
    } else {
        return Frag<false>{acc_frag<S, RELU>(a)};
    }
}
// ReLU as exactly one instruction: written as fmaxf(x, 0) hipcc first canonicalises an MFMA output with v_max x, x, x
// (IEEE-mode maxnum wants quieted inputs) — two instructions per ReLU in every per-view pass.  A signed integer max on the
// bit pattern is the same function (negative floats, -0 included, have the sign bit set) and needs no canonical input.
// (Not inline asm: the compiler's MFMA -> VALU hazard handling does not look inside it, and a v_max_f32 written that way
// read accumulators before the MFMA had landed.)
__device__ __forceinline__ float relu1(float x) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}
__device__ __forceinline__ float dot16_relu(const f32x16& a, const f32x16& w) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = fmaf(relu1(a[i]), w[i], s);
    return s;
}

// ---- fast-path arithmetic (this translation unit is built with -ffp-contract=fast) ------------
// The fused kernel trades the last ulp of the geometry for VALU issue slots: reciprocal-based
// division, v_rsq / v_sqrt / v_log, fused multiply-adds.  The one discrete decision of the path,
// the per-bundle sample count (bundle_sampler.py:179), keeps its IEEE division (sample_count()).
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// F.normalize(p=2, eps=1e-12): x / max(|x|, 1e-12)
__device__ __forceinline__ void fnormalize3(const float a[3], float o[3]) {
    float r = frsq(fmaxf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2], 1e-24f));
    o[0] = a[0] * r; o[1] = a[1] * r; o[2] = a[2] * r;
}

// ---- every fusion of the gather's coordinate arithmetic is SPELLED OUT ------------------------------------------------------------------
// This translation unit is built with -ffp-contract=fast: where the source leaves a*b + c to the compiler, whether it becomes one fma or
// two roundings is decided per kernel instantiation, by the code around it.  For the coordinates that is not harmless at GDB_PREC_F16:
// an ulp of difference in a mip level moves a tap weight by 1e-7, and where the tap sum sits next to a rounding boundary its staged
// HALF flips - one f16 ulp (1e-4) of one feature of one sample, seen as 2.7e-5 between two schedules of the same frame (round 5, when
// the dense and flat kernels became one body).  So the functions between a sample's position and its tap weights run with contraction
// off and say fmaf where a fused multiply-add is meant: every kernel then computes the same bits.
__device__ __forceinline__ void tex_coord_f(float u, int size, int& i0, int& i1, float& f) {
#pragma clang fp contract(off)
    float x = fminf(fmaxf(fmaf(u, (float)size, -0.5f), 0.f), (float)(size - 1));
    float xf = floorf(x);
    i0 = (int)xf;
    i1 = min(i0 + 1, size - 1);
    f = x - xf;
}
// Bilinear taps of one mip level of the chunk-planar pyramid ([chunk][y][x] of 16-byte chunks): byte offsets of
// the four taps inside chunk plane 0 (level offset lvlB included), the plane size in bytes, and weights;
// clamp-to-edge.  lw scales the level's weights.
struct Taps { unsigned o00, o10, o01, o11, planeB; float w00, w10, w01, w11; };  // byte offsets inside chunk plane 0; bytes per chunk plane
__device__ __forceinline__ Taps make_taps(float u, float v, int W, int H, unsigned lvlB, float lw) {
#pragma clang fp contract(off)
    int x0, x1, y0, y1; float fx, fy;
    tex_coord_f(u, W, x0, x1, fx);
    tex_coord_f(v, H, y0, y1, fy);
    Taps t;
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): rows and widths are far below 2^24
    unsigned r0 = __umul24(y0, W), r1 = __umul24(y1, W);
    t.o00 = lvlB + 16u * (r0 + x0); t.o10 = lvlB + 16u * (r0 + x1);
    t.o01 = lvlB + 16u * (r1 + x0); t.o11 = lvlB + 16u * (r1 + x1);
    t.planeB = 16u * __umul24(W, H);
    float ex = (1.f - fx) * lw, wx = fx * lw;
    t.w00 = ex * (1.f - fy); t.w10 = wx * (1.f - fy); t.w01 = ex * fy; t.w11 = wx * fy;
    return t;
}
__device__ __forceinline__ void tap_acc(float4& o, const float4 a, float w) {
#if defined(GDB_DIAG) && defined(GDB_XP_PK)  // diagnostic build only.  Reproducer kept in-tree (tools/ab_flags.sh "-DGDB_XP_PK=1" "-DGDB_XP_PK=2"; profiles/r02/ab_packed_f32_scratch_vs_registers.txt).
    // The "packed f32 corrupts lanes 48..63" finding of round 1 is NOT a packed-math hazard: with the tap accumulate written as
    // two v_pk_fma_f32, hipcc stops promoting the Taps / RgbTaps weight structs to registers and forms the op_sel operand pairs
    // (w00,w10) (w10,w01) (w01,w11) with overlapping, 4-byte-aligned scratch_load_dwordx2 of the struct in PRIVATE MEMORY, right
    // behind a lane-masked scratch_store_dwordx3/x4 (.private_segment_fixed_size 40, no spill).  That private-memory round trip
    // returns stale lanes at full occupancy (variant 1 fails test_fused_matches_unfused_at_full_size with errors 0.25-0.6);
    // variant 2 has the same 30 packed FMAs at the same place with the pairs built in registers (scratch size 0) and passes
    // every parity and determinism test.  tests/test_build_resources.py keeps every product kernel's scratch data-free.
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v lo = {o.x, o.y}, hi = {o.z, o.w};
#if GDB_XP_PK == 2  // the weight made opaque: the {w, w} pair must be built in registers (in variant 1 hipcc keeps
    // the Taps struct in PRIVATE MEMORY and forms the op_sel pairs with overlapping 8-byte scratch loads of it)
    asm volatile("" : "+v"(w));
#endif
    const f2v alo = {a.x, a.y}, ahi = {a.z, a.w}, ww = {w, w};
    lo = __builtin_elementwise_fma(alo, ww, lo); hi = __builtin_elementwise_fma(ahi, ww, hi);
    o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
#else
    o.x = fmaf(a.x, w, o.x); o.y = fmaf(a.y, w, o.y); o.z = fmaf(a.z, w, o.z); o.w = fmaf(a.w, w, o.w);
#endif
}
// The fetch is split into "issue the loads" and "accumulate" so a view's loads can all be in flight before the
// first one is consumed (one memory round trip instead of one per block).
// Chunk ownership: half h takes 16-byte chunks h and 2+h (channels 4h..4h+3, 8+4h..8+4h+3) and the 8-byte half h of
// chunk 4 (channels 16+2h, 17+2h; channel 19 is padding) — the same unconditional loads in both halves, no branch.
struct TapData { float4 t[2][4]; float2 u[4]; };
__device__ __forceinline__ void taps_load(const void* __restrict__ pyr, const Taps& t, int h, TapData& d) {
    const unsigned c0 = h ? t.planeB : 0u, c1 = c0 + 2u * t.planeB, c4 = 4u * t.planeB + 8u * (unsigned)h;
    d.t[0][0] = ldu<float4>(pyr, t.o00 + c0); d.t[0][1] = ldu<float4>(pyr, t.o10 + c0);
    d.t[0][2] = ldu<float4>(pyr, t.o01 + c0); d.t[0][3] = ldu<float4>(pyr, t.o11 + c0);
    d.t[1][0] = ldu<float4>(pyr, t.o00 + c1); d.t[1][1] = ldu<float4>(pyr, t.o10 + c1);
    d.t[1][2] = ldu<float4>(pyr, t.o01 + c1); d.t[1][3] = ldu<float4>(pyr, t.o11 + c1);
    d.u[0] = ldu<float2>(pyr, t.o00 + c4); d.u[1] = ldu<float2>(pyr, t.o10 + c4);
    d.u[2] = ldu<float2>(pyr, t.o01 + c4); d.u[3] = ldu<float2>(pyr, t.o11 + c4);
}
// INIT: the accumulators are (re)started by the first tap (acc = a*w, bit-identical to fma(a, w, 0)).
template <bool INIT>
__device__ __forceinline__ void taps_acc(const Taps& t, const TapData& d, float4 acc[3]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (INIT) acc[s] = make_float4(d.t[s][0].x * t.w00, d.t[s][0].y * t.w00, d.t[s][0].z * t.w00, d.t[s][0].w * t.w00);
        else tap_acc(acc[s], d.t[s][0], t.w00);
        tap_acc(acc[s], d.t[s][1], t.w10);
        tap_acc(acc[s], d.t[s][2], t.w01);
        tap_acc(acc[s], d.t[s][3], t.w11);
    }
    const float ax = INIT ? d.u[0].x * t.w00 : fmaf(d.u[0].x, t.w00, acc[2].x), ay = INIT ? d.u[0].y * t.w00 : fmaf(d.u[0].y, t.w00, acc[2].y);
    acc[2].x = fmaf(d.u[3].x, t.w11, fmaf(d.u[2].x, t.w01, fmaf(d.u[1].x, t.w10, ax)));
    acc[2].y = fmaf(d.u[3].y, t.w11, fmaf(d.u[2].y, t.w01, fmaf(d.u[1].y, t.w10, ay)));
}

// ---- half-precision pyramid (GDB_PREC_F16; gdb_internal.h PYR16_*): the taps are gather_view16's (below) ---------------------------------
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

// Two x-adjacent floats in one 8-byte load (4-byte aligned; gfx950 global loads need dword alignment only).
typedef float F2v __attribute__((ext_vector_type(2)));
typedef F2v F2u __attribute__((aligned(4)));  // a vector type, so it can be loaded through any address space

// Bilinear RGB, grid_sample border / align_corners=False, from planar (3,Ho,Wo); 32-bit offsets.  The x pair
// (x0, x0+1) is one 8-byte load: at the right edge the pair is shifted left by one and the weight moved onto
// its second element, which keeps the pair inside the row (needs Wo >= 2).
struct RgbTaps { unsigned o0, o1; float w00, w10, w01, w11; };
struct RgbData { F2u a[3], b[3]; };
__device__ __forceinline__ RgbTaps rgb_taps(int Ho, int Wo, float px, float py) {
#pragma clang fp contract(off)
    // px, py are pixel coordinates: grid g = 2*px/Wo - 1 -> ((g+1)*Wo - 1)/2 = px - 0.5
    float x = fminf(fmaxf(px - 0.5f, 0.f), (float)(Wo - 1)), y = fminf(fmaxf(py - 0.5f, 0.f), (float)(Ho - 1));
    float yf = floorf(y);
    int x0 = min((int)floorf(x), Wo - 2), y0 = (int)yf;
    float wx = x - (float)x0, wy = y - yf;
    int y1 = min(y0 + 1, Ho - 1);  // a clamped row carries weight 0 (wy = 0 at the edge)
    RgbTaps t;
    t.w00 = (1.f - wx) * (1.f - wy); t.w10 = wx * (1.f - wy); t.w01 = (1.f - wx) * wy; t.w11 = wx * wy;
    t.o0 = 4u * (__umul24(y0, Wo) + x0); t.o1 = 4u * (__umul24(y1, Wo) + x0);  // byte offsets inside a colour plane
    return t;
}
__device__ __forceinline__ void rgb_load(const float* __restrict__ img, unsigned plane, const RgbTaps& t, RgbData& d) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { d.a[c] = ldu<F2u>(img + c * plane, t.o0); d.b[c] = ldu<F2u>(img + c * plane, t.o1); }
}
__device__ __forceinline__ void rgb_combine(const RgbTaps& t, const RgbData& d, float rgb[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = fmaf(d.b[c].y, t.w11, fmaf(d.b[c].x, t.w01, fmaf(d.a[c].y, t.w10, d.a[c].x * t.w00)));
}

extern __shared__ float4 smem4[];
// Wave priority by progress (s_setprio; the SIMD's arbiter otherwise favours the oldest wave, which then finishes first and leaves
// its mates to end the launch alone): a wave with tiles (slots) still to come outranks a wave on its last one, and within a tile the
// gather phase - address arithmetic and load issue, whose latency everything else waits for - outranks the MLP / composite phases.
// Measured (profiles/r04/ab_wave_priority.txt, same box): k_render_dense c2 f32 -0.8 %, c2 f16 -3.0 %, c4 f32 -2.3 %, c4 f16 -4.1 %;
// k_render_flat +1.9 % (two equal tiles per wave: left alone); the inverse senses lose everywhere.
template <bool ON>
__device__ __forceinline__ void wave_prio(bool last_tile, bool gather) {
    if constexpr (ON) {
        if (!last_tile) { if (gather) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
        else            { if (gather) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    }
}
#if !defined(GDB_DIAG) || !defined(GDB_XP_PRIO_SOLO)   // (an experiment of the diagnostic build: profiles/r04/ab_wave_priority.txt)
#undef GDB_XP_PRIO_SOLO
#define GDB_XP_PRIO_SOLO 0
#endif


// This lane's 12 staged feature values (channels 8s+4h+e, the accumulator rows it owns) and, for half
// 0, the 4 direction values; from them the two f16 operand fragments of the per-view tail vector
// tv[32] = [feat ⊕ rgb 19 | 0 | dir 4 at 24..27 | 0].
template <bool X> struct Tail { float fv[12]; Frag<X> T0, T1; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 h8_of(unsigned a, unsigned b, unsigned c, unsigned d) { const u32x4 v = {a, b, c, d}; return __builtin_bit_cast(half8, v); }
// GDB_PREC_F16: the staged halves ARE the operand elements.  u[0..3] = this lane's channel pairs of k-step 0 (accumulator registers
// 0..7 = channels 8s + 4h + e), u[4], u[5] = channels (16, 17), (18, 19 = 0).  Both lane halves read the SAME rows for the k-step-1
// operand (channels 16..19, the direction code): half 1's k slots there (k = 20..23, 28..31) meet zero weights in every fragment, so
// what it carries is irrelevant as long as it is finite - and a copy of half 0's values is exactly as finite as those.  (Until round 5
// half 1 read zeros under an exec mask: four s_and_saveexec / s_or pairs per call, three calls per view.)
struct TailP { unsigned u[6]; half8 T0, T1; };
__device__ __forceinline__ TailP load_tailp(const float* __restrict__ st, int j, int h) {
    constexpr int RF = row_feat<GDB_PREC_F16>(), RD = row_dir<GDB_PREC_F16>();
    const unsigned* su = (const unsigned*)st + j;
    TailP t;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) t.u[2 * s + r] = su[(RF + 4 * s + 2 * h + r) * 32];   // channels 8s + 4h + 2r, + 1
    t.u[4] = su[(RF + 8) * 32]; t.u[5] = su[(RF + 9) * 32];
    const unsigned d01 = su[(RD + 0) * 32], d23 = su[(RD + 1) * 32];
    t.T0 = h8_of(t.u[0], t.u[1], t.u[2], t.u[3]);
    // tv[19] is padding: the constant one that carries view_fc's bias (half 1's twin slot, k = 23, has zero weights)
    t.T1 = h8_of(t.u[4], (t.u[5] & 0xFFFFu) | 0x3C000000u, d01, d23);   // dir sits at tv[24..27]
    return t;
}
template <bool X>
__device__ __forceinline__ Tail<X> load_tail(const float* __restrict__ st, int j, int h) {
    constexpr int RF = row_feat<X ? GDB_PREC_F32X : GDB_PREC_F16>(), RD = row_dir<X ? GDB_PREC_F32X : GDB_PREC_F16>();
    Tail<X> t;
    if constexpr (!X) {   // GDB_PREC_F16: packed channel pairs
        const TailP p = load_tailp(st, j, h);
#pragma unroll
        for (int i = 0; i < 6; ++i) { const hpair q = unpack_h2(p.u[i]); t.fv[2 * i] = (float)q.x; t.fv[2 * i + 1] = (float)q.y; }
        t.T0.hi = p.T0; t.T1.hi = p.T1;
        return t;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int ch = 8 * s + 4 * h + e;
            t.fv[4 * s + e] = ch < GDB_CFR ? st[(RF + ch) * 32 + j] : 0.f;
        }
    if constexpr (X) {  // the direction code is staged as four fp32 rows (as for GDB_PREC_F32)
        float v1[8];
#pragma unroll
        for (int i = 0; i < 3; ++i) v1[i] = t.fv[8 + i];  // channels 16..18 (half 0) / zeros (half 1)
        v1[3] = h == 0 ? 1.f : 0.f;                        // tv[19] is padding: constant one that carries view_fc's bias
#pragma unroll
        for (int i = 0; i < 4; ++i) v1[4 + i] = h == 0 ? st[(RD + i) * 32 + j] : 0.f;  // dir sits at tv[24..27], owned by half 0
        t.T0 = split8(t.fv);
        t.T1 = split8(v1);
    }
    return t;
}

// g_v = feat ⊕ rgb + ReLU(view_fc(dir)) of one staged view, in accumulator layout   nerf.py:69-71
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
template <bool X>
__device__ __forceinline__ f32x16 view_g(const Tail<X>& t, const Frag<X>& a_view) {
    f32x16 g = mm<X>(a_view, t.T1, zero16());
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = t.fv[i] + relu1(g[i]);
#pragma unroll
    for (int i = 12; i < 16; ++i) g[i] = 0.f;
    return g;
}

// View-direction code   bundle_sampler.py:362-367.  Contraction is off here: with a source camera at the target
// pose td == sd, and the reference's td - sd is exactly 0 (so the code is 0,0,0,1); a fused
// fma(dd_t, r_t, -dd_s * r_s) would leave the product's rounding error instead, which the normalisation
// then blows up to a unit vector of noise.
// (td, the unit direction from the target camera to the sample, is the same for every view: target_dir(), once per sample)
__device__ __forceinline__ void target_dir(const float ctr[3], const float* __restrict__ to, float td[3]) {
#pragma clang fp contract(off)
    float dd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) dd[r] = ctr[r] - to[r];
    fnormalize3(dd, td);
}
__device__ __forceinline__ void view_dir_code(const float ctr[3], const float td[3], const float* __restrict__ so, float dir[4]) {
#pragma clang fp contract(off)
    float sd[3], dd[3], dif[3], dnn[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) dd[r] = ctr[r] - so[r];
    fnormalize3(dd, sd);
#pragma unroll
    for (int r = 0; r < 3; ++r) dif[r] = td[r] - sd[r];
    fnormalize3(dif, dnn);
    dir[0] = dnn[0]; dir[1] = dnn[1]; dir[2] = dnn[2]; dir[3] = td[0] * sd[0] + td[1] * sd[1] + td[2] * sd[2];
}

// Gather of one (sample slot, view) for this lane: two sub-ray colours, this half's feature
// chunks at the footprint's mip level, the view-direction code.  bundle_sampler.py:327-369
// P16: the feature taps come from the half-precision pyramid (GDB_PREC_F16).
// a_i <- half 0's a_i in both halves, b_i <- half 1's a_i in both halves (b_i = a_i on entry): v_permlane32_swap_b32 a, b trades the upper
// 32 lanes of a for the lower 32 lanes of b.  Same caveats as swap32_3: every operand is the result of a plain VALU instruction, the
// s_nop pairs are the hazard padding; scalars in and out of the asm statement (see xchg9).
__device__ __forceinline__ void xchg8(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, float& a4, float& a5, float& a6, float& a7,
                                      unsigned& b0, unsigned& b1, unsigned& b2, unsigned& b3, float& b4, float& b5, float& b6, float& b7) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                 "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                 "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                   "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
}
// Offset of mip level l (a per-lane value) among the wave-uniform level offsets, as a SELECT-free sum of masked terms: written as the
// ternary chain  l == 0 ? 0 : (l == 1 ? lo1 : ...)  hipcc turned each of the chains of a view into divergent branches (s_and_saveexec /
// s_cbranch_execnz regions around one v_cndmask each - four of them per view in the f16 gather, read off the ISA in round 6).
__device__ __forceinline__ unsigned level_off(int l, unsigned lo1, unsigned lo2, unsigned lo3) {
    const unsigned m1 = 0u - (unsigned)(l == 1), m2 = 0u - (unsigned)(l == 2), m3 = 0u - (unsigned)(l >= 3);
    return (lo1 & m1) | (lo2 & m2) | (lo3 & m3);
}
// GDB_PREC_F32 / F32X: the feature taps come from the fp32 pyramid, the colours from the planar fp32 source images.
// The bilinear taps of the two mip levels are computed ONCE per sample (round 5): lane half 0 forms level l0's byte offsets and weights,
// half 1 level l1's (the same instructions on per-lane level data), and v_permlane32_swap hands each half the other's - 16 + 2
// instructions against make_taps' ~86 a second time.  On the fp32 datapath a vector instruction is matrix time.
// RGB = false (bundle_size 1 / 4, round 6): the sub-ray colours are not gathered here - a lane half owns two sub-rays of a 2 x 2 bundle,
// not b^2 / 2 of any bundle - but by k_bundle_colours afterwards, from the blend weights this kernel leaves per (sample, view).
template <bool RGB = true>
__device__ __forceinline__ void gather_view(const DevFrame& f, int bi, int v, int h, const float xyz[2][3], const float ctr[3],
                                            float ball, const float td[3], float4 feat[3], float dir[4], float rgb[2][3], int skip) {
#pragma clang fp contract(off)   // (every fused multiply-add below is written fmaf: see tex_coord_f)
    float sc[SRC_STRIDE];  // (bi, v) are wave-uniform: scalar loads (of the entries used below), the block lives in SGPRs
    {
        const kfloat* scg = kptr(src_cam(f, bi, v));
        asm volatile("" : "+s"(scg));  // not loop-invariant for the compiler: inside a slot loop LICM would hoist all V blocks
#pragma unroll
        for (int i = 0; i < SRC_STRIDE; ++i) sc[i] = scg[i];
    }
    const float* img = f.src_images + ((size_t)bi * f.V + v) * 3 * f.Ho * f.Wo;
    const bool do_rgb = RGB && !SKIPPED(skip, 1), do_tex = !SKIPPED(skip, 2);
    // ---- addresses and weights --------------------------------------------------------------------
    // sphere centre in the camera frame: the mean of the sub-ray points maps to the mean of their images   :340
    float cc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        cc[r] = fmaf(sc[S_E + 4 * r], ctr[0], fmaf(sc[S_E + 4 * r + 1], ctr[1], fmaf(sc[S_E + 4 * r + 2], ctr[2], sc[S_E + 4 * r + 3])));
    // footprint -> mip level   :343-348
    float d2 = fmaf(cc[2], cc[2], fmaf(cc[1], cc[1], cc[0] * cc[0]));
    float icz = frcp(cc[2]), ib = frcp(ball);
    float sec2 = d2 * icz * icz;
    float aa = fsqrt(fmaxf(fmaf(d2 * ib, ib, -1.f), 1e-12f)), cq = fsqrt(fmaxf(sec2 - 1.f, 1e-12f));
    float level = __builtin_amdgcn_logf(sec2 * frcp(aa + cq) * sc[S_IPIXR]);  // v_log_f32 = log2
    float ci[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) ci[r] = fmaf(sc[S_KS + 3 * r], cc[0], fmaf(sc[S_KS + 3 * r + 1], cc[1], sc[S_KS + 3 * r + 2] * cc[2]));
    float izc = frcp(fmaxf(ci[2], 1e-6f));
    float tu = ci[0] * izc * f.invW, tvv = ci[1] * izc * f.invH;   // :351-353
    // linear-mipmap-linear fetch as one 8-tap weighted sum   :355-359
    int l0, l1; float frac;
    mip_select(level, f.levels, l0, l1, frac);
    const void* pyr = (const void*)(f.pyr + ((size_t)bi * f.V + v) * f.pyrStride);
    // level offsets as register values: left as f.lvlOff[...] selects, the compiler selects the *address* and
    // issues a per-lane load from kernarg memory (a full vector-memory round trip for a constant)
    unsigned lo1 = f.lvlOff[1], lo2 = f.lvlOff[2], lo3 = f.lvlOff[3];
    asm volatile("" : "+s"(lo1), "+s"(lo2), "+s"(lo3));
    // this half's level -> byte offsets (level offset included) and weights; then each half gets the other's
    Taps t0, t1;
    {
        const int lm = h ? l1 : l0;
        const unsigned om = lm == 0 ? 0u : (lm == 1 ? lo1 : (lm == 2 ? lo2 : lo3));   // (ONE chain per view here; as level_off() the split-f16 flat build spills four scalar registers)
        const Taps m = make_taps(tu, tvv, f.W >> lm, f.H >> lm, om << 2, h ? frac : 1.f - frac);
        unsigned a0 = m.o00, a1 = m.o10, a2 = m.o01, a3 = m.o11, b0 = a0, b1 = a1, b2 = a2, b3 = a3;
        float a4 = m.w00, a5 = m.w10, a6 = m.w01, a7 = m.w11, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
        xchg8(a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7);   // a: half 0's values (level l0), b: half 1's (level l1)
        t0.o00 = a0; t0.o10 = a1; t0.o01 = a2; t0.o11 = a3; t0.w00 = a4; t0.w10 = a5; t0.w01 = a6; t0.w11 = a7;
        t1.o00 = b0; t1.o10 = b1; t1.o01 = b2; t1.o11 = b3; t1.w00 = b4; t1.w10 = b5; t1.w01 = b6; t1.w11 = b7;
        t0.planeB = 16u * __umul24(f.W >> l0, f.H >> l0); t1.planeB = 16u * __umul24(f.W >> l1, f.H >> l1);
    }
    const bool two = frac > 0.f && do_tex;
    RgbTaps rt[2];
    if constexpr (RGB) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {  // this half's two sub-rays   :327-337
        float im[3];  // K (E x + t) as one pre-multiplied 3x4 (S_P): 9 FMAs per point instead of 18
#pragma unroll
        for (int r = 0; r < 3; ++r)
            im[r] = fmaf(sc[S_P + 4 * r], xyz[e][0], fmaf(sc[S_P + 4 * r + 1], xyz[e][1], fmaf(sc[S_P + 4 * r + 2], xyz[e][2], sc[S_P + 4 * r + 3])));
        float iz = frcp(fmaxf(im[2], 1e-6f));
        rt[e] = rgb_taps(f.Ho, f.Wo, im[0] * iz, im[1] * iz);
    }
    }
    // ---- issue: level 0 and both sub-rays' colours in flight together -------------------------------
    TapData d0, d1;
    RgbData rd[2];
    if (do_tex) taps_load(pyr, t0, h, d0);
    const unsigned plane = (unsigned)(f.Ho * f.Wo);
    if (do_rgb) { rgb_load(img, plane, rt[0], rd[0]); rgb_load(img, plane, rt[1], rd[1]); }
    // ---- consume level 0, issue level 1, consume the colours under its latency, consume level 1 -------
    if (do_tex) taps_acc<true>(t0, d0, feat);
    else feat[0] = feat[1] = feat[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (two) taps_load(pyr, t1, h, d1);
    if (do_rgb) { rgb_combine(rt[0], rd[0], rgb[0]); rgb_combine(rt[1], rd[1], rgb[1]); }
    else { for (int e = 0; e < 2; ++e) for (int c = 0; c < 3; ++c) rgb[e][c] = 0.f; }
    if (two) taps_acc<false>(t1, d1, feat);
    view_dir_code(ctr, td, sc + S_C, dir);
}

// ---- the GDB_PREC_F16 gather (round 5) -------------------------------------------------------------------------------------------------
// Same arithmetic as gather_view<true> on the same half-precision pyramid, re-cut for what bounds the f16 kernels (profiles/r05/
// pmc_c5_f16_baseline.txt: texture addresser 86 % busy, vector ALU 84 %): fewer load instructions and fewer address instructions.
//  * colours from the half-precision RGBA copy of the source images (gdb_internal.h IMG16_*): an x pair of a tap row is ONE 16-byte
//    load, 4 per view and lane instead of 12 eight-byte ones;
//  * plane 2 of the pyramid (channels 16..19) as x pairs: one 16-byte load per tap row, 2 per level instead of 4 four-byte ones;
//  * the bilinear taps of the two mip levels are computed ONCE per sample: lane half 0 forms level l0's texel indices and weights,
//    half 1 level l1's, and v_permlane32_swap hands each half the other's (18 instructions against make_taps16's ~86 a second time).
// 28 -> 16 load instructions and ~60 fewer vector instructions per (sample, view).
struct Taps16s { unsigned p00, p10, p01, p11; int edge; float w00, w10, w01, w11; };   // main-plane texel indices; x0 is the row's last texel
__device__ __forceinline__ Taps16s make_taps16s(float u, float v, int W, int H, float lw) {
#pragma clang fp contract(off)
    int x0, x1, y0, y1; float fx, fy;
    tex_coord_f(u, W, x0, x1, fx);
    tex_coord_f(v, H, y0, y1, fy);
    Taps16s t;
    const unsigned r0 = __umul24(y0, W), r1 = __umul24(y1, W);
    t.p00 = r0 + x0; t.p10 = r0 + x1; t.p01 = r1 + x0; t.p11 = r1 + x1;
    t.edge = (x0 > W - 2 && W >= 2) ? 1 : 0;   // the pair (x0, x0 + 1) would leave the row: taken one texel to the left, weight on its second element
    const float ex = (1.f - fx) * lw, wx = fx * lw;
    t.w00 = ex * (1.f - fy); t.w10 = wx * (1.f - fy); t.w01 = ex * fy; t.w11 = wx * fy;
    return t;
}
// a <- half 0's a in both halves, b <- half 1's a in both halves (v_permlane32_swap_b32 a, b: the upper 32 lanes of a trade places with the
// lower 32 lanes of b; same caveats as swap32_3: operands are results of plain VALU instructions, the s_nop pairs are the hazard padding)
// (scalars in and out of the asm statement, never members of a struct by reference: a struct whose members are asm operands was
// parked in PRIVATE MEMORY by hipcc in one instantiation - the kind of round trip build.py refuses in these kernels)
__device__ __forceinline__ void xchg9(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, int& a4, float& a5, float& a6, float& a7, float& a8,
                                      unsigned& b0, unsigned& b1, unsigned& b2, unsigned& b3, int& b4, float& b5, float& b6, float& b7, float& b8) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %9\n\tv_permlane32_swap_b32 %1, %10\n\tv_permlane32_swap_b32 %2, %11\n\t"
                 "v_permlane32_swap_b32 %3, %12\n\tv_permlane32_swap_b32 %4, %13\n\tv_permlane32_swap_b32 %5, %14\n\t"
                 "v_permlane32_swap_b32 %6, %15\n\tv_permlane32_swap_b32 %7, %16\n\tv_permlane32_swap_b32 %8, %17\n\ts_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8),
                   "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7), "+v"(b8));
}
struct TapData16s { half8v q[4]; half8v u[2]; };   // four main-plane texels; plane-2 x pairs of the two tap rows
__device__ __forceinline__ void taps_load16s(const void* __restrict__ pyr16, const Taps16s& t, unsigned lvlB, unsigned hw, int h, TapData16s& d) {
    const unsigned b0 = lvlB + (h ? 16u * hw : 0u), b2 = lvlB + PYR16_PLANE2(hw);
    d.q[0] = ldu<half8v>(pyr16, b0 + 16u * t.p00); d.q[1] = ldu<half8v>(pyr16, b0 + 16u * t.p10);
    d.q[2] = ldu<half8v>(pyr16, b0 + 16u * t.p01); d.q[3] = ldu<half8v>(pyr16, b0 + 16u * t.p11);
    typedef half8v half8u __attribute__((aligned(8)));   // (8-byte aligned 16-byte loads: dword alignment suffices on gfx950)
    d.u[0] = ldu<half8u>(pyr16, b2 + 8u * (t.p00 - (unsigned)t.edge)); d.u[1] = ldu<half8u>(pyr16, b2 + 8u * (t.p01 - (unsigned)t.edge));
}
template <bool INIT>
__device__ __forceinline__ void taps_acc16s(const Taps16s& t, const TapData16s& d, int h, float4 acc[3]) {
    const float w[4] = {t.w00, t.w10, t.w01, t.w11};
    float* a0 = (float*)&acc[0]; float* a1 = (float*)&acc[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0[e] = (INIT && k == 0) ? (float)d.q[k][e] * w[k] : fmaf((float)d.q[k][e], w[k], a0[e]);
            a1[e] = (INIT && k == 0) ? (float)d.q[k][4 + e] * w[k] : fmaf((float)d.q[k][4 + e], w[k], a1[e]);
        }
    }
    // plane 2: pair element 0 = texel x0 - edge, element 1 = the next; at the row's end the tap x0 IS element 1 (and x1 = x0 carries no weight)
    const float pa = t.edge ? 0.f : t.w00, pb = t.edge ? t.w00 + t.w10 : t.w10, pc = t.edge ? 0.f : t.w01, pd = t.edge ? t.w01 + t.w11 : t.w11;
    // this half's two channels (16 + 2h, 17 + 2h) of both texels of a pair = dword h (first texel) and dword 2 + h (second) of the 16 bytes
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v ua = __builtin_bit_cast(u4v, d.u[0]), ub = __builtin_bit_cast(u4v, d.u[1]);
    const hpair a0p = unpack_h2(h ? ua.y : ua.x), a1p = unpack_h2(h ? ua.w : ua.z), b0p = unpack_h2(h ? ub.y : ub.x), b1p = unpack_h2(h ? ub.w : ub.z);
    const _Float16 r0x0a = a0p.x, r0x0b = a0p.y, r0x1a = a1p.x, r0x1b = a1p.y, r1x0a = b0p.x, r1x0b = b0p.y, r1x1a = b1p.x, r1x1b = b1p.y;
    // (same products in the same order as taps_acc16: tap (x0,y0), (x1,y0), (x0,y1), (x1,y1); a zero-weight product adds an exact zero)
    float sx = INIT ? (float)r0x0a * pa : fmaf((float)r0x0a, pa, acc[2].x), sy = INIT ? (float)r0x0b * pa : fmaf((float)r0x0b, pa, acc[2].y);
    sx = fmaf((float)r0x1a, pb, sx); sy = fmaf((float)r0x1b, pb, sy);
    sx = fmaf((float)r1x0a, pc, sx); sy = fmaf((float)r1x0b, pc, sy);
    acc[2].x = fmaf((float)r1x1a, pd, sx); acc[2].y = fmaf((float)r1x1b, pd, sy);
}
// colour taps from the half-precision RGBA copy: byte offsets of the two tap rows' x pairs inside one (batch, view) image
__device__ __forceinline__ RgbTaps rgb_taps16(int Ho, int Wo, float px, float py) {
    RgbTaps t = rgb_taps(Ho, Wo, px, py);
    t.o0 *= 2u; t.o1 *= 2u;   // 8 bytes per pixel instead of 4
    return t;
}
struct RgbData16 { half8v a, b; };
__device__ __forceinline__ void rgb_load16(const void* __restrict__ img16, const RgbTaps& t, RgbData16& d) {
    typedef half8v half8u __attribute__((aligned(8)));
    d.a = ldu<half8u>(img16, t.o0); d.b = ldu<half8u>(img16, t.o1);
}
__device__ __forceinline__ void rgb_combine16(const RgbTaps& t, const RgbData16& d, float rgb[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = fmaf((float)d.b[4 + c], t.w11, fmaf((float)d.b[c], t.w01, fmaf((float)d.a[4 + c], t.w10, (float)d.a[c] * t.w00)));
}
template <bool RGB = true>
__device__ __forceinline__ void gather_view16(const DevFrame& f, int bi, int v, int h, const float xyz[2][3], const float ctr[3],
                                              float ball, const float td[3], float4 feat[3], float dir[4], float rgb[2][3], int skip) {
#pragma clang fp contract(off)   // (every fused multiply-add below is written fmaf: see tex_coord_f)
    float sc[SRC_STRIDE];
    {
        const kfloat* scg = kptr(src_cam(f, bi, v));
        asm volatile("" : "+s"(scg));
#pragma unroll
        for (int i = 0; i < SRC_STRIDE; ++i) sc[i] = scg[i];
    }
    const bool do_rgb = RGB && !SKIPPED(skip, 1), do_tex = !SKIPPED(skip, 2);
    // ---- footprint -> mip level, texture coordinates (as gather_view)   bundle_sampler.py:340-353 -----------------------------------
    float cc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        cc[r] = fmaf(sc[S_E + 4 * r], ctr[0], fmaf(sc[S_E + 4 * r + 1], ctr[1], fmaf(sc[S_E + 4 * r + 2], ctr[2], sc[S_E + 4 * r + 3])));
    float d2 = fmaf(cc[2], cc[2], fmaf(cc[1], cc[1], cc[0] * cc[0]));
    float icz = frcp(cc[2]), ib = frcp(ball);
    float sec2 = d2 * icz * icz;
    float aa = fsqrt(fmaxf(fmaf(d2 * ib, ib, -1.f), 1e-12f)), cq = fsqrt(fmaxf(sec2 - 1.f, 1e-12f));
    float level = __builtin_amdgcn_logf(sec2 * frcp(aa + cq) * sc[S_IPIXR]);
    float ci[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) ci[r] = fmaf(sc[S_KS + 3 * r], cc[0], fmaf(sc[S_KS + 3 * r + 1], cc[1], sc[S_KS + 3 * r + 2] * cc[2]));
    float izc = frcp(fmaxf(ci[2], 1e-6f));
    float tu = ci[0] * izc * f.invW, tvv = ci[1] * izc * f.invH;
    int l0, l1; float frac;
    mip_select(level, f.levels, l0, l1, frac);
    // the pyramid block and the image block of this (batch, view): wave-uniform bases, 32-bit offsets below
    const size_t bvi = (size_t)bi * f.V + v;
    const void* pyr = (const void*)((const char*)f.pyr16 + bvi * f.pyrStride * 2);
    const void* img = (const void*)((const char*)f.pyr16 + IMG16_REL(f.pyrStride, (size_t)f.B * f.V) + bvi * ((size_t)f.Ho * f.Wo * 8));
    unsigned lo1 = f.lvlOff[1], lo2 = f.lvlOff[2], lo3 = f.lvlOff[3];
    asm volatile("" : "+s"(lo1), "+s"(lo2), "+s"(lo3));
    // ---- this half's level: texel indices and weights; then each half gets the other's ----------------------------------------
    const int lm = h ? l1 : l0;
    Taps16s t0, t1;
    {
        const Taps16s m = make_taps16s(tu, tvv, f.W >> lm, f.H >> lm, h ? frac : 1.f - frac);
        unsigned a0 = m.p00, a1 = m.p10, a2 = m.p01, a3 = m.p11, b0 = a0, b1 = a1, b2 = a2, b3 = a3;
        int a4 = m.edge, b4 = a4;
        float a5 = m.w00, a6 = m.w10, a7 = m.w01, a8 = m.w11, b5 = a5, b6 = a6, b7 = a7, b8 = a8;
        xchg9(a0, a1, a2, a3, a4, a5, a6, a7, a8, b0, b1, b2, b3, b4, b5, b6, b7, b8);   // a: half 0's values (level l0), b: half 1's (level l1)
        t0.p00 = a0; t0.p10 = a1; t0.p01 = a2; t0.p11 = a3; t0.edge = a4; t0.w00 = a5; t0.w10 = a6; t0.w01 = a7; t0.w11 = a8;
        t1.p00 = b0; t1.p10 = b1; t1.p01 = b2; t1.p11 = b3; t1.edge = b4; t1.w00 = b5; t1.w10 = b6; t1.w01 = b7; t1.w11 = b8;
    }
    const unsigned o0 = 2u * level_off(l0, lo1, lo2, lo3), hw0 = __umul24(f.W >> l0, f.H >> l0);
    const unsigned o1 = 2u * level_off(l1, lo1, lo2, lo3), hw1 = __umul24(f.W >> l1, f.H >> l1);
    RgbTaps rt[2];
    if constexpr (RGB) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {  // this half's two sub-rays   :327-337
        float im[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            im[r] = fmaf(sc[S_P + 4 * r], xyz[e][0], fmaf(sc[S_P + 4 * r + 1], xyz[e][1], fmaf(sc[S_P + 4 * r + 2], xyz[e][2], sc[S_P + 4 * r + 3])));
        float iz = frcp(fmaxf(im[2], 1e-6f));
        rt[e] = rgb_taps16(f.Ho, f.Wo, im[0] * iz, im[1] * iz);
    }
    }
    // ---- issue: level 0 and both sub-rays' colours in flight together (6 + 4 loads) ---------------------------------------------
    // (Measured, round 5 - profiles/r05/ab_f16_both_levels_in_flight.txt: level 1 fetched unconditionally - its products are exact zeros
    // when frac = 0 - and issued beside level 0, one exposed round trip per view instead of two, is SLOWER: c2 39.9 -> 41.2 us, c5 680 ->
    // 720.  The texture addresser is 67 % busy at c5; twelve more loads in flight per wave cost more than the second round trip.)
    const bool two = frac > 0.f && do_tex;
    TapData16s d0, d1;
    RgbData16 rd[2];
    if (do_tex) taps_load16s(pyr, t0, o0, hw0, h, d0);
    if (do_rgb) { rgb_load16(img, rt[0], rd[0]); rgb_load16(img, rt[1], rd[1]); }
    if (do_tex) taps_acc16s<true>(t0, d0, h, feat);
    else feat[0] = feat[1] = feat[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (two) taps_load16s(pyr, t1, o1, hw1, h, d1);
    if (do_rgb) { rgb_combine16(rt[0], rd[0], rgb[0]); rgb_combine16(rt[1], rd[1], rgb[1]); }
    else { for (int e = 0; e < 2; ++e) for (int c = 0; c < 3; ++c) rgb[e][c] = 0.f; }
    if (two) taps_acc16s<false>(t1, d1, h, feat);
    view_dir_code(ctr, td, sc + S_C, dir);
}

// Everything the views contribute to sample slot k of this wave's 32 bundles goes to LDS; returns
// false (and writes an empty composite record) when no lane has a sample in this slot.
// Sample slot k of the bundle q (already loaded) for this lane: the views' contributions go to the wave's staging area.
// vox[i] = voxel-feature channel 4h + i of this lane's sample.
// BB = 4: a 2 x 2 bundle (the sub-ray colours gathered here, two sub-rays per lane half); BB = 1: the bundle's CENTRE ray alone
// (load_bundle_center: bundle_size 1 / 4, whose colours k_bundle_colours gathers afterwards).
template <int PREC, int BB = 4>
__device__ __forceinline__ void slot_gather_q(const DevFrame& f, float* stage, const float* __restrict__ tc, const Bundle<BB>& q, int k, int bi,
                                              int j, int h, int skip, bool act, float& z, float vox[4]) {
#pragma clang fp contract(off)   // (the voxel taps' coordinates and weights: see tex_coord_f; the tap sums below are written fmaf)
    static_assert(BB == 4 || BB == 1, "a 2 x 2 bundle, or the centre ray of any other");
    constexpr bool RGB = BB == 4;
    const int V = f.V;
    float dn, ball, xyz[BB][3], ctr[3];
    bundle_sample<BB, true>(f, q, min(k, q.count - 1), z, dn, xyz, ctr, ball);  // bundle_sampler.py:246-263

    // Lanes WITHOUT a sample in this slot gather too (round 5): their bundle and slot are clamped to a real sample of the window (its
    // coordinates are as valid as any lane's), so the whole gather is straight-line code - no exec-masked region, no zero-initialised
    // merge values (20 v_mov per view) - and what they stage is finite.  Their results never reach an output (the composite masks them).
    vox[0] = vox[1] = vox[2] = vox[3] = 0.f;  // voxel feature, channels 4h..4h+3   :322-324
    if ((PREC != GDB_PREC_F32X || act) && !SKIPPED(skip, 4)) {
        float gx = gs_coord(q.u, f.W), gy = gs_coord(q.v, f.H), gz = gs_coord(dn, f.D);
        float xf = floorf(gx), yf = floorf(gy), zf = floorf(gz);
        float wy = gy - yf, wz = gz - zf;
        int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
        const unsigned plane = (unsigned)(f.H * f.W), cs = plane * (unsigned)f.D;
        const float* vol = f.feat_volume + (size_t)bi * GDB_CV * cs;
        const unsigned cb = h ? 4u * cs : 0u;
        if (f.W >= 2) {  // x pair in one 8-byte load (shifted left at the right edge, weight on its second element)
            int xp = min(x0, f.W - 2);
            float wxp = gx - (float)xp;
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    int yy = min(y0 + dy, f.H - 1), zz = min(z0 + dz, f.D - 1);  // a clamped tap carries weight 0
                    float wyz = (dy ? wy : 1.f - wy) * (dz ? wz : 1.f - wz);
                    float wa = (1.f - wxp) * wyz, wb = wxp * wyz;
                    unsigned off = 4u * (cb + __umul24(__umul24(zz, f.H) + yy, f.W) + xp);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        F2u pr = ldu_pin<F2u>(vol + c * cs, off);
                        vox[c] = fmaf(pr.y, wb, fmaf(pr.x, wa, vox[c]));
                    }
                }
        } else {
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    int yy = min(y0 + dy, f.H - 1), zz = min(z0 + dz, f.D - 1);
                    float wgt = (dy ? wy : 1.f - wy) * (dz ? wz : 1.f - wz);
                    unsigned off = 4u * (cb + __umul24(__umul24(zz, f.H) + yy, f.W));
#pragma unroll
                    for (int c = 0; c < 4; ++c) vox[c] = fmaf(ldu_pin<float>(vol + c * cs, off), wgt, vox[c]);
                }
        }
    }
    float xyzh[2][3];  // this half's two sub-ray points
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 3; ++r) xyzh[e][r] = RGB ? (h ? xyz[(2 + e) % BB][r] : xyz[e % BB][r]) : 0.f;

    float td[3];
    target_dir(ctr, tc + T_O, td);
    for (int v = 0; v < V; ++v) {
        float* st = stage + (size_t)v * stage_v<PREC>();
        float4 feat[3];
        float dir[4], rgb[2][3];
        if constexpr (PREC == GDB_PREC_F16) gather_view16<RGB>(f, bi, v, h, xyzh, ctr, ball, td, feat, dir, rgb, skip);
        else if constexpr (PREC == GDB_PREC_F32) gather_view<RGB>(f, bi, v, h, xyzh, ctr, ball, td, feat, dir, rgb, skip);
        else {   // split-f16 at three waves per SIMD has no register for the straight-line form (it spills one): lanes without a sample skip
            if (act) gather_view<RGB>(f, bi, v, h, xyzh, ctr, ball, td, feat, dir, rgb, skip);
            else {
                const float u = __builtin_nondeterministic_value(0.f);
                feat[0] = feat[1] = feat[2] = make_float4(u, u, u, u);
#pragma unroll
                for (int e = 0; e < 4; ++e) dir[e] = u;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int c = 0; c < 3; ++c) rgb[e][c] = u;
            }
        }
        constexpr int RF = row_feat<PREC>(), RD = row_dir<PREC>();
        // (RGB = false: the colour rows are written all the same - zeros - so that the blend pass of the MLP core reads finite values)
        if constexpr (PREC == GDB_PREC_F16) {  // colours as packed halves: row 2c + h = (sub-ray 2h, 2h + 1) of colour c
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            unsigned* su = (unsigned*)st;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const h2 p = {(_Float16)rgb[0][c], (_Float16)rgb[1][c]};
                su[(2 * c + h) * 32 + j] = __builtin_bit_cast(unsigned, p);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) st[(c * 4 + 2 * h + e) * 32 + j] = rgb[e][c];  // channel c*b²+sub   :337
        }
        const float* ff = (const float*)feat;
        if constexpr (PREC == GDB_PREC_F16) {   // channel pairs as packed halves: rows 4s + 2h, + 1 (channels 8s + 4h ..), row 8 + h (16 + 2h, 17 + 2h)
            unsigned* su = (unsigned*)st + j;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                su[(RF + 4 * s + 2 * h) * 32] = pack_h2(ff[4 * s], ff[4 * s + 1]);
                su[(RF + 4 * s + 2 * h + 1) * 32] = pack_h2(ff[4 * s + 2], ff[4 * s + 3]);
            }
            su[(RF + 8 + h) * 32] = pack_h2(feat[2].x, h == 0 ? feat[2].y : 0.f);   // half 1's .y is the padding channel 19: stored as 0
        } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[(RF + 8 * s + 4 * h + e) * 32 + j] = ff[4 * s + e];
        st[(RF + 16 + 2 * h) * 32 + j] = feat[2].x;           // channel 16 (half 0) / 18 (half 1)
        if (h == 0) st[(RF + 17) * 32 + j] = feat[2].y;       // channel 17; half 1's .y is the padding channel 19
        }
        if (PREC != GDB_PREC_F16) {  // four fp32 rows; both halves computed the same code, each stores two of them
            st[(RD + 2 * h) * 32 + j] = h ? dir[2] : dir[0];
            st[(RD + 2 * h + 1) * 32 + j] = h ? dir[3] : dir[1];
        } else if (h == 0) {
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const half2v p01 = {(_Float16)dir[0], (_Float16)dir[1]}, p23 = {(_Float16)dir[2], (_Float16)dir[3]};
            unsigned* su = (unsigned*)st;
            su[(RD + 0) * 32 + j] = __builtin_bit_cast(unsigned, p01);
            su[(RD + 1) * 32 + j] = __builtin_bit_cast(unsigned, p23);
        }
    }
}

// Workgroup-composite kernel: loads the bundle, decides whether the slot holds a sample; returns false (and writes an
// empty composite record) when no lane has one.
template <int PREC>
__device__ __forceinline__ bool slot_gather(const DevFrame& f, float* stage, float* ck, const float* __restrict__ tc, const float* rng, int k,
                                            int bi, int row, int x, bool inrow, int j, int h, int skip, bool& act, float& z, float vox[4]) {
    Bundle<4> q;
    load_bundle<4, true>(f, tc, bi, row, min(x, f.W - 1), q, rng);
    act = inrow && k < q.count;
    if (!__any(act)) {  // wave-uniform: nothing to sample in this slot
        for (int c = h; c < COMP_CH; c += 2) ck[c * COMP_LD + j] = 0.f;
        if (h == 0) ck[COMP_ALPHA + j] = 0.f;
        return false;
    }
    slot_gather_q<PREC>(f, stage, tc, q, k, bi, j, h, skip, act, z, vox);
    return true;
}

// NeRF MLP (nerf.py:58-115) of one slot from its staged views; writes the slot's composite record.
// `mf` = MFMA section of the packed weights, in global memory or (LDS-resident variant) in LDS.
// The operand fragments of a view (tail T0/T1, g_v) are rebuilt from LDS in each of the three view passes: keeping them
// in registers across the passes (a compile-time-V variant) needed 24 more VGPRs and spilled at 3 waves per SIMD.
// Weight fragments are software-pipelined by hand across the phases: every phase first issues the loads of the NEXT
// phase's fragments (they fly under this phase's MFMAs and VALU work), then computes with fragments loaded one phase
// earlier.  The fences keep the compiler from moving the loads any further (hoisting all ~45 of them spills), so
// without this each phase's first MFMA waits a full L2 round trip.
// Outputs per lane (j, h): bacc[i] = blended channel own_chan(h, i) of [rgbs | feat | rgb] (31 used), fhv[i] = ReLU'd feat_head
// channel 4h+i, sig = sigma pre-activation (valid in half 0).
template <bool X>
__device__ __forceinline__ void slot_mlp_core(const DevFrame& f, const float* __restrict__ mf, const float* stage, const float vox[4], int lane, int j,
                                              int h, float b_agg, float b_w2, unsigned* dbg, float bacc[16], float fhv[4], float& sig) {
    const int V = f.V;
    constexpr int STAGE_V = stage_v<X ? GDB_PREC_F32X : GDB_PREC_F16>();
    Frag<X> H1;  // k-step 1 of the [vox | im] operand: this half's 4 voxel channels, then the constant one
    {   // element 4 carries the biases of lr0, weight.0, feat_head and sigma (weights there are zero for half 1)
        const float v[8] = {vox[0], vox[1], vox[2], vox[3], 1.f, 0.f, 0.f, 0.f};
        if constexpr (X) H1 = split8(v);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) H1.hi[i] = (_Float16)v[i];
        }
    }
    f32x16 base;
    Frag<X> a_view, a_ga0, a_ga1;
    f32x16 w_agg;          // split-f16 path only
    half8 fca0, fca1;      // GDB_PREC_F16 only: fc (+) agg_w_fc on ReLU(G_v)
    {   LANE_KEYS();  // mean / unbiased variance of g_v over views   nerf.py:73
        a_view = load_fragx<X>(mf, F_VIEW, lane_o);
        const Frag<X> gv0 = load_fragx<X>(mf, F_GVAR, lane_o), gv1 = load_fragx<X>(mf, F_GVAR + 1, lane_o);
        const Frag<X> gm0 = load_fragx<X>(mf, F_GMEAN, lane_o), gm1 = load_fragx<X>(mf, F_GMEAN + 1, lane_o);
        a_ga0 = load_fragx<X>(mf, F_GA, lane_o); a_ga1 = load_fragx<X>(mf, F_GA + 1, lane_o);  // next phase
        if constexpr (X) w_agg = load_tab(mf, TD_AGG, h_o);
        else { fca0 = load_frag(mf, F_FCA, lane_o); fca1 = load_frag(mf, F_FCA + 1, lane_o); }
        // sum and sum of squares (var = (sum g^2 - V mean^2) / (V - 1)): two instructions per value and view, Welford's update four.
        // Cancellation: 1e-7 |g|^2 absolute on the variance - below what the operand rounding of these two paths does to it for
        // O(1) features (f16 operands: 5e-4 relative; operand pairs: 2.4e-7 relative); the exact-fp32 core (slot_mlp_core_f32)
        // accumulates around a shift instead, which needs twelve registers this core does not have at three waves per SIMD.
        f32x16 mean, m2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { mean[i] = 0.f; m2[i] = 0.f; }
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const Tail<X> tl = load_tail<X>(stage + (size_t)v * STAGE_V, j, h);
            f32x16 g = view_g<X>(tl, a_view);
#pragma unroll
            for (int i = 0; i < 12; ++i) { mean[i] += g[i]; m2[i] = fmaf(g[i], g[i], m2[i]); }
        }
        const float rv = frcp((float)V), iv = frcp((float)(V - 1));
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float s1 = mean[i];
            mean[i] = s1 * rv;
            m2[i] = (m2[i] - s1 * mean[i]) * iv;
        }
        m2[12] = 1.f;  // spare slot 24 of the variance operand: constant one that carries global_fc's bias
        // global_fc: bias + W[:,19:38] var + W[:,38:57] mean, shared by all views   nerf.py:77-78
        base = mm<X>(gv0, accf<0, false, X>(m2), zero16());
        base = mm<X>(gv1, accf<1, false, X>(m2), base);
        base = mm<X>(gm0, accf<0, false, X>(mean), base);
        base = mm<X>(gm1, accf<1, false, X>(mean), base);
    }
    PHASE_FENCE();
    STAMP(3);
    Frag<X> H0, lr0, lr1, lr2, lr3;
    if constexpr (!X) {
        // ---- GDB_PREC_F16 (round 5): per-view global feature, softmax over views, fc - all sums taken on the matrix pipe -------------
        // im = fc(sum_v a_v ReLU(G_v)) = b_fc + sum_v a_v (W_fc ReLU(G_v)): fc is linear, so the softmax-weighted sum moves BEHIND it and
        // runs over 16 values (8 registers) instead of 32; the logit agg_w_fc . ReLU(G_v) is row 16 / 20 of the same product (F_FCA), in
        // register 8 of both lane halves (no cross-half add).  ReLU(G_v) is rounded to f16 once, as the operand it becomes either way.
        // g_v = feat + ReLU(view_fc(dir)) is formed on the packed halves (its only use here is as an f16 operand): 18 instructions.
        LANE_KEYS();
        lr0 = load_fragx<X>(mf, F_LR0, lane_o); lr1 = load_fragx<X>(mf, F_LR0 + 1, lane_o);  // next phase
        lr2 = load_fragx<X>(mf, F_LR0 + 2, lane_o); lr3 = load_fragx<X>(mf, F_LR0 + 3, lane_o);
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        const f32x8 b_fc = ldu_pin<f32x8>(mf + TB_FC, (unsigned)h_o * 64u);   // registers 0..7 = rows < 16 of the [h][16] table
        float ia[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ia[i] = 0.f;
        float mx = -INFINITY, den = 0.f;
        const half2v z2 = {0, 0};
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const TailP tp = load_tailp(stage + (size_t)v * STAGE_V, j, h);
            const f32x16 a = MFMA(a_view.hi, tp.T1, zero16());
            unsigned gp[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                half2v r = {(_Float16)a[2 * q], (_Float16)a[2 * q + 1]};
                r = __builtin_elementwise_max(r, z2);
                gp[q] = __builtin_bit_cast(unsigned, (half2v)(r + __builtin_bit_cast(half2v, tp.u[q])));
            }
            f32x16 G = MFMA(a_ga0.hi, h8_of(gp[0], gp[1], gp[2], gp[3]), base);
            G = MFMA(a_ga1.hi, h8_of(gp[4], gp[5], 0u, 0u), G);
            const half8 R0 = acc_frag<0, true>(G), R1 = acc_frag<1, true>(G);
            f32x16 t = MFMA(fca0, R0, zero16());
            t = MFMA(fca1, R1, t);
            const float sv = relu1(t[8] + b_agg);  // nerf.py:79
            const float mn = fmaxf(mx, sv);
            const float sc_old = __expf(mx - mn), e = __expf(sv - mn);
            den = den * sc_old + e;
#pragma unroll
            for (int i = 0; i < 8; ++i) ia[i] = fmaf(e, t[i], ia[i] * sc_old);
            mx = mn;
        }
        const float r = frcp(den);
        half8 hh;
#pragma unroll
        for (int i = 0; i < 8; ++i) hh[i] = (_Float16)fmaf(ia[i], r, b_fc[i]);   // im   nerf.py:82
        const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        H0.hi = __builtin_elementwise_max(hh, z8);
    } else {
    f32x16 agg, im;
    Frag<X> fc0, fc1;
    {   LANE_KEYS();  // per-view global feature, softmax-weighted sum over views (online)   nerf.py:78-80
        fc0 = load_fragx<X>(mf, F_FC, lane_o); fc1 = load_fragx<X>(mf, F_FC + 1, lane_o);  // next phase
        im = load_tab(mf, TB_FC, h_o);
#pragma unroll
        for (int i = 0; i < 16; ++i) agg[i] = 0.f;
        float mx = -INFINITY, den = 0.f;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const f32x16 g = view_g<X>(load_tail<X>(stage + (size_t)v * STAGE_V, j, h), a_view);
            const Frag<X> g0 = accf<0, false, X>(g), g1 = accf<1, false, X>(g);
            f32x16 G = mm<X>(a_ga0, g0, base);
            G = mm<X>(a_ga1, g1, G);
            float sp = dot16_relu(G, w_agg);
            float sv = relu1(sp + __shfl_xor(sp, 32) + b_agg);  // nerf.py:79
            float mn = fmaxf(mx, sv);
            float sc_old = __expf(mx - mn), e = __expf(sv - mn);
            den = den * sc_old + e;
#pragma unroll
            for (int i = 0; i < 16; ++i) agg[i] = agg[i] * sc_old + e * relu1(G[i]);
            mx = mn;
        }
        float r = frcp(den);
#pragma unroll
        for (int i = 0; i < 16; ++i) agg[i] *= r;
    }
    PHASE_FENCE();
    {   LANE_KEYS();  // im = fc(agg)   nerf.py:82
        lr0 = load_fragx<X>(mf, F_LR0, lane_o); lr1 = load_fragx<X>(mf, F_LR0 + 1, lane_o);  // next phase
        lr2 = load_fragx<X>(mf, F_LR0 + 2, lane_o); lr3 = load_fragx<X>(mf, F_LR0 + 3, lane_o);
        im = mm<X>(fc0, accf<0, false, X>(agg), im);
        im = mm<X>(fc1, accf<1, false, X>(agg), im);
        H0 = accf<0, true, X>(im);
    }
    }
    PHASE_FENCE();
    STAMP(4);
    Frag<X> X00, X01, X10, X11, fhb, fh0, fh1, fh2, fh3;
    {   LANE_KEYS();  // x = ReLU(lr0([vox | im]))   nerf.py:100-101
        fhb = load_fragx<X>(mf, F_FHB, lane_o); fh0 = load_fragx<X>(mf, F_FH, lane_o); fh1 = load_fragx<X>(mf, F_FH + 1, lane_o);  // next phase
        fh2 = load_fragx<X>(mf, F_FH + 2, lane_o); fh3 = load_fragx<X>(mf, F_FH + 3, lane_o);
        f32x16 x0 = mm<X>(lr0, H0, zero16());
        x0 = mm<X>(lr1, H1, x0);
        X00 = accf<0, true, X>(x0); X01 = accf<1, true, X>(x0);
        f32x16 x1 = mm<X>(lr2, H0, zero16());
        x1 = mm<X>(lr3, H1, x1);
        X10 = accf<0, true, X>(x1); X11 = accf<1, true, X>(x1);
    }
    PHASE_FENCE();
    Frag<X> wa0, wa1, wa2, wa3, wb0, wb1;
    {   LANE_KEYS();  // rows 0..7 feat_head (nerf.py:112), row 8 sigma pre-activation (:102)
        wa0 = load_fragx<X>(mf, F_W0A + 0, lane_o); wa1 = load_fragx<X>(mf, F_W0A + 1, lane_o);  // next phase
        wa2 = load_fragx<X>(mf, F_W0A + 2, lane_o); wa3 = load_fragx<X>(mf, F_W0A + 3, lane_o);
        wb0 = load_fragx<X>(mf, F_W0B + 0, lane_o); wb1 = load_fragx<X>(mf, F_W0B + 1, lane_o);
        f32x16 fh = mm<X>(fhb, H1, zero16());
        fh = mm<X>(fh0, X00, fh);
        fh = mm<X>(fh1, X01, fh);
        fh = mm<X>(fh2, X10, fh);
        fh = mm<X>(fh3, X11, fh);
#pragma unroll
        for (int i = 0; i < 4; ++i) fhv[i] = relu1(fh[i]);
        sig = fh[4];
    }
    PHASE_FENCE();
    // shared part of weight.0: columns on x and on [vox | im]   nerf.py:106-109
    f32x16 hs0, hs1;
    Frag<X> wc0, wc1, wc2, wc3, wd0, wd1;
    {   LANE_KEYS();
        wc0 = load_fragx<X>(mf, F_W0A + 4, lane_o); wc1 = load_fragx<X>(mf, F_W0A + 5, lane_o);  // next phase
        wc2 = load_fragx<X>(mf, F_W0A + 6, lane_o); wc3 = load_fragx<X>(mf, F_W0A + 7, lane_o);
        wd0 = load_fragx<X>(mf, F_W0B + 2, lane_o); wd1 = load_fragx<X>(mf, F_W0B + 3, lane_o);
        hs0 = mm<X>(wa0, X00, zero16());
        hs0 = mm<X>(wa1, X01, hs0);
        hs0 = mm<X>(wa2, X10, hs0);
        hs0 = mm<X>(wa3, X11, hs0);
        hs0 = mm<X>(wb0, H0, hs0);
        hs0 = mm<X>(wb1, H1, hs0);
    }
    PHASE_FENCE();
    Frag<X> c00, c01, c10, c11;
    f32x16 w20, w21;                  // split-f16 path only
    half8 w2r0, w2r1, w2r2, w2r3;     // GDB_PREC_F16 only: weight.2 on ReLU(weight.0's output), as MFMA fragments
    {   LANE_KEYS();
        // operands of the per-view blend pass (next phase; loop-invariant there: loaded once per slot, not per view)
        c00 = load_fragx<X>(mf, F_W0C + 0, lane_o); c01 = load_fragx<X>(mf, F_W0C + 1, lane_o);
        c10 = load_fragx<X>(mf, F_W0C + 2, lane_o); c11 = load_fragx<X>(mf, F_W0C + 3, lane_o);
        if constexpr (X) { w20 = load_tab(mf, TD_W2, h_o); w21 = load_tab(mf, TD_W2 + 32, h_o); }
        else {
            w2r0 = load_frag(mf, F_W2R + 0, lane_o); w2r1 = load_frag(mf, F_W2R + 1, lane_o);
            w2r2 = load_frag(mf, F_W2R + 2, lane_o); w2r3 = load_frag(mf, F_W2R + 3, lane_o);
        }
        hs1 = mm<X>(wc0, X00, zero16());
        hs1 = mm<X>(wc1, X01, hs1);
        hs1 = mm<X>(wc2, X10, hs1);
        hs1 = mm<X>(wc3, X11, hs1);
        hs1 = mm<X>(wd0, H0, hs1);
        hs1 = mm<X>(wd1, H1, hs1);
    }
    PHASE_FENCE();
    STAMP(5);
    // per-view blend weight, softmax-weighted blend of [rgbs | feat | rgb]   nerf.py:108-110
#pragma unroll
    for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
    if constexpr (!X) {
        // ---- GDB_PREC_F16 (round 5) ------------------------------------------------------------------------------------------------
        // The logit weight.2 . ReLU(hv) is an MFMA on the ReLU'd, f16-rounded accumulators (F_W2R: rows 0 and 4, i.e. register 0 of both
        // lane halves) instead of a 64-term dot product on the vector ALU; and the softmax over views takes two passes - the logits first
        // (left in the view's direction row, dead once its tail operand is in registers), then exp(logit - max) x the staged values -
        // where the online form rescaled 16 running sums per view.
        constexpr int RD = row_dir<GDB_PREC_F16>();
        float mx = -INFINITY;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            float* st = const_cast<float*>(stage) + (size_t)v * STAGE_V;
            const TailP tp = load_tailp(st, j, h);
            f32x16 hv0 = MFMA(c00.hi, tp.T0, hs0);
            hv0 = MFMA(c01.hi, tp.T1, hv0);
            f32x16 hv1 = MFMA(c10.hi, tp.T0, hs1);
            hv1 = MFMA(c11.hi, tp.T1, hv1);
            const half8 R00 = acc_frag<0, true>(hv0), R01 = acc_frag<1, true>(hv0), R10 = acc_frag<0, true>(hv1), R11 = acc_frag<1, true>(hv1);
            f32x16 u = MFMA(w2r0, R00, zero16());   // (one chain of four: a second accumulator is 16 registers at this kernel's register peak - it spilled)
            u = MFMA(w2r1, R01, u);
            u = MFMA(w2r2, R10, u);
            u = MFMA(w2r3, R11, u);
            const float uv = relu1(u[0] + b_w2);  // nerf.py:109
            st[RD * 32 + j] = uv;                 // (both lane halves hold the same value and write the same word)
            mx = fmaxf(mx, uv);
        }
        float den = 0.f;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const float* st = stage + (size_t)v * STAGE_V;
            const float e = __expf(st[RD * 32 + j] - mx);
            den += e;
            float val[16];
            load_blend16<GDB_PREC_F16>(st, j, h, val);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = fmaf(e, val[i], bacc[i]);  // nerf.py:110
        }
        const float r = frcp(den);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] *= r;
    } else {
        float mx = -INFINITY, den = 0.f;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const float* st = stage + (size_t)v * STAGE_V;
            const Tail<X> t = load_tail<X>(st, j, h);
            const Frag<X>& T0 = t.T0; const Frag<X>& T1 = t.T1;
            float up;
            {
                f32x16 hv = mm<X>(c00, T0, hs0);
                hv = mm<X>(c01, T1, hv);
                up = dot16_relu(hv, w20);
            }
            {
                f32x16 hv = mm<X>(c10, T0, hs1);
                hv = mm<X>(c11, T1, hv);
                up += dot16_relu(hv, w21);
            }
            float uv = relu1(up + __shfl_xor(up, 32) + b_w2);  // nerf.py:109
            float mn = fmaxf(mx, uv);
            float sc_old = __expf(mx - mn), e = __expf(uv - mn);
            den = den * sc_old + e;
            float val[16];
            load_blend16<X ? GDB_PREC_F32X : GDB_PREC_F16>(st, j, h, val);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = bacc[i] * sc_old + e * val[i];  // nerf.py:110
            mx = mn;
        }
        float r = frcp(den);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] *= r;
    }
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    STAMP(6);
}

// ---- the same MLP in fp32: v_mfma_f32_32x32x2_f32 (GDB_PREC_F32) -------------------------------------------------
// Every layer is the reference's own fp32 arithmetic (nerf.py:84-115 never leaves fp32): the MFMA is bit for bit a
// k-ordered fmaf chain with one rounding per product (MI355X_MICROARCH.md, FP32-input MFMA), the accumulator starts
// from the layer's bias, and a finished accumulator register is the next layer's B operand as it stands (see the
// section layout above) — no conversion, no LDS, no lane movement; ReLU is one integer max per register.
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 load_quad(const float* __restrict__ m32, int q, int lane) {
#if defined(GDB_DIAG) && defined(GDB_XP_NOW) && GDB_XP_NOW == 1   // diagnostic build only - timing experiment (wrong results): no weight loads at all
    f32x4 r; asm volatile("" : "=v"(r)); return r;
#elif defined(GDB_DIAG) && defined(GDB_XP_NOW) && GDB_XP_NOW == 2  // diagnostic build only - timing experiment (wrong results): every weight load hits the same 1 KiB (L1-resident)
    return ldu_pin<f32x4>(m32 + (size_t)(q & 1) * 256, (unsigned)lane * 16u);
#else
    return ldu_pin<f32x4>(m32 + (size_t)q * 256, (unsigned)lane * 16u);
#endif
}
// acc += W[:, steps] · b over NS steps whose weight quads are already in registers
template <int NS>
__device__ __forceinline__ f32x16 chain32w(const f32x4* __restrict__ w, const float* __restrict__ b, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc = MFMA32(w[s >> 2][s & 3], b[s], acc);
    }
    return acc;
}
template <int NS>
__device__ __forceinline__ void load_quads(const float* __restrict__ m32, int q0, int lane, f32x4* w) {
#pragma unroll
    for (int q = 0; q < (NS + 3) / 4; ++q) w[q] = load_quad(m32, q0 + q, lane);
}
template <int NS>
__device__ __forceinline__ f32x16 chain32(const float* __restrict__ m32, int q0, int lane, const float* __restrict__ b, f32x16 acc) {
    f32x4 w[(NS + 3) / 4];
    load_quads<NS>(m32, q0, lane, w);
    return chain32w<NS>(w, b, acc);
}
// v_mfma_f32_16x16x4_f32 for the two layers with <= 16 output rows (see the section layout above)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// v_permlane16_swap_b32 a, b: the odd 16-lane rows of a trade places with the even rows of b, i.e. afterwards
//   a = [a.row0, b.row0, a.row2, b.row2]   b = [a.row1, b.row1, a.row3, b.row3]
// Inline asm, not __builtin_amdgcn_permlane16_swap: hipcc 7.2 hands back the builtin's FIRST result for both elements of the
// pair (read in the ISA: both uses name the same register).  The compiler's hazard recogniser does not look inside inline asm
// (DESIGN.md 4.1), so (1) every operand here must be the result of a plain VALU instruction, never of an MFMA, and (2) the
// s_nop pairs are the wait states hipcc itself places around the builtin (VALU write -> permlane read, permlane write -> use).
__device__ __forceinline__ void swap16_4(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3, float& b3) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\ts_nop 1"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
}
// v_permlane32_swap_b32 a, b: the upper 32 lanes of a trade places with the lower 32 lanes of b (same caveats)
__device__ __forceinline__ void swap32_3(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\ts_nop 1"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2));
}
__device__ __forceinline__ f32x4 load_tab16(const float* __restrict__ m32, int off, int lane) {
    return ldu_pin<f32x4>(m32 + off, (unsigned)lane * 16u);
}
// One 16-row layer on v_mfma_f32_16x16x4_f32: NG k-groups over the 2 NG VALU-produced registers x (swapped IN PLACE: x is dead
// afterwards as a 32x32 operand), accumulators d0 (samples 0..15) / d1 (samples 16..31) start from the bias table.
template <int NG>
__device__ __forceinline__ void chain16(const f32x4* __restrict__ w, float* __restrict__ x, f32x4& d0, f32x4& d1) {
    static_assert(NG % 4 == 0, "k-groups come in quads");
#pragma unroll
    for (int g = 0; g < NG; g += 4)
        swap16_4(x[2 * g], x[2 * g + 1], x[2 * g + 2], x[2 * g + 3], x[2 * g + 4], x[2 * g + 5], x[2 * g + 6], x[2 * g + 7]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        d0 = MFMA16(w[g >> 2][g & 3], x[2 * g], d0);
        d1 = MFMA16(w[g >> 2][g & 3], x[2 * g + 1], d1);
    }
}
// ReLU of both D tiles (the VALU step between the MFMAs and the swap), then the swap: e0[r] / e1[r] of lane (j, h) = output rows
// 8h + r / 8h + 4 + r of sample j
__device__ __forceinline__ void finish16(const f32x4& d0, const f32x4& d1, float e0[4], float e1[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { e0[r] = relu1(d0[r]); e1[r] = relu1(d1[r]); }
    swap16_4(e0[0], e1[0], e0[1], e1[1], e0[2], e1[2], e0[3], e1[3]);
}

// This lane's B-operand registers of one staged view: the 19-vector feat (+) rgb in accumulator layout (register 4s+e
// carries channel 8s + 4h + e) - where half 1 of registers 8..11 (channels 20..23: none) carries dir 0..3, the form the
// weight.0 tail chain takes (the chains on g_v have zero weights in those slots; half 0's register 11, channel 19, reads
// dir 0 and meets a zero weight everywhere) - and the two direction steps of view_fc (step s carries dir 2s + h).
struct Tail32 { float fv[12]; float d[2]; };
template <bool WITH_D>
__device__ __forceinline__ Tail32 load_tail32(const float* __restrict__ st, int j, int h) {
    constexpr int RF = row_feat<GDB_PREC_F32>(), RD = row_dir<GDB_PREC_F32>();
    Tail32 t;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) t.fv[4 * s + e] = st[(RF + 8 * s + 4 * h + e) * 32 + j];
    const float* st2 = st + h * ((RD - (RF + 16)) * 32);  // half 0: channels 16, 17, 18, (dir 0); half 1: dir 0..3
#pragma unroll
    for (int e = 0; e < 4; ++e) t.fv[8 + e] = st2[(RF + 16 + e) * 32 + j];
    if (WITH_D) {
        t.d[0] = st[(RD + h) * 32 + j];
        t.d[1] = st[(RD + 2 + h) * 32 + j];
    } else t.d[0] = t.d[1] = 0.f;
    return t;
}
// g_v = feat (+) rgb + ReLU(view_fc(dir)), registers 0..11 (accumulator layout)   nerf.py:69-71
// (view_fc's rows 19..31 are zero with a zero bias, so the slots that do not carry a channel pass through unchanged)
__device__ __forceinline__ void view_g32(const Tail32& t, const f32x4 q_view, const f32x16& b_view, float g[12]) {
    f32x16 a = MFMA32(q_view[0], t.d[0], b_view);
    a = MFMA32(q_view[1], t.d[1], a);
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = t.fv[i] + relu1(a[i]);
}

// Same contract as slot_mlp_core.  `m32` = f32-MFMA section of the packed weights (global memory, L2-resident).
// The matrix pipe is the bound here (276 MFMA-equivalents x 64 cycles per slot at V = 3), so the weight stream must never make
// it wait: as in the f16 core every phase first issues the loads of the NEXT phase's quads and bias tables — they fly under this
// phase's MFMA chain — and then computes with operands loaded a phase earlier.  How much is prefetched is set by the
// 168-register budget of three waves per SIMD.
__device__ __forceinline__ void slot_mlp_core_f32(const DevFrame& f, const float* __restrict__ m32, const float* stage, const float vox[4], int lane,
                                                  int j, int h, float b_agg, float b_w2, unsigned* dbg, float bacc[16], float fhv[4], float& sig) {
    const int V = f.V;
    constexpr int STAGE_V = stage_v<GDB_PREC_F32>();
    f32x16 base;
    f32x4 q_view;
    f32x16 b_view;
    f32x4 wg[3];
    {   LANE_KEYS();  // mean / unbiased variance of g_v over views   nerf.py:73
        q_view = load_quad(m32, Q_VIEW, lane_o);
        b_view = load_tab(m32, T32_VIEW, h_o);
        f32x4 wv[3], wm[2];
        load_quads<12>(m32, Q_GVAR, lane_o, wv);
        load_quads<8>(m32, Q_GMEAN, lane_o, wm);
        base = load_tab(m32, T32_GLOB, h_o);
        load_quads<12>(m32, Q_GA, lane_o, wg);  // next phase
        // Sums of d = g_v - g_0 and of d^2 (shifted data: mean = g_0 + sum d / V, var = (sum d^2 - (sum d)^2 / V) / (V - 1)).  Two
        // instructions per value and view like the plain sum / sum of squares (view 0 is the shift and adds nothing; the others pay
        // one subtraction), Welford's update four - but the cancellation is relative to the SPREAD of g over the views, not to its
        // magnitude (unnormalised FPN features of a real checkpoint; torch.var_mean is two-pass), and equal views give exactly 0.
        float g0[12], s1[12], s2[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        {
            const Tail32 tl = load_tail32<true>(stage, j, h);
            view_g32(tl, q_view, b_view, g0);
        }
#pragma unroll 1
        for (int v = 1; v < V; ++v) {
            const Tail32 tl = load_tail32<true>(stage + (size_t)v * STAGE_V, j, h);
            float g[12];
            view_g32(tl, q_view, b_view, g);
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float d = g[i] - g0[i]; s1[i] += d; s2[i] = fmaf(d, d, s2[i]); }
        }
        const float rv = frcp((float)V), iv = frcp((float)(V - 1));
        float mean[12], m2[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { const float md = s1[i] * rv; mean[i] = g0[i] + md; m2[i] = (s2[i] - s1[i] * md) * iv; }
        // mean channels 16..18 (half 0 of registers 8..10) move into half 1 of the variance registers 8..10, whose own
        // channels (20..22) do not exist: the mean chain then needs 8 steps instead of 12
        swap32_3(m2[8], mean[8], m2[9], mean[9], m2[10], mean[10]);
        // global_fc: bias + W[:,19:38] var + W[:,38:57] mean, shared by all views   nerf.py:77-78
        // (11 steps, not 12: register 11 carries channels 19 / 23, which do not exist - its weight step is all zeros, as is Q_GA's)
        base = chain32w<11>(wv, m2, base);
        base = chain32w<8>(wm, mean, base);
    }
    PHASE_FENCE();
    STAMP(3);
    float agg[16];
    f32x4 wfc[2];
    f32x4 bfc;
    {   LANE_KEYS();  // per-view global feature, softmax-weighted sum over views (online)   nerf.py:78-80
        const f32x16 w_agg = load_tab(m32, T32_AGG, h_o);
        load_quads<8>(m32, Q_FC, lane_o, wfc);  // next phase: fc
        bfc = load_tab16(m32, T16_FC, lane_o);
#pragma unroll
        for (int i = 0; i < 16; ++i) agg[i] = 0.f;
        float mx = -INFINITY, den = 0.f;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            float g[12];
            view_g32(load_tail32<true>(stage + (size_t)v * STAGE_V, j, h), q_view, b_view, g);
            const f32x16 G = chain32w<11>(wg, g, base);
            float sp = dot16_relu(G, w_agg);
            float sv = relu1(sp + __shfl_xor(sp, 32) + b_agg);  // nerf.py:79
            float mn = fmaxf(mx, sv);
            float sc_old = __expf(mx - mn), e = __expf(sv - mn);
            den = den * sc_old + e;
#pragma unroll
            for (int i = 0; i < 16; ++i) agg[i] = agg[i] * sc_old + e * relu1(G[i]);
            mx = mn;
        }
        float r = frcp(den);
#pragma unroll
        for (int i = 0; i < 16; ++i) agg[i] *= r;
    }
    PHASE_FENCE();
    float hb[12];  // [vox | im] operand: steps 0..7 = ReLU(im) registers 0..7, steps 8..11 = vox
    f32x4 wl0[3], wl1[3];
    f32x16 x0, x1;
    {   LANE_KEYS();  // im = ReLU(fc(agg))   nerf.py:82, on the 16-row MFMA
        load_quads<12>(m32, Q_LR0, lane_o, wl0); load_quads<12>(m32, Q_LR0 + 3, lane_o, wl1);  // next phase: lr0
        x0 = load_tab(m32, T32_LR0, h_o); x1 = load_tab(m32, T32_LR0 + 32, h_o);
        f32x4 d0 = bfc, d1 = bfc;
        chain16<8>(wfc, agg, d0, d1);
        finish16(d0, d1, hb, hb + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) hb[8 + i] = vox[i];
    }
    PHASE_FENCE();
    STAMP(4);
    float X[32];  // x = ReLU(lr0([vox | im])): [tile][register]   nerf.py:100-101
    f32x4 wa[8], wb[3];
    f32x16 hs0, hs1;
    {   LANE_KEYS();
        load_quads<32>(m32, Q_W0A, lane_o, wa); load_quads<12>(m32, Q_W0B, lane_o, wb);  // next phase: weight.0 rows 0..31
        hs0 = load_tab(m32, T32_W0, h_o);
        x0 = chain32w<12>(wl0, hb, x0);
        x1 = chain32w<12>(wl1, hb, x1);
#pragma unroll
        for (int i = 0; i < 16; ++i) { X[i] = relu1(x0[i]); X[16 + i] = relu1(x1[i]); }
    }
    PHASE_FENCE();
    // shared part of weight.0: bias + columns on x and on [vox | im]   nerf.py:106-109
    f32x4 wa2[8];
    {   LANE_KEYS();
        load_quads<24>(m32, Q_W0A + 8, lane_o, wa2);  // next phase: weight.0 rows 32..63, the first 6 of its 8 quads on x (the
        hs1 = load_tab(m32, T32_W0 + 32, h_o);        // last two and the [vox | im] quads follow there: 168 registers = 3 waves per SIMD)
        hs0 = chain32w<32>(wa, X, hs0);
        hs0 = chain32w<12>(wb, hb, hs0);
    }
    PHASE_FENCE();
    f32x4 wfh[4];
    f32x4 bfh;
    {   LANE_KEYS();
        load_quads<8>(m32, Q_W0A + 14, lane_o, wa2 + 6);
        load_quads<12>(m32, Q_W0B + 3, lane_o, wb);
        load_quads<16>(m32, Q_FH, lane_o, wfh);  // next phase: feat_head + sigma
        bfh = load_tab16(m32, T16_FH, lane_o);
        hs1 = chain32w<32>(wa2, X, hs1);
        hs1 = chain32w<12>(wb, hb, hs1);
    }
    PHASE_FENCE();
    f32x4 wc0[3];  // per-view blend pass operands: loop-invariant, loaded once per slot
    f32x16 w20;
    {   LANE_KEYS();  // feat_head (nerf.py:112) and the sigma pre-activation (:102) on the 16-row MFMA; x is dead afterwards
        load_quads<12>(m32, Q_W0C, lane_o, wc0);  // next phase
        w20 = load_tab(m32, T32_W2, h_o);
        f32x4 d0 = bfh, d1 = bfh;
        chain16<16>(wfh, X, d0, d1);
        float e1[4];
        finish16(d0, d1, fhv, e1);
        sig = e1[0] - e1[1];  // rows 4 / 5 = +s / -s (half 0): ReLU(s) - ReLU(-s) = s
    }
    PHASE_FENCE();
    STAMP(5);
    // per-view blend weight, softmax-weighted blend of [rgbs | feat | rgb] (online)   nerf.py:108-110
#pragma unroll
    for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
    {
        LANE_KEYS();
        f32x4 wc1[3];
        load_quads<12>(m32, Q_W0C + 3, lane_o, wc1);
        const f32x16 w21 = load_tab(m32, T32_W2 + 32, h_o);
        // Two passes over the views (round 5): the logits first - each left in the first direction row of its view, dead once the tail
        // operand is in registers - then exp(logit - max) x the staged values; the online form rescaled 16 running sums per view, and a
        // vector instruction is matrix time on the fp32 datapath.
        constexpr int RD = row_dir<GDB_PREC_F32>();
        float mx = -INFINITY, den = 0.f;
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            float* st = const_cast<float*>(stage) + (size_t)v * STAGE_V;
            const Tail32 t = load_tail32<false>(st, j, h);
            float up = dot16_relu(chain32w<12>(wc0, t.fv, hs0), w20);
            up += dot16_relu(chain32w<12>(wc1, t.fv, hs1), w21);
            const float uv = relu1(up + __shfl_xor(up, 32) + b_w2);  // nerf.py:109
            st[RD * 32 + j] = uv;   // (both lane halves hold the same value and write the same word)
            mx = fmaxf(mx, uv);
        }
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            const float* st = stage + (size_t)v * STAGE_V;
            const float e = __expf(st[RD * 32 + j] - mx);
            den += e;
            float val[16];
            load_blend16<GDB_PREC_F32>(st, j, h, val);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = fmaf(e, val[i], bacc[i]);  // nerf.py:110
        }
        float r = frcp(den);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] *= r;
    }
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    STAMP(6);
}

// alpha = 1 - exp(-softplus(sig)) (nerf.py:102 Softplus, utils.py:34) = 1 - 1/(1 + e^sig) = sigmoid(sig): one exp and one
// reciprocal instead of log1p(exp()) followed by another exp (beyond Softplus's threshold 20 the two differ by e^-40)
__device__ __forceinline__ float alpha_of(float sig) { return frcp(1.f + __expf(-sig)); }

// Workgroup-composite kernel: MLP of one slot, then the slot's composite record.
template <int PREC>
__device__ __forceinline__ void slot_mlp(const DevFrame& f, const float* __restrict__ mf, const float* stage, float* ck, bool act, float z,
                                         const float vox[4], int lane, int j, int h, float b_agg, float b_w2, unsigned* dbg) {
    float bacc[16], fhv[4], sig;
    if (PREC == GDB_PREC_F32) slot_mlp_core_f32(f, mf, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
    else slot_mlp_core<PREC == GDB_PREC_F32X>(f, mf, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
    // hand this slot to the composite (the record may alias this wave's staging area, dead by now)
    store_own16(ck, j, h, [&](int i) { return act ? bacc[i] : 0.f; });
#pragma unroll
    for (int i = 0; i < 4; ++i) ck[(NBLEND + 4 * h + i) * COMP_LD + j] = act ? fhv[i] : 0.f;
    if (h == 0) {
        float zz = f.inv_depth ? 1.f / z : z;  // network.py:83-84
        ck[NOUT * COMP_LD + j] = act ? zz : 0.f;
        ck[COMP_ALPHA + j] = act ? alpha_of(sig) : 0.f;
    }
}

// Composite of one segment from its S slot records in LDS (all threads of the team; two workgroup barriers inside):
// transmittance weights per bundle, normalised (utils.py:35-41), then the weighted sums (utils.py:109-119) written as
// the segment's contiguous (32 x 39) block, depth and opacity (network.py:88-89).  SC > 0: S == SC at compile time.
// PACKED: output rows are [feat 39 | depth | opacity] (41 floats, one contiguous run per segment; the multi-GPU gather unit).
template <int SC, bool PACKED>
__device__ __forceinline__ void composite(const FusedArgs& a, float* rec_team, size_t rec_stride, int S_rt, int tt, int tthreads,
                                          int seg, int row, int bi) {
    const DevFrame& f = a.f;
    const int S = SC > 0 ? SC : S_rt;
    constexpr int NK = SC > 0 ? SC : 1;
    constexpr int LDO = PACKED ? NOUT + 2 : NOUT;
    if (tt < 32) {
        if (SC > 0) {
            float al[NK], w[NK], T = 1.f, sum = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) al[k] = rec_team[(size_t)k * rec_stride + COMP_ALPHA + tt];
#pragma unroll
            for (int k = 0; k < NK; ++k) { w[k] = al[k] * T; T = T * (1.f - al[k]); sum += w[k]; }
            const float rden = 1.f / fmaxf(sum, 1e-6f);  // one division per bundle; w_k * (1/den) is within an ulp of w_k / den
#pragma unroll
            for (int k = 0; k < NK; ++k) rec_team[(size_t)k * rec_stride + COMP_WN + tt] = w[k] * rden;
        } else {
            float T = 1.f, sum = 0.f;
            for (int k = 0; k < S; ++k) {
                float* rk = rec_team + (size_t)k * rec_stride;
                float al = rk[COMP_ALPHA + tt];
                float w = al * T;
                T = T * (1.f - al);
                rk[COMP_WN + tt] = w;
                sum += w;
            }
            const float rden = 1.f / fmaxf(sum, 1e-6f);
            for (int k = 0; k < S; ++k) {
                float* rk = rec_team + (size_t)k * rec_stride;
                rk[COMP_WN + tt] = rk[COMP_WN + tt] * rden;
            }
        }
    }
    __syncthreads();
    const int nvalid = min(32, f.W - seg * 32);
    const size_t b0 = ((size_t)bi * f.H + row) * f.W + (size_t)seg * 32;
    // sum over the slots of weight x record row cr of bundle jj (is_op: of the weight alone = opacity); one code path for both
    // output layouts, so that they agree bit for bit
    auto chan_sum = [&](int cr, int jj, bool is_op) {
        float acc = 0.f;
        if (SC > 0) {
            float v[NK], w[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { const float* rk = rec_team + (size_t)k * rec_stride; v[k] = rk[cr * COMP_LD + jj]; w[k] = rk[COMP_WN + jj]; }
#pragma unroll
            for (int k = 0; k < NK; ++k) acc = fmaf(is_op ? 1.f : v[k], w[k], acc);
        } else {
            for (int k = 0; k < S; ++k) {
                const float* rk = rec_team + (size_t)k * rec_stride;
                acc = fmaf(is_op ? 1.f : rk[cr * COMP_LD + jj], rk[COMP_WN + jj], acc);
            }
        }
        return acc;
    };
    for (int qi = tt; qi < nvalid * LDO; qi += tthreads) {
        int jj = qi / LDO, c = qi - jj * LDO;
        const bool is_op = PACKED && c == NOUT + 1;  // opacity = sum of the weights; row NOUT of a record is z
        float acc = chan_sum(is_op ? 0 : c, jj, is_op);
        if (PACKED && c == NOUT && f.inv_depth) acc = 1.f / acc;  // network.py:88-89
        a.bf[b0 * LDO + qi] = acc;
    }
    if (!PACKED && tt < 64) {
        int jj = tt & 31, which = tt >> 5;
        if (jj < nvalid) {
            const float acc = chan_sum(which ? 0 : NOUT, jj, which != 0);
            if (which) a.opac[b0 + jj] = acc;
            else a.depth[b0 + jj] = f.inv_depth ? 1.f / acc : acc;  // network.py:88-89
        }
    }
}

// Workgroup = one 32-bundle segment x S sample slots, one wave per slot; LOOP: fewer waves than slots, the waves loop
// over slots.  The MLP weights come from global memory (L2-resident): an LDS-resident copy shared by a 9-12 wave
// workgroup was measured 25-30 % slower (it caps the CU at 6-9 waves), see DESIGN.md.
// PREC: GDB_PREC_F16 (f16 MFMA operands, 3 waves per SIMD) or GDB_PREC_F32 (fp32 MFMA, matrix-pipe-bound: 2 waves per SIMD).
template <bool LOOP, int WAVES, int PREC>
__global__ void __launch_bounds__(64 * WAVES, (LOOP || WAVES > 4) ? 2 : 3) k_render_fused(FusedArgs a) {
    const DevFrame& f = a.f;
    float* smem = (float*)smem4;
    // the wave index is uniform but derived from threadIdx: without readfirstlane everything computed from it (team,
    // segment, row, batch index, camera and staging pointers) sits in VGPRs and the camera blocks are vector loads
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int V = f.V, S = f.S_max;
    unsigned* dbg = a.dbg; (void)dbg;

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD one
    // contiguous band of tiles so vertically adjacent rows hit the same L2.
    const int chunk = (a.ntiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= a.ntiles) return;  // whole workgroup leaves together
    const int k0 = wid;
    const int seg = tile % a.nseg, rr = tile / a.nseg;
    const int row = a.row_begin + rr % a.nrows, bi = rr / a.nrows;
    const int x = seg * 32 + j;
    const bool inrow = x < f.W;

    // LDS: [per-wave staging V x stage_v] [composite records unless aliased].  With one slot per wave the record of
    // slot k reuses wave k's own staging area, dead by then.
    const size_t wave_fl = (size_t)V * stage_v<PREC>();
    float* stage = smem + (size_t)wid * wave_fl;
    const bool alias = a.alias != 0;
    const size_t rec_stride = alias ? wave_fl : (size_t)COMP_REC;
    float* rec_team = alias ? smem : smem + (size_t)nw * wave_fl;
    float rng[4];  // this lane's bundle ranges: issued first, the scalar loads below fly under their latency
    load_ranges(f, bi, row, min(x, f.W - 1), rng);
    float tc[TAR_STRIDE];  // target camera block of this wave's batch entry, in SGPRs
    {
        const kfloat* tcg = kptr(tar_cam(f, bi));
#pragma unroll
        for (int i = 0; i < TAR_STRIDE; ++i) tc[i] = tcg[i];
    }
    // MFMA section of the packed weights for this precision, and the two scalar biases (agg_w_fc, weight.2)
    const float* mfg = a.pw + PW_FP32_FLOATS + (PREC == GDB_PREC_F32 ? MFMA_FLOATS : 0);
    const float b_agg = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BAGG : TS_BAGG], b_w2 = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BW2 : TS_BW2];

    STAMP(0);
    if (LOOP) {
        for (int k = k0; k < S; k += nw) {
            float* ck = rec_team + (size_t)k * rec_stride;
            bool act; float z; float vox[4];
            if (slot_gather<PREC>(f, stage, ck, tc, rng, k, bi, row, x, inrow, j, h, a.skip, act, z, vox)) {
                __builtin_amdgcn_wave_barrier();
                PHASE_FENCE();
                if (!SKIPPED(a.skip, 8)) slot_mlp<PREC>(f, mfg, stage, ck, act, z, vox, lane, j, h, b_agg, b_w2, dbg);
            }
        }
    } else {  // one slot per wave: straight-line code, nothing to hoist out of a loop
        float* ck = rec_team + (size_t)k0 * rec_stride;
        bool act; float z; float vox[4];
        const bool any = slot_gather<PREC>(f, stage, ck, tc, rng, k0, bi, row, x, inrow, j, h, a.skip, act, z, vox);
        STAMP(2);
        __builtin_amdgcn_wave_barrier();
        PHASE_FENCE();
        if (any && !SKIPPED(a.skip, 8)) slot_mlp<PREC>(f, mfg, stage, ck, act, z, vox, lane, j, h, b_agg, b_w2, dbg);
    }
    STAMP(7);
    __syncthreads();
    STAMP(8);
    const int tt = threadIdx.x, tthreads = (int)blockDim.x;
    // The slot count is tiny: with it as a compile-time constant the S reads of a sum are all in flight before the first
    // is used (as a runtime loop every iteration waited for its own LDS read).
    if (a.ldo == NOUT) {
        switch (S) {
            case 1: composite<1, false>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
            case 2: composite<2, false>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
            case 3: composite<3, false>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
            case 4: composite<4, false>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
            default: composite<0, false>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
        }
    } else {
        switch (S) {
            case 3: composite<3, true>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
            default: composite<0, true>(a, rec_team, rec_stride, S, tt, tthreads, seg, row, bi); break;
        }
    }
    STAMP(9);
}

// ---------------------------------------------------------------------------------------------------------------
// One wave = one 32-bundle row segment and ALL its sample slots, one after the other; the composite runs in registers.
// Against the workgroup-composite kernel above this drops the per-slot LDS record, the barrier wait for the slowest
// slot and the composite pass, and a frame's waves fit on the chip in fewer rounds (c2: 2560 waves, one round).
// Lane (j, h) owns blended channels own_chan(h, 0..15) and feat_head channels 4h..4h+3 of bundle j; transmittance, weight
// sum and depth are kept by both halves.  utils.py:35-41 (weights), :109-119 (sums), network.py:83-89 (depth).
// WPS = waves per SIMD the register allocation is held to: 3 where LDS admits 12 waves per CU (V <= 3 at f16), else 2.
template <int PREC, int WPS>
__global__ void __launch_bounds__(64, WPS) k_render_solo(FusedArgs a_) {
    float* stage = (float*)smem4;  // V x stage_v floats; reused for the output transpose at the end
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    // Everything wave-uniform (kernel arguments, camera block, pointers) AND the bundle itself (its ranges, its sample count)
    // is re-derived inside each slot iteration from an opaque pointer to the kernel arguments: as loop invariants they would
    // be live across the whole loop body, and only the composite state (22 registers) may live across slots if the body is
    // to keep its 3 waves per SIMD without spilling.
    // (the kernel-argument segment itself, constant address space: &a_ would be a private copy of the struct)
    typedef const FusedArgs __attribute__((address_space(4))) KArgs;
    KArgs* const ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    int seg, row, bi;
    {
        const FusedArgs& a = *(const FusedArgs*)ap;
        const int chunk = (a.ntiles + 7) >> 3;  // XCD-aware tile order, as above
        const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        if (tile >= a.ntiles) return;
        seg = tile % a.nseg;
        const int rr = tile / a.nseg;
        row = a.row_begin + rr % a.nrows; bi = rr / a.nrows;
    }
    float oacc[16], of[4], dz = 0.f, T = 1.f, wsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) of[i] = 0.f;
    const int S = ap->f.S_max;
    for (int k = 0; k < S; ++k) {
        wave_prio<GDB_XP_PRIO_SOLO != 0>(k + 1 >= S, true);
        KArgs* apk = ap;
        asm volatile("" : "+s"(apk));  // nothing read through apk is loop-invariant to the compiler
        // (a by-value copy per slot, as k_render_dense makes per tile, LOSES here: c5 f16 926 -> 970 us - this kernel has no register to spare)
        const FusedArgs& a = *(const FusedArgs*)apk;
        const DevFrame& f = a.f;
        unsigned* dbg = a.dbg; (void)dbg;
        seg = __builtin_amdgcn_readfirstlane(seg); row = __builtin_amdgcn_readfirstlane(row); bi = __builtin_amdgcn_readfirstlane(bi);
        const int x = seg * 32 + j;
        float tc[TAR_STRIDE];
        {
            const kfloat* tcg = kptr(tar_cam(f, bi));
            asm volatile("" : "+s"(tcg));
#pragma unroll
            for (int i = 0; i < TAR_STRIDE; ++i) tc[i] = tcg[i];
        }
        Bundle<4> q;  // recomputed per slot (ranges re-read: an L1 hit): cheaper than ~30 registers live across the loop
        load_bundle<4, true>(f, tc, bi, row, min(x, f.W - 1), q);
        const bool act = x < f.W && k < q.count;
        if (!__any(act)) break;  // a bundle's samples are slots 0..count-1: nothing left in this segment
        const float* mfg = a.pw + PW_FP32_FLOATS + (PREC == GDB_PREC_F32 ? MFMA_FLOATS : 0);
        const float b_agg = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BAGG : TS_BAGG], b_w2 = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BW2 : TS_BW2];
        float z; float vox[4];
        slot_gather_q<PREC>(f, stage, tc, q, k, bi, j, h, a.skip, act, z, vox);
        wave_prio<GDB_XP_PRIO_SOLO != 0>(k + 1 >= S, false);
        __builtin_amdgcn_wave_barrier();
        PHASE_FENCE();
        float bacc[16], fhv[4], sig;
        if (PREC == GDB_PREC_F32) slot_mlp_core_f32(f, mfg, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
        else slot_mlp_core<PREC == GDB_PREC_F32X>(f, mfg, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
        const float sig0 = __shfl(sig, j);  // sigma sits in half 0
        if (act) {  // lanes without a sample hold unspecified MLP outputs: keep them out of the sums
            const float al = alpha_of(sig0);
            const float w = al * T;
            T = T * (1.f - al);
            wsum += w;
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[i] = fmaf(w, bacc[i], oacc[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) of[i] = fmaf(w, fhv[i], of[i]);
            dz = fmaf(w, f.inv_depth ? 1.f / z : z, dz);
        }
        PHASE_FENCE();
    }
    // normalise, transpose through LDS, store the segment's (32 x 39) block as one contiguous run
    const FusedArgs& a = *(const FusedArgs*)ap;
    const DevFrame& f = a.f;
    const float rden = 1.f / fmaxf(wsum, 1e-6f);
    __builtin_amdgcn_wave_barrier();
    float* o = stage;  // the output record: [32 bundles][ld] (+ depth, opacity behind it in the three-tensor form)
    const int ld = a.ldo;
    out_store_own16(o, ld, j, h, [&](int i) { return oacc[i] * rden; });
#pragma unroll
    for (int i = 0; i < 4; ++i) o[j * ld + NBLEND + 4 * h + i] = of[i] * rden;
    if (h == 0) {
        const float d = dz * rden;
        *out_depth_slot(o, ld, j) = f.inv_depth ? 1.f / d : d;  // network.py:88-89
        *out_opac_slot(o, ld, j) = wsum * rden;
    }
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    const int nvalid = min(32, f.W - seg * 32);
    const size_t b0 = ((size_t)bi * f.H + row) * f.W + (size_t)seg * 32;
    out_copy(a, o, b0, nvalid, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Dense schedule (GDB_SCHED_DENSE): the reference's flat, compacted sample list (bundle_sampler.py:182-189) cut into waves.
// One wave = one window of a bundle-map row: a run of whole consecutive bundles holding at most 32 samples.  The plan (plan_row:
// k_prepare or k_plan) lists the row's samples in the reference's order, entry s = [bundle | slot | count], and cuts the row into
// windows [start_w, start_{w+1}) - greedily (a window ends only where the next bundle would not fit) or at fixed offsets (window w
// = the bundles whose first sample falls into [L w, L (w + 1)), L = 33 - S_max), see plan_row.  The wave of window w reads the
// entries start_w + j: lane j IS that sample.  No per-wave count, scan or LDS map.
// Lane (j, h): j = sample, h = half as everywhere.  Nearly every lane carries a sample, where the slot schedules leave a lane idle
// whenever its bundle has fewer samples than the slot index: c2 79.6 % of the slot lanes busy -> 92.2 % (fixed cut; greedy 95.6 %),
// c3 79.6 -> 96.2 % (greedy), c4 (S_max 6) 44.4 -> 95.5 % (greedy; fixed cut 83.4 %).
// The composite runs in registers across the lanes of a bundle (its samples are consecutive lanes): transmittance by looking
// back over the earlier samples, the 22 weighted sums by a segmented suffix sum - for S_max <= 4 a Horner chain of DPP
// wave shifts (acc <- v + next(acc): two VALU instructions per value and step, no LDS crossbar), beyond that log2(S_max)
// ds_bpermute doubling steps -, the bundle's first lane normalises and hands the row to the LDS transpose.
// utils.py:35-41, :109-119, network.py:83-89.
// NWG = waves per workgroup (independent windows, no workgroup barrier anywhere): LDS is allocated in 1280-byte granules, and
// two waves' areas in one allocation can fit where single ones lose a wave per CU to the rounding (fp32 staging, V = 3: 13,440 B
// per wave = 11 one-wave workgroups per CU, but 6 two-wave ones = 12 waves).
// lane i <- lane i + 1 / i - 1 of the wave (0 past the ends): gfx9 DPP wave shifts
__device__ __forceinline__ float wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// Persistent tiles (round 4).  The launch is exactly the resident grid - as many workgroups as the chip holds at once (CUs x the
// workgroups LDS and registers admit per CU, launch_dense_n) - and every wave WALKS tiles: XCD x owns the contiguous band
// [x chunk, (x + 1) chunk) of the dense tile numbering, and the wave in slot s of that XCD renders tiles x chunk + s, + stride,
// + 2 stride, ... (stride = the XCD's wave slots).  Static, so deterministic; no atomics, no cross-workgroup dependency; row
// strips and batch items (grid y) stay bit-exact because a tile is still one window of one row.  Against one workgroup per tile
// (round 3) no wave slot waits for the dispatcher between two tiles: the occupancy trace of that form showed the launch drop from
// 3.0 to 2.05 resident waves per SIMD when the first round of workgroups ended together (profiles/r03/stamps_f32_schedule3.txt).
// Tiles are numbered densely over the windows IN USE (plan_row leaves the count per row in WsLayout::nwinOff); a wave turns a
// tile index into (row, window) with one scan of those counts - four rows per lane, 256 rows per round - restricted to the rows
// [rlo, rhi) this launch renders.  Everything comes out wave-uniform.
__device__ __forceinline__ int4 dense_counts(const DevFrame& f, int rlo, int rhi, int c0, int lane, int& s4) {
    // this lane's four rows of round c0: clamped to what the plan was sized for, zero outside the strip
    const int q0 = rlo >> 2, nq = ((rhi + 3) >> 2) - q0;  // int4s of the count array that touch the strip
    const int q = c0 + lane;
    int4 n = make_int4(0, 0, 0, 0);
    if (q < nq) n = ((const int4*)f.nwin)[q0 + q];
    const int r4 = (q0 + q) << 2;
    n.x = (r4 + 0 >= rlo && r4 + 0 < rhi) ? min(max(n.x, 0), f.planMW) : 0;
    n.y = (r4 + 1 >= rlo && r4 + 1 < rhi) ? min(max(n.y, 0), f.planMW) : 0;
    n.z = (r4 + 2 >= rlo && r4 + 2 < rhi) ? min(max(n.z, 0), f.planMW) : 0;
    n.w = (r4 + 3 >= rlo && r4 + 3 < rhi) ? min(max(n.w, 0), f.planMW) : 0;
    s4 = n.x + n.y + n.z + n.w;
    return n;
}
__device__ __forceinline__ int wave_scan_incl(int v, int lane) {  // inclusive prefix over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(v, d); if (lane >= d) v += u; }
    return v;
}
// live tiles of the rows [rlo, rhi)
__device__ __forceinline__ int dense_total(const DevFrame& f, int rlo, int rhi, int lane) {
    const int nq = ((rhi + 3) >> 2) - (rlo >> 2);
    int T = 0;
    for (int c0 = 0; c0 < nq; c0 += 64) {
        int s; dense_counts(f, rlo, rhi, c0, lane, s);
        T += __shfl(wave_scan_incl(s, lane), 63);
    }
    return T;
}
// This wave's tiles - slot `slot` of XCD `xcd`: tiles xcd chunk + slot, + stride, ... inside the XCD's band [xcd chunk, (xcd + 1) chunk),
// chunk = ceil(T / 8), at most 64 of them (the launcher sizes the grid so) - as descriptors row << 16 | window, one per LANE of a
// single register: the scan of the per-row window counts runs once per wave, a tile of the walk then costs one v_readlane.
// (Looked up per tile - a reload of the counts, a 6-step scan, a ballot - it was ~150 vector instructions per tile, 4 % of the
// wave's, read off the ISA.)  Returns the number of descriptors written; rows are the global row index (< 65536, checked by the
// launcher), windows < planMW <= 65535.
__device__ __forceinline__ int dense_fill(const DevFrame& f, int rlo, int rhi, int lane, int xcd, int slot, int stride, unsigned& vdesc) {
    const int q0 = rlo >> 2, nq = ((rhi + 3) >> 2) - q0;
    int s4;
    int4 n = dense_counts(f, rlo, rhi, 0, lane, s4);
    int incl = wave_scan_incl(s4, lane);
    // the total: this round's for strips of up to 256 rows (every BASELINE bundle map but c3 / c4 / c5), else one more pass over the counts
    const int T = nq <= 64 ? __builtin_amdgcn_readlane(incl, 63) : __builtin_amdgcn_readfirstlane(dense_total(f, rlo, rhi, lane));
    const int chunk = (T + 7) >> 3, t0 = xcd * chunk + slot, tend = min(T, (xcd + 1) * chunk);
    int run = 0, i = 0;
    vdesc = 0xFFFFFFFFu;
    for (int c0 = 0; c0 < nq && i < 64; c0 += 64) {
        if (c0) { n = dense_counts(f, rlo, rhi, c0, lane, s4); incl = wave_scan_incl(s4, lane); }
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        for (; i < 64; ++i) {   // this wave's tiles that fall into the rows of this round
            const int t = t0 + i * stride;
            if (t >= tend || t - run >= tot) break;
            const int tl = t - run;
            const int ls = __builtin_ctzll(__ballot(tl < incl));  // the lane whose four rows hold the tile
            const int ex = __builtin_amdgcn_readlane(incl - s4, ls);
            const int nx = __builtin_amdgcn_readlane(n.x, ls), ny = __builtin_amdgcn_readlane(n.y, ls), nz = __builtin_amdgcn_readlane(n.z, ls);
            int r = (q0 + c0 + ls) << 2, rem = tl - ex;
            if (rem >= nx) { rem -= nx; ++r; if (rem >= ny) { rem -= ny; ++r; if (rem >= nz) { rem -= nz; ++r; } } }
            if (lane == i) vdesc = ((unsigned)r << 16) | ((unsigned)rem & 0xFFFFu);   // (a v_cndmask on a wave-uniform value)
        }
        if (t0 + i * stride >= tend) break;
        run += tot;
    }
    return i;
}

// ---------------------------------------------------------------------------------------------------------------
// Flat schedule (GDB_SCHED_FLAT): the rows' compacted sample lists read as ONE list - literally the reference's flat sample list
// (bundle_sampler.py:182-189) - cut into windows of exactly 32 consecutive samples: every lane of every wave but the list's last
// carries a sample (the window plan of the dense schedule keeps bundles whole: 92-96 % of the lanes).  c2: 6,112 tiles instead of
// 6,631 - which also is at most TWO tiles per resident wave slot (3,072), where 6,631 left a third, nearly empty round that a lone
// wave per SIMD crawled through.
// The price: a window may begin or end INSIDE a bundle.  Such a bundle is composited by neither wave's in-register composite:
// both leave the per-sample records of their part (20 pre-weight values per lane half, alpha, the depth term) in the workspace,
// keyed by the window boundary, and whichever of the two waves ARRIVES LAST at the boundary's counter composites the bundle from
// those records (flat_fix_boundary) with exactly the arithmetic of the in-wave composite (same products, same order of sums), so
// that a bundle's result depends neither on where the windows fall nor on which wave arrived last: row strips of the frame stay
// bit-identical to the full render and to GDB_SCHED_DENSE.  (Until round 5 a second launch, k_flat_fix, did that: 5 us of launch
// ramp and two dependent memory round trips behind every render.)
// A window may also span bundle-map rows (the row of a lane is per-lane; a row holds >= W samples, so two rows in all but tiny maps).
// Tiles are looked up once per wave like the dense schedule's: descriptor = first row << 16 | sample offset inside that row.
__device__ __forceinline__ int4 flat_counts(const DevFrame& f, int rlo, int rhi, int c0, int lane, int& s4) {
    const int q0 = rlo >> 2, nq = ((rhi + 3) >> 2) - q0;
    const int q = c0 + lane;
    int4 n = make_int4(0, 0, 0, 0);
    if (q < nq) n = ((const int4*)f.nsamp)[q0 + q];
    const int r4 = (q0 + q) << 2, cap = f.smapStride - 32;   // (a row's list holds at most W S_max <= smapStride - 32 entries)
    n.x = (r4 + 0 >= rlo && r4 + 0 < rhi) ? min(max(n.x, 0), cap) : 0;
    n.y = (r4 + 1 >= rlo && r4 + 1 < rhi) ? min(max(n.y, 0), cap) : 0;
    n.z = (r4 + 2 >= rlo && r4 + 2 < rhi) ? min(max(n.z, 0), cap) : 0;
    n.w = (r4 + 3 >= rlo && r4 + 3 < rhi) ? min(max(n.w, 0), cap) : 0;
    s4 = n.x + n.y + n.z + n.w;
    return n;
}
// samples of the rows [rlo, rhi)
__device__ __forceinline__ int flat_total(const DevFrame& f, int rlo, int rhi, int lane) {
    const int nq = ((rhi + 3) >> 2) - (rlo >> 2);
    int T = 0;
    for (int c0 = 0; c0 < nq; c0 += 64) {
        int s; flat_counts(f, rlo, rhi, c0, lane, s);
        T += __shfl(wave_scan_incl(s, lane), 63);
    }
    return T;
}
// This wave's tiles (slot `slot` of XCD `xcd`, as dense_fill) -> descriptors, one per lane; t0 = its first tile's index in the launch.
__device__ __forceinline__ int flat_fill(const DevFrame& f, int rlo, int rhi, int lane, int xcd, int slot, int stride, unsigned& vdesc, int& t0_out) {
    const int q0 = rlo >> 2, nq = ((rhi + 3) >> 2) - q0;
    int s4;
    int4 n = flat_counts(f, rlo, rhi, 0, lane, s4);
    int incl = wave_scan_incl(s4, lane);
    const int NS = nq <= 64 ? __builtin_amdgcn_readlane(incl, 63) : __builtin_amdgcn_readfirstlane(flat_total(f, rlo, rhi, lane));
    const int T = (NS + 31) >> 5;
    const int chunk = (T + 7) >> 3, t0 = xcd * chunk + slot, tend = min(T, (xcd + 1) * chunk);
    t0_out = t0;
    int run = 0, i = 0;
    vdesc = 0xFFFFFFFFu;
    for (int c0 = 0; c0 < nq && i < 64; c0 += 64) {
        if (c0) { n = flat_counts(f, rlo, rhi, c0, lane, s4); incl = wave_scan_incl(s4, lane); }
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        for (; i < 64; ++i) {   // this wave's tiles whose FIRST sample falls into the rows of this round
            const int t = t0 + i * stride;
            if (t >= tend || 32 * t - run >= tot) break;
            const int gl = 32 * t - run;                            // offset of the tile's first sample inside this round
            const int ls = __builtin_ctzll(__ballot(gl < incl));   // the lane whose four rows hold it
            const int ex = __builtin_amdgcn_readlane(incl - s4, ls);
            const int nx = __builtin_amdgcn_readlane(n.x, ls), ny = __builtin_amdgcn_readlane(n.y, ls), nz = __builtin_amdgcn_readlane(n.z, ls);
            int r = (q0 + c0 + ls) << 2, rem = gl - ex;
            if (rem >= nx) { rem -= nx; ++r; if (rem >= ny) { rem -= ny; ++r; if (rem >= nz) { rem -= nz; ++r; } } }
            if (lane == i) vdesc = ((unsigned)r << 16) | ((unsigned)rem & 0xFFFFu);
        }
        if (t0 + i * stride >= tend) break;
        run += tot;
    }
    return i;
}

// the side record of sample slot k, lane half h, of the bundle straddling window boundary b
__device__ __forceinline__ float* flat_rec(const DevFrame& f, int b, int k, int h) {
    return f.side + ((size_t)((size_t)b * f.S_max + k) * 2 + h) * FLAT_REC;
}
// ---- the hand-off of a straddling bundle between its two waves (MI355X guide, "Workgroup dispatch ... inter-workgroup visibility") -----------
// Producer side, every lane that owns a sample of a straddling bundle: its 24-float record as six WRITE-THROUGH 16-byte stores (sc1: the
// bytes leave this XCD's L2, nothing to write back later), then the wave drains its stores (s_waitcnt vmcnt(0)) and ONE lane per boundary
// adds 1 to the boundary's counter (agent-scope returning atomic).  The wave whose add returns 1 is the last of the two: one agent-scope
// acquire (buffer_inv sc1), then it loads both parts' records with sc1 loads and composites the bundle.  Nobody spins, so no schedule of
// the two waves can deadlock; the counter is returned to 0 by the last arriver (k_prepare / k_plan zero it once per frame).
typedef float F4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1_16(float* p, float a, float b, float c, float d) {
    const F4s v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int load_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The bundle that straddles window boundary b, from the per-sample records both waves left: the whole wave, lane (h, i) = lane half x
// value index (0..19 the weighted values, 20 the weight, 21 weight x depth term), EXACTLY the in-wave composite's arithmetic:
// transmittance as the product over the earlier samples nearest first, weights alpha x T, the sums right-nested (S_max <= 4: the Horner
// chain) or by doubling (S_max > 4: the ds_bpermute steps).  NS = slots unrolled (registers, all records requested up front: one memory
// round trip behind the header's): 4 for S_max <= 4, 16 else.
template <int NS>
__device__ __forceinline__ void flat_fix_boundary(const FusedArgs& a, const DevFrame& f, int b, int lane) {
#pragma clang fp contract(off)
    const int h = lane >> 5, i = lane & 31, S = f.S_max;
    const float* r0 = flat_rec(f, b, 0, h);
    const int gb = load_sc1((const int*)r0 + 22), cnt = load_sc1((const int*)r0 + 23);   // the header rides in slot 0's record (written by the tail side)
    float val[NS], alv[NS], ztv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {   // every slot the config can have, unconditionally (slots past the count are unused)
        const float* r = r0 + (size_t)min(k, S - 1) * 2 * FLAT_REC;
        alv[k] = load_sc1(r + 20); ztv[k] = load_sc1(r + 21); val[k] = load_sc1(r + (i < 20 ? i : 0));
    }
    if (gb < 0 || cnt < 2 || cnt > S || cnt > NS || gb >= f.H * f.W) return;   // (cannot happen with records of this launch)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        float Tr = 1.f;
#pragma unroll
        for (int d = 1; d <= k; ++d) if (S > 4 ? d < S : d < 4) Tr *= 1.f - alv[k - d];
        const float w = alv[k] * Tr;
        val[k] = i < 20 ? w * val[k] : (i == 20 ? w : w * ztv[k]);
    }
    float acc;
    if (S <= 4) {
        acc = 0.f;
        bool started = false;
#pragma unroll
        for (int k = NS - 1; k >= 0; --k)
            if (k < cnt) { acc = started ? val[k] + acc : val[k]; started = true; }
    } else {
#pragma unroll
        for (int d = 1; d < NS; d <<= 1)
            if (d < S) {
#pragma unroll
                for (int k = 0; k + d < NS; ++k) if (k + d < cnt) val[k] = val[k] + val[k + d];   // ascending k reads the not yet updated k + d
            }
        acc = val[0];
    }
    const float wsum = __shfl(acc, (h << 5) | 20), rden = 1.f / fmaxf(wsum, 1e-6f);
    const size_t row = (size_t)(f.B > 1 ? a.row_lo / f.H : 0) * f.H * f.W + (size_t)gb;
    int ch = -1;
    if (i < 16) ch = own_chan(h, i);
    else if (i < 20) ch = NBLEND + 4 * h + (i - 16);
    if (ch >= 0) a.bf[row * a.ldo + ch] = acc * rden;
    if (h == 0 && i == 20) { if (a.ldo == NOUT) a.opac[row] = acc * rden; else a.bf[row * a.ldo + NOUT + 1] = acc * rden; }
    if (h == 0 && i == 21) {
        const float d = acc * rden, dd = f.inv_depth ? 1.f / d : d;
        if (a.ldo == NOUT) a.depth[row] = dd; else a.bf[row * a.ldo + NOUT] = dd;
    }
}

// The wave's arrival at its window's boundaries, settled: ticket 1 = it was the second of the boundary's two waves - the other part's
// records are in memory (their wave drained its write-through stores before its add) and the loads of flat_fix_boundary go past this
// CU's L1 (sc1); it composites the bundle and returns the counter to zero for the next render.  pend: bit 0 = arrived at boundary b
// (lane 0 holds that ticket), bit 1 = at boundary b + 1 (lane 1).  GDB_FLAT_ACQUIRE adds the agent-scope acquire of the guide's general
// recipe (buffer_inv sc1: every wave of the CU then refills its L1): +4.7 .. 11 us per launch (profiles/r06/
// ab_flat_record_128B_and_acquire.txt), so it is a build switch and the default rests on one 128-byte line per side record instead -
// DESIGN.md 4.2b, "What the hand-off's memory ordering rests on".
__device__ __forceinline__ void flat_settle(const FusedArgs& a, const DevFrame& f, int ticket, int pend, int b, int lane) {
    const bool last_h = (pend & 1) && __builtin_amdgcn_readlane(ticket, 0) == 1, last_t = (pend & 2) && __builtin_amdgcn_readlane(ticket, 1) == 1;
    if (!(last_h | last_t)) return;
#ifdef GDB_FLAT_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if ((lane == 0 && last_h) || (lane == 1 && last_t)) __hip_atomic_store(f.side_hdr + b + (lane & 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.S_max <= 4) {
        if (last_h) flat_fix_boundary<4>(a, f, b, lane);
        if (last_t) flat_fix_boundary<4>(a, f, b + 1, lane);
    } else {
        if (last_h) flat_fix_boundary<GDB_MAX_SAMPLES>(a, f, b, lane);
        if (last_t) flat_fix_boundary<GDB_MAX_SAMPLES>(a, f, b + 1, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ONE body for both list schedules (FLAT = false: the dense schedule's windows of whole bundles; true: windows of exactly 32
// consecutive samples).  k_render_dense / k_render_flat below are its two kernels.
// PERSIST: the wave walks several tiles (the launch is the resident grid).  false: the grid has one wave per tile of the worst case
// and a wave renders at most ONE tile - then nothing has to be kept out of the tile loop's way and the kernel arguments are plain
// loop-free SGPR values again.  GDB_SCHED_AUTO always walks (profiles/r04/ab_walk_vs_one_tile.txt); the one-tile form is what the
// split-f16 flat build at three waves per SIMD takes (no register left for the walk's loop state) and a diagnostic switch.
// BB = 4: bundle_size 2, everything in this launch.  BB = 1 (dense only; round 6): bundle_size 1 / 4 - the list kernel renders the bundle's
// CENTRE ray (everything of a sample but its 3 b^2 sub-ray colours: the MLP never sees those, nerf.py:98 slices them off) into packed
// rows of the b = 2 shape and leaves, per (sample, view), the weight its colours get in the bundle's output - normalised composite
// weight (utils.py:35-41) x softmax blend weight (nerf.py:108-110); k_bundle_colours (below) then gathers the b^2 colours per sample and
// view with those weights and assembles the (N_b, 3 b^2 + 27 [+ 2]) rows.
template <int PREC, int NWG, bool PERSIST, bool FLAT, int BB = 4>
__device__ __forceinline__ void render_list_body(const FusedArgs& a_) {
    static_assert(BB == 4 || (BB == 1 && !FLAT && PREC != GDB_PREC_F32X), "the centre-ray form exists for the dense schedule at fp32 / f16 operands");
    // Everything wave-uniform is re-derived inside each tile iteration from an opaque pointer to the kernel-argument segment (as in
    // k_render_solo): as loop invariants those values would be live across the whole body, which has no register to spare.
    typedef const FusedArgs __attribute__((address_space(4))) KArgs;
    KArgs* const ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    // ---- this wave's tiles: t0, t0 + stride, ... < tend, looked up once (dense_fill / flat_fill) ---------------------------------
    // Rows are addressed by their global index (batch item x H + row), so one launch covers the rows [row_lo, row_hi) of ALL batch
    // items (dense; the flat schedule launches per batch item), and nothing but the descriptor register and two counters lives across a tile.
    unsigned vdesc;
    int ntile, t0 = 0;
    {
        const FusedArgs& a = *(const FusedArgs*)ap;
        const int lane = threadIdx.x & 63, wv = NWG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
        // XCD-aware order: workgroups b, b + 8, ... share an XCD (and its L2); XCD x walks the band [x chunk, (x + 1) chunk) of the tiles
        if constexpr (FLAT) {
            ntile = __builtin_amdgcn_readfirstlane(flat_fill(a.f, a.row_lo, a.row_hi, lane, (int)(blockIdx.x & 7), (int)(blockIdx.x >> 3) * NWG + wv,
                                                             a.tile_stride, vdesc, t0));
            t0 = __builtin_amdgcn_readfirstlane(t0);
        } else {
            ntile = __builtin_amdgcn_readfirstlane(dense_fill(a.f, a.row_lo, a.row_hi, lane, (int)(blockIdx.x & 7), (int)(blockIdx.x >> 3) * NWG + wv,
                                                              a.tile_stride, vdesc));
        }
    }
    if (ntile <= 0) return;
#ifdef GDB_DIAG
    // Experiment (GDB_XP_STAGGER = n, diagnostic build; profiles/r06/xp_start_stagger.txt): the waves of a CU start their first gather
    // together - a burst on the memory system with the matrix pipe idle - and the oldest workgroups of a CU then finish 30 us before
    // the youngest.  n > 0: the workgroup of dispatch rank r on its CU (0 = oldest of R) sleeps (R - 1 - r) x n x 4,096
    // cycles before its first tile (the old ones, which have the slack, yield the start); n < 0: rank r sleeps r x |n| x 4,096.
    {
        const FusedArgs& a = *(const FusedArgs*)ap;
        if (a.xp_stagger && a.xp_per_rank > 0) {
            const int slot = (int)(blockIdx.x >> 3), R = max(((int)(gridDim.x >> 3) + a.xp_per_rank - 1) / a.xp_per_rank, 1), r = min(slot / a.xp_per_rank, R - 1);
            const int n = a.xp_stagger > 0 ? (R - 1 - r) * a.xp_stagger : r * -a.xp_stagger;
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }
#endif
    int it = 0;
    // flat: the arrival ticket of the tile just rendered and which of its two boundaries it arrived at (bit 0: the one before it, bit 1:
    // the one behind it; 0: nothing pending).  The ticket is READ at the top of the next tile (or behind the loop): by then the output
    // stores issued after the atomic have long completed, so the in-order vmcnt wait for it costs nothing.
    // (ONE register across tiles: lanes 0 / 1 hold the tickets, lane 2 the pending bits, lane 3 the index of the boundary before that tile)
    int ticket = 0;
    do {   // (one pass, and no loop at all for the compiler, when !PERSIST)
    wave_prio<PERSIST && !FLAT>(it + 1 >= ntile, true);   // (flat: two equal tiles per wave, left alone: +1.9 % with it; one-tile dense: 1 % slower with it)
    KArgs* apk = ap;
    if constexpr (PERSIST) asm volatile("" : "+s"(apk));  // nothing read through apk is loop-invariant to the compiler
    // The arguments as VALUES: the by-value kernel parameter when the wave renders one tile (the compiler fetches it in a few wide
    // scalar loads at kernel entry), a copy made at the top of every tile of the walk (one burst of scalar loads, through the opaque
    // pointer so that nothing of it lives across tiles).  Read field by field through the pointer instead, each use paid a scalar
    // load and its wait where it stood: +2.5 % on a tile (c3 fp32 180.5 -> 176.0 us, c4 211.7 -> 206.9: profiles/r04/ab_kernarg_access.txt).
    const FusedArgs a_copy = *(const FusedArgs*)apk;
    const FusedArgs& a = PERSIST ? a_copy : a_;
    const DevFrame& f = a.f;
    // ... and so is everything per-lane: nothing derived from the lane index may be hoisted out of the tile loop either
    const int tid = PERSIST ? opaque((int)threadIdx.x) : (int)threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = NWG > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    float* stage = (float*)smem4 + (size_t)wv * a.wave_floats;
#ifdef GDB_DEBUG_STAMPS  // one stamp record per (wave slot, tile iteration)
    unsigned* dbg = a.dbg ? a.dbg + (size_t)it * gridDim.x * (16 * 16 * 2) : nullptr;
#else
    unsigned* dbg = nullptr; (void)dbg;
#endif
    const unsigned desc = (unsigned)__builtin_amdgcn_readlane((int)vdesc, it);
    if (desc == 0xFFFFFFFFu) continue;
    typedef const int __attribute__((address_space(4))) kint;  // written by an earlier launch: scalar loads
    // ---- the window: its samples (lane j = sample j of it), its bundles [first, first + nb) as indices into the batch item's map --------
    // dense: tile index -> (row, window); flat: tile -> (first row, offset in that row) and the rows the 32 samples cover.
    // Everything read from the plan is clamped to the frame, so that a plan that does not belong to this frame's depth prior renders
    // garbage instead of reading or writing outside the frame.
    int bi, n, tile = 0, row_l = 0, my_row = 0;
    unsigned m = 0xFFFFFFFFu;
    bool have = true;
    if constexpr (FLAT) {
        tile = t0 + it * a.tile_stride;        // this tile's index in the launch = the boundary behind it
        const int r0 = (int)(desc >> 16), off0 = (int)(desc & 0xFFFFu);
        if (r0 < a.row_lo || r0 >= a.row_hi) continue;
        bi = __builtin_amdgcn_readfirstlane(f.B > 1 ? r0 / f.H : 0);   // (one launch per batch item: every row of the window is its)
        const kint* kns = (const kint*)f.nsamp;
        int cov = 0, my_ent = 0;
        my_row = r0; have = false;
        {
            int rr = r0, eoff = off0;
            while (cov < 32 && rr < a.row_hi) {   // (one or two rows in all but tiny maps)
                const int ns = min(max(kns[rr], 0), f.smapStride - 32);
                const int take = min(max(ns - eoff, 0), 32 - cov);
                if (j >= cov && j < cov + take) { my_row = rr; my_ent = eoff + (j - cov); have = true; }
                cov += take; ++rr; eoff = 0;
            }
        }
        n = cov;
        if (n <= 0) continue;
        if (have) m = ldu<unsigned>(f.smap, 4u * (unsigned)(my_row * f.smapStride + my_ent));
    } else {
        const int rowid = (int)(desc >> 16), win = (int)(desc & 0xFFFFu);
        if (rowid < a.row_lo || rowid >= a.row_hi) continue;
        bi = __builtin_amdgcn_readfirstlane(f.B > 1 ? rowid / f.H : 0); row_l = rowid - bi * f.H;
        const kint* krec = (const kint*)(f.plan + (size_t)rowid * (f.planMW + 2));
        if (win >= krec[0]) continue;  // (cannot happen with a plan that belongs to this depth prior)
        // This window = the row's sample offsets [s0, s0 + n), n <= 32, whole bundles (plan_row).
        const int s0 = min(max(krec[1 + win], 0), f.smapStride - 32);
        n = min(max(krec[2 + win] - s0, 0), 32);
        if (n <= 0) continue;
        m = ldu<unsigned>(f.smap + (size_t)rowid * f.smapStride, 4u * (unsigned)(s0 + j));  // lane j = sample s0 + j of the row
    }
    // flat: settle the previous tile's arrivals here - its output stores have had the window lookup above to complete (the in-order
    // wait for the ticket waits for them), and the target camera's 24 scalars are not live yet
    if constexpr (FLAT && PERSIST) {
        const int pend = __builtin_amdgcn_readlane(ticket, 2);
        if (pend) { flat_settle(a, f, ticket, pend, __builtin_amdgcn_readlane(ticket, 3), lane); ticket = 0; }
    }
    float tc[TAR_STRIDE];
    {
        const kfloat* tcg = kptr(tar_cam(f, bi));
        if constexpr (PERSIST) asm volatile("" : "+s"(tcg));
#pragma unroll
        for (int i = 0; i < TAR_STRIDE; ++i) tc[i] = tcg[i];
    }
    STAMP(0);
    int first, nb, hp = 0, tp = 0, mx;
    const int k_g = (int)((m >> 16) & 0xFFu), mcnt = (int)(m >> 24);
    bool act;
    if constexpr (FLAT) {
        mx = min((int)(m & 0xFFFFu), f.W - 1);
        row_l = my_row - bi * f.H;                       // this lane's bundle-map row inside its batch item
        const int gbl = row_l * f.W + mx;                // ... and its bundle index inside the batch item's map (< 2^24, checked by the launcher)
        // the window's bundles: lane 0's .. the last lane's; bundles hold >= 1 sample each, so they are consecutive map indices, <= 32
        first = __builtin_amdgcn_readfirstlane(gbl);
        nb = min(max(__builtin_amdgcn_readlane(gbl, n - 1) - first + 1, 0), 32);
        if (nb <= 0 || first < 0 || first + nb > f.H * f.W) continue;
        act = have && m != 0xFFFFFFFFu && gbl - first >= 0 && gbl - first < nb && k_g < mcnt && mcnt >= 1 && mcnt <= f.S_max;
        // does the window begin / end inside a bundle?  (its first sample is not slot 0 / its last sample is not the bundle's last)
        hp = __builtin_amdgcn_readfirstlane(k_g) > 0 ? 1 : 0;
        tp = (__builtin_amdgcn_readlane(k_g, n - 1) + 1 < __builtin_amdgcn_readlane(mcnt, n - 1)) ? 1 : 0;
        // what has to survive gather + MLP in ONE register: map index (24 bits) | slot (4) | count - 1 (4)
        m = (unsigned)gbl | ((unsigned)min(k_g, 15) << 24) | ((unsigned)(min(max(mcnt, 1), 16) - 1) << 28);
    } else {
        mx = (int)(m & 0xFFFFu);
        // the window's bundles: first (lane 0's: a window starts at a bundle's first sample) .. the last lane's, nb <= 32
        first = min(max(__builtin_amdgcn_readfirstlane(mx), 0), f.W - 1);
        nb = min(max(__builtin_amdgcn_readlane(mx, n - 1) - first + 1, 0), min(32, f.W - first));
        if (nb <= 0) continue;
        const int bj_g = min(max(mx - first, 0), nb - 1);        // this sample's bundle inside the window (= its output column)
        act = j < n && m != 0xFFFFFFFFu && mx - first >= 0 && mx - first < nb && j - k_g >= 0 && k_g < mcnt && mcnt <= f.S_max;
        mx = first + bj_g;
    }
    const float* mfg = a.pw + PW_FP32_FLOATS + (PREC == GDB_PREC_F32 ? MFMA_FLOATS : 0);
    const float b_agg = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BAGG : TS_BAGG], b_w2 = kptr(mfg)[PREC == GDB_PREC_F32 ? S32_BW2 : TS_BW2];
    float vox[4];
    {
        float z_g;
        Bundle<BB> q;
        if constexpr (BB == 4) load_bundle<4, true, false>(f, tc, bi, row_l, mx, q);
        else load_bundle_center<false>(f, tc, bi, row_l, mx, q);
        q.count = min(max(mcnt, 1), f.S_max);  // the plan's count (bundle_sampler.py:179 evaluated by plan_row): no second IEEE division per lane
        STAMP(1);
        slot_gather_q<PREC, BB>(f, stage, tc, q, min(k_g, q.count - 1), bi, j, h, a.skip, act, z_g, vox);
    }
    STAMP(2);
    wave_prio<PERSIST && !FLAT>(it + 1 >= ntile, false);
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    float v[22];  // 0..15 blended channels own_chan(h, i), 16..19 feat_head 4h.., 20 weight, 21 weight x depth
    float sig;
    {
        float bacc[16], fhv[4];
        if (PREC == GDB_PREC_F32) slot_mlp_core_f32(f, mfg, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
        else slot_mlp_core<PREC == GDB_PREC_F32X>(f, mfg, stage, vox, lane, j, h, b_agg, b_w2, dbg, bacc, fhv, sig);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = bacc[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[16 + i] = fhv[i];
    }
    // ---- composite across the lanes of a bundle --------------------------------------------------------------------------
    // The sample's slot, count and output column are decoded AGAIN from its list entry (one register across gather + MLP instead
    // of four: the split-f16 build has none to spare).
    const unsigned m_c = (unsigned)opaque((int)m);
    int k, cnt, bj, gb;   // slot, count, the sample's bundle inside the window (= its output column), its index in the batch item's map
    if constexpr (FLAT) {
        gb = (int)(m_c & 0xFFFFFFu); k = (int)((m_c >> 24) & 0xFu); cnt = min((int)(m_c >> 28) + 1, f.S_max);
        bj = min(max(gb - first, 0), nb - 1);
    } else {
        k = (int)((m_c >> 16) & 0xFFu); cnt = min(max((int)(m_c >> 24), 1), f.S_max);
        bj = min(max((int)(m_c & 0xFFFFu) - first, 0), nb - 1);
        gb = row_l * f.W + first + bj;
    }
    float z;  // the sample's depth, derived again from the depth prior (two loads the gather has left in L1 / L2) as bundle_sample does
    {
        const size_t hw = (size_t)f.H * f.W;
        const unsigned pz = 4u * (unsigned)gb;
        float n0 = ldu<float>(f.depth_range + ((size_t)bi * 2) * hw, pz), f0 = ldu<float>(f.depth_range + ((size_t)bi * 2 + 1) * hw, pz);
        if (f.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; }
        z = sample_mid<true>(n0, f0, cnt, min(k, cnt - 1));
        if (f.inv_depth) z = gdiv<true>(1.f, z);
    }
    const float al = act ? alpha_of(__shfl(sig, j)) : 0.f;  // sigma sits in half 0
    const float zt = f.inv_depth ? 1.f / z : z;
    if constexpr (FLAT) {
        // a straddling bundle's samples go to the side records of their boundary (the one before this tile for the window's first
        // bundle, the one behind it for its last): whichever wave arrives last at the boundary composites them (below)
        const bool in_head = hp && bj == 0, in_tail = tp && bj == nb - 1;
        if (act && (in_head || in_tail)) {
            float* r = flat_rec(f, a.flat_base + tile + (in_head ? 0 : 1), k, h);
#pragma unroll
            for (int i = 0; i < 5; ++i) store_sc1_16(r + 4 * i, v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            // (the boundary's header - map index, sample count - rides in every record; the fix-up reads slot 0's, a tail-side sample)
            store_sc1_16(r + 20, al, zt, __builtin_bit_cast(float, gb), __builtin_bit_cast(float, cnt));
        }
    }
    const int S = f.S_max;
    float w_own = 0.f;   // this sample's transmittance weight alpha x T (utils.py:35), before the bundle's normalisation
    if (S <= 4) {
        // An active sample's earlier samples are the lanes just below it in the same half (k <= j), a bundle's later samples the
        // lanes just above (they end at lane 31 at the latest): whole-wave DPP shifts never carry a value across a bundle's edge
        // that the predicates below do not mask.
        float Tr = 1.f, ap_ = al;
#pragma unroll
        for (int d = 1; d < 4; ++d) {  // transmittance: product over the bundle's earlier samples (lanes j-1 .. j-k)
            ap_ = wave_shr1(ap_);
            if (d <= k) Tr *= 1.f - ap_;
        }
        const float w = al * Tr;
        w_own = w;
#pragma unroll
        for (int i = 0; i < 20; ++i) v[i] = act ? w * v[i] : 0.f;  // lanes without a sample hold unspecified MLP outputs
        v[20] = w;
        v[21] = act ? w * zt : 0.f;
        // Segmented suffix sums as a Horner chain: acc <- v + (the bundle has a next sample ? acc of the next lane : 0); after
        // S - 1 steps the bundle's first lane (k == 0) holds the bundle's sums.  The neighbour is masked with a SELECT, not a
        // multiply by 0: a non-finite sum of the next bundle must stay in its own bundle (as in the other schedules and the reference).
        // (The select sits on the SOURCE lane - "I continue the bundle of the lane below" - so that the shift folds into the add:
        // v_cndmask + v_add_f32_dpp, two instructions per value and step; profiles/r04/ab_single_changes_f32.txt has the earlier forms.)
        // (flat: lane 32 = sample 0 of the other half is never a continuation of lane 31's bundle, although its slot is > 0 when the
        // window begins inside a bundle)
        const bool is_cont = FLAT ? (act && k > 0 && j > 0) : (act && k > 0);   // this lane continues the bundle of the lane below it
        float acc[22];
#pragma unroll
        for (int i = 0; i < 22; ++i) acc[i] = v[i];
        for (int d = 1; d < S; ++d) {
#pragma unroll
            // mask at the SOURCE lane (a continuation lane: active, slot > 0), then one DPP add per value: acc <- v + shl(masked acc)
            for (int i = 0; i < 22; ++i) { const float mk = is_cont ? acc[i] : 0.f; acc[i] = v[i] + wave_shl1(mk); }
        }
#pragma unroll
        for (int i = 0; i < 22; ++i) v[i] = acc[i];
    } else {
        float Tr = 1.f;
        for (int d = 1; d < S; ++d) {  // transmittance: product over the bundle's earlier samples (lanes j-1 .. j-k)
            const float ap_ = __shfl_up(al, d, 32);
            if (d <= k) Tr *= 1.f - ap_;
        }
        const float w = al * Tr;
        w_own = w;
#pragma unroll
        for (int i = 0; i < 20; ++i) v[i] = act ? w * v[i] : 0.f;
        v[20] = w;
        v[21] = act ? w * zt : 0.f;
        for (int d = 1; d < S; d <<= 1) {  // segmented suffix sums by doubling (ds_bpermute): the bundle's first lane ends with its sums
            const bool take = act && k + d < cnt;
#pragma unroll
            for (int i = 0; i < 22; ++i) {
                const float tt = __shfl_down(v[i], d, 32);
                if (take) v[i] += tt;
            }
        }
    }
    if constexpr (BB != 4) {
        // ---- bundle_size 1 / 4: the weight this sample's colours of view v get in the bundle's output (k_bundle_colours) ----------------
        // = w / max(sum over the bundle, 1e-6) (utils.py:38-41; the bundle's first lane holds the sum) x exp(logit_v - max) / sum_v (nerf.py:
        // 109-110: the logits lie in the first direction row of each staged view, where the blend pass left them; the same exp / reciprocal
        // the pass itself took).  Written by lane half 0 of every lane that carries a sample; slots past a bundle's count are never read.
        constexpr int RD = row_dir<PREC>(), SV = stage_v<PREC>();
        const float wsum = __shfl(v[20], (h << 5) | max(j - k, 0));
        const float wn = w_own * (1.f / fmaxf(wsum, 1e-6f));
        float mxl = -INFINITY, den = 0.f;
        for (int vv = 0; vv < f.V; ++vv) mxl = fmaxf(mxl, stage[(size_t)vv * SV + RD * 32 + j]);
        for (int vv = 0; vv < f.V; ++vv) den += __expf(stage[(size_t)vv * SV + RD * 32 + j] - mxl);
        const float rdn = frcp(den);
        if (act && h == 0) {
            float* wp = a.wv + (((size_t)bi * f.H * f.W + (size_t)gb) * f.S_max + k) * f.V;
            for (int vv = 0; vv < f.V; ++vv) wp[vv] = wn * (__expf(stage[(size_t)vv * SV + RD * 32 + j] - mxl) * rdn);
        }
    }
    STAMP(7);
    // ---- flat: arrive at the boundaries this window shares with its neighbours (see "the hand-off" above) ----------------------------
    // The wave's side records (issued before the composite above) have left the CU before its arrival is counted: the stores are
    // inline asm, the compiler's own wait counting does not see them.  The returning atomic is only REQUESTED here; its value is read
    // behind the output copy, whose stores do not delay it (loads / atomics return in issue order, and the copy is issued later).
    if constexpr (FLAT) {
        if (hp | tp) {   // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int* cp = f.side_hdr + a.flat_base + tile + (lane & 1);   // lane 0: the boundary before this tile, lane 1: the one behind it
            ticket = lane == 2 ? (hp | (tp << 1)) : (lane == 3 ? a.flat_base + tile : 0);
            if ((lane == 0 && hp) || (lane == 1 && tp)) ticket = __hip_atomic_fetch_add(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    // the window's WHOLE bundles (columns hp .. nb - 1 - tp; dense: all of them) to the output record, column 0 = the first whole one
    float* o = stage;  // the output record: [bundles of the window][ld] (+ depth, opacity behind it in the three-tensor form)
    const int ld = a.ldo, col = bj - hp, ncol = nb - hp - tp;
    if (act && k == 0 && (!FLAT || (col >= 0 && col < ncol))) {
        const float rden = 1.f / fmaxf(v[20], 1e-6f);
        out_store_own16(o, ld, col, h, [&](int i) { return v[i] * rden; });
#pragma unroll
        for (int i = 0; i < 4; ++i) o[col * ld + NBLEND + 4 * h + i] = v[16 + i] * rden;
        if (h == 0) {
            const float d = v[21] * rden;
            *out_depth_slot(o, ld, col) = f.inv_depth ? 1.f / d : d;  // network.py:88-89
            *out_opac_slot(o, ld, col) = v[20] * rden;
        }
    }
    __builtin_amdgcn_wave_barrier();
    PHASE_FENCE();
    if constexpr (FLAT) { if (ncol > 0) out_copy(a, o, (size_t)bi * f.H * f.W + (size_t)(first + hp), ncol, lane); }
    else out_copy(a, o, ((size_t)bi * f.H + row_l) * f.W + (size_t)first, nb, lane);
    STAMP(8);
    STAMP(9);
    __builtin_amdgcn_wave_barrier();  // the next tile's gather overwrites the area the stores above read
    PHASE_FENCE();
    } while (PERSIST && ++it < ntile);
    if constexpr (FLAT) {   // the last tile's arrival (every tile's, when the wave renders one)
        const int pend = __builtin_amdgcn_readlane(ticket, 2);
        if (pend) {
            const FusedArgs& a = *(const FusedArgs*)ap;
            flat_settle(a, a.f, ticket, pend, __builtin_amdgcn_readlane(ticket, 3), (int)(threadIdx.x & 63));
        }
    }
}

template <int PREC, int WPS, int NWG, bool PERSIST, int BB = 4>
__global__ void __launch_bounds__(64 * NWG, WPS) k_render_dense(FusedArgs a_) { render_list_body<PREC, NWG, PERSIST, false, BB>(a_); }
template <int PREC, int WPS, int NWG, bool PERSIST>
__global__ void __launch_bounds__(64 * NWG, WPS) k_render_flat(FusedArgs a_) { render_list_body<PREC, NWG, PERSIST, true>(a_); }


// LDS above 64 KB per workgroup needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a per-device property of the
// function: the once-flag is a bit per device ordinal (relaxed atomics; setting it twice is harmless).
#include <atomic>
template <class K>
static hipError_t allow_big_lds(K kernel, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_relaxed) & bit)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev < 64) done.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

template <int PREC, int WPS>
static hipError_t launch_solo(const FusedArgs& a, unsigned grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    hipError_t e = allow_big_lds(k_render_solo<PREC, WPS>, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_render_solo<PREC, WPS>), dim3(grid), dim3(64), lds, st, a);
    return hipGetLastError();
}

// lds: bytes per wave (a multiple of 16).  NWG waves per workgroup: whichever of 1, 2, 4 puts the most waves on a CU (LDS comes in
// 1280-byte granules, so several waves' areas in one allocation can fit where single ones lose a wave to the rounding); on a tie
// the smaller workgroup, unless GDB_DENSE_PREFER_WIDE asks for the larger one (fewer workgroups for the dispatcher to launch).
#ifndef GDB_DENSE_PREFER_WIDE
#define GDB_DENSE_PREFER_WIDE 0
#endif
// Workgroups of this kernel a CU holds at once x the CUs of the device: the persistent grid.  Asked of the runtime once per
// (kernel instantiation, device, LDS size) and kept in one atomic word per device ordinal (relaxed; two threads racing just ask twice).
template <class K>
static hipError_t resident_workgroups(K kernel, int threads, size_t lds, std::atomic<unsigned long long>* cache, int& per_cu, int& cus) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long key = (unsigned long long)lds & 0x3FFFFull;
    if (dev < 64) {
        const unsigned long long c = cache[dev].load(std::memory_order_relaxed);
        if ((c >> 63) && (c & 0x3FFFFull) == key) { per_cu = (int)((c >> 18) & 0xFF); cus = (int)((c >> 26) & 0xFFF); return hipSuccess; }
    }
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, threads, lds);
    if (e != hipSuccess) return e;
    // the LDS bound as the hardware allocates it (1280-byte granules): never above it, whatever the API says
    const int by_lds = (int)((size_t)(160 * 1024) / ((lds + 1279) / 1280 * 1280));
    per_cu = per_cu < 1 ? 1 : (per_cu > by_lds ? (by_lds < 1 ? 1 : by_lds) : per_cu);
    if (per_cu > 255) per_cu = 255;
    if (cus < 1) cus = 1;
    if (cus > 4095) cus = 4095;
    if (dev < 64) cache[dev].store((1ull << 63) | ((unsigned long long)cus << 26) | ((unsigned long long)per_cu << 18) | key, std::memory_order_relaxed);
    return hipSuccess;
}

template <int PREC, int WPS, int NWG, int BB = 4>
static hipError_t launch_dense_n(FusedArgs& a, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0}, done1{0};
    static std::atomic<unsigned long long> resident[64];
    hipError_t e = allow_big_lds(k_render_dense<PREC, WPS, NWG, true, BB>, done);
#ifdef GDB_DIAG   // (the one-tile-per-wave form exists in the diagnostic build only: GDB_SCHED_AUTO always walks)
    if constexpr (BB == 4) if (e == hipSuccess) e = allow_big_lds(k_render_dense<PREC, WPS, NWG, false>, done1);
#endif
    (void)done1;
    if (e != hipSuccess) return e;
    int per_cu = 1, cus = 1;
    e = resident_workgroups(k_render_dense<PREC, WPS, NWG, true, BB>, 64 * NWG, NWG * lds, resident, per_cu, cus);
    if (e != hipSuccess) return e;
#ifdef GDB_DIAG  // diagnostic build: over- / under-subscribe the persistent grid (workgroups per CU) from the environment
    static const int env_wgs = getenv("GDB_DENSE_WGS_PER_CU") ? atoi(getenv("GDB_DENSE_WGS_PER_CU")) : 0;
    if (env_wgs > 0) per_cu = env_wgs;
#endif
    // The resident grid, shared between the batch items (grid y), never more workgroups than the worst case has tiles
    // (a.ntiles: every bundle at S_max), a multiple of 8 (one octet = one workgroup per XCD).
    long long grid = (long long)per_cu * cus;
    grid = grid >= 8 ? grid / 8 * 8 : 8;
    // One launch renders the rows [row_lo, row_hi) of the global row index (batch item x H + row): all batch items at once when
    // the strip is the whole frame (or B = 1), one launch per batch item otherwise.
    const bool whole = a.f.B == 1 || (a.row_begin == 0 && a.nrows == a.f.H);
    const int nl = whole ? 1 : a.f.B;
    for (int l = 0; l < nl; ++l) {
        a.row_lo = (whole ? 0 : l * a.f.H) + a.row_begin;
        a.row_hi = whole ? (a.f.B - 1) * a.f.H + a.row_begin + a.nrows : a.row_lo + a.nrows;
        // never more workgroups than the worst case has tiles (every bundle at S_max)
        const long long tiles = (long long)(a.row_hi - a.row_lo) * a.f.planMW;
        const long long worst = ((tiles + NWG - 1) / NWG + 7) / 8 * 8;
        long long g = grid > worst ? worst : grid;
        // a wave keeps its tiles' descriptors in the 64 lanes of one register: at most 64 tiles per wave (8 XCD bands of ceil(tiles / 8))
        const long long need = 8 * (((tiles + 7) / 8 + 64LL * NWG - 1) / (64LL * NWG));
        if (g < need) g = need;
        // Tile walk or one tile per wave: both forms are exact, the choice is speed only.  Measured on one MI355X, same process order
        // (profiles/r04/ab_walk_vs_one_tile.txt; kernel us, walk / one tile per wave): c2 fp32 99.8 / 102.9, f16 50.8 / 52.4; c4 (S_max
        // 6) fp32 202.9 / 206.0, f16 105.2 / 112.2, split-f16 140.0 / 146.3; but c3 fp32 181.6 / 175.7, f16 94.6 / 92.3; c3' fp32
        // 216.6 / 209.8.  The walk saves the dispatcher's gap between two tiles of a wave slot; the dispatcher's dynamic order balances
        // the long S_max 3 frames better than a static stride.  Rule fitted to those rows: the walk while the worst case is at most
        // three tiles per resident wave, or when S_max > 4.
        // (Round 4, with the walk's wave priorities in place - profiles/r04/ab_walk_vs_one_tile.txt, second table: the walk now wins on the
        // long S_max 3 frames too: c3 fp32 174.9 / 176.2, c3' 210.4 / 211.0, c3 f16 87.8 / 93.8, split-f16 115.7 / 120.5 - always the walk.)
        bool persist = true;
#ifdef GDB_DIAG
        static const int env_persist = getenv("GDB_DENSE_PERSIST") ? atoi(getenv("GDB_DENSE_PERSIST")) : -1;
        if (env_persist >= 0 && BB == 4) persist = env_persist != 0;
#endif
        if (!persist) g = worst;
        a.tile_stride = (int)(g >> 3) * NWG;
#ifdef GDB_DIAG
        if (!persist && BB == 4) hipLaunchKernelGGL((k_render_dense<PREC, WPS, NWG, false>), dim3((unsigned)g), dim3(64 * NWG), NWG * lds, st, a);
        else
#endif
        hipLaunchKernelGGL((k_render_dense<PREC, WPS, NWG, true, BB>), dim3((unsigned)g), dim3(64 * NWG), NWG * lds, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}
template <int PREC, int WPS>
static hipError_t launch_dense(FusedArgs& a, size_t lds, hipStream_t st) {
    const size_t gran = 1280, cap = 160 * 1024;
    auto waves = [&](size_t n) { return n * lds > cap ? (size_t)0 : n * (cap / ((n * lds + gran - 1) / gran * gran)); };
    const size_t w1 = waves(1), w2 = waves(2), w4 = waves(4);
    a.wave_floats = (int)(lds / sizeof(float));
    const bool wide = GDB_DENSE_PREFER_WIDE != 0;
    if ((w4 > w2 && w4 > w1) || (wide && w4 >= w2 && w4 >= w1 && w4 > 0)) return launch_dense_n<PREC, WPS, 4>(a, lds, st);
    if (w2 > w1 || (wide && w2 >= w1 && w2 > 0)) return launch_dense_n<PREC, WPS, 2>(a, lds, st);
    return launch_dense_n<PREC, WPS, 1>(a, lds, st);
}

// Flat schedule: one launch of k_render_flat per batch item (a window never crosses batch items); the bundles that straddle a window
// boundary are composited inside the same launch by whichever of their two waves arrives last (flat_fix_boundary).  Grid as launch_dense_n.
template <int PREC, int WPS, int NWG>
static hipError_t launch_flat_n(FusedArgs& a, size_t lds, int max_tiles_per_item, hipStream_t st) {
    // (the split-f16 build at three waves per SIMD has no register left for the walk's loop state: one tile per wave there)
    constexpr bool CAN_WALK = !(PREC == GDB_PREC_F32X && WPS == 3);
    static std::atomic<unsigned long long> done{0}, done1{0};
    static std::atomic<unsigned long long> resident[64];
    hipError_t e = allow_big_lds(k_render_flat<PREC, WPS, NWG, CAN_WALK>, done);
#ifdef GDB_DIAG
    if (e == hipSuccess) e = allow_big_lds(k_render_flat<PREC, WPS, NWG, false>, done1);
#endif
    (void)done1;
    if (e != hipSuccess) return e;
    int per_cu = 1, cus = 1;
    e = resident_workgroups(k_render_flat<PREC, WPS, NWG, CAN_WALK>, 64 * NWG, NWG * lds, resident, per_cu, cus);
    if (e != hipSuccess) return e;
#ifdef GDB_DIAG
    static const int env_wgs = getenv("GDB_DENSE_WGS_PER_CU") ? atoi(getenv("GDB_DENSE_WGS_PER_CU")) : 0;
    if (env_wgs > 0) per_cu = env_wgs;
#endif
    long long grid = (long long)per_cu * cus;
    grid = grid >= 8 ? grid / 8 * 8 : 8;
    for (int bi = 0; bi < a.f.B; ++bi) {
        a.row_lo = bi * a.f.H + a.row_begin;
        a.row_hi = a.row_lo + a.nrows;
        a.flat_base = bi * (max_tiles_per_item + 1);
        const long long tiles = ((long long)a.nrows * a.f.W * a.f.S_max + 31) / 32 + 1;   // worst case: every bundle at S_max
        const long long worst = ((tiles + NWG - 1) / NWG + 7) / 8 * 8;
        long long g = grid > worst ? worst : grid;
        const long long need = 8 * (((tiles + 7) / 8 + 64LL * NWG - 1) / (64LL * NWG));   // at most 64 tiles per wave (one descriptor per lane)
        if (g < need) g = need;
        bool persist = CAN_WALK;   // (always the walk where the build can, as launch_dense_n)
#ifdef GDB_DIAG
        static const int env_persist = getenv("GDB_DENSE_PERSIST") ? atoi(getenv("GDB_DENSE_PERSIST")) : -1;
        if (env_persist >= 0) persist = CAN_WALK && env_persist != 0;
#endif
        if (!persist) g = worst;
        a.tile_stride = (int)(g >> 3) * NWG;
#ifdef GDB_DIAG
        if (!persist && CAN_WALK) hipLaunchKernelGGL((k_render_flat<PREC, WPS, NWG, false>), dim3((unsigned)g), dim3(64 * NWG), NWG * lds, st, a);
        else
#endif
        hipLaunchKernelGGL((k_render_flat<PREC, WPS, NWG, CAN_WALK>), dim3((unsigned)g), dim3(64 * NWG), NWG * lds, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
template <int PREC, int WPS>
static hipError_t launch_flat(FusedArgs& a, size_t lds, int max_tiles_per_item, hipStream_t st) {
    const size_t gran = 1280, cap = 160 * 1024;
    auto waves = [&](size_t n) { return n * lds > cap ? (size_t)0 : n * (cap / ((n * lds + gran - 1) / gran * gran)); };
    const size_t w1 = waves(1), w2 = waves(2), w4 = waves(4);
    a.wave_floats = (int)(lds / sizeof(float));
    if (w4 > w2 && w4 > w1) return launch_flat_n<PREC, WPS, 4>(a, lds, max_tiles_per_item, st);
    if (w2 > w1) return launch_flat_n<PREC, WPS, 2>(a, lds, max_tiles_per_item, st);
    return launch_flat_n<PREC, WPS, 1>(a, lds, max_tiles_per_item, st);
}

template <bool LOOP, int WAVES, int PREC>
static hipError_t launch_fused(const FusedArgs& a, unsigned grid, int nw, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    hipError_t e = allow_big_lds(k_render_fused<LOOP, WAVES, PREC>, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_render_fused<LOOP, WAVES, PREC>), dim3(grid), dim3(64 * nw), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// bundle_size 1 / 4 (configs/dtu_pretrain.yaml:33 "bundle_size: 2  # 4 for 4*4"; network.py:31-34), second launch of the fused path: the
// b^2 sub-ray colours of every bundle.  out[c b^2 + s] = sum over the bundle's samples k and the views v of wv[k][v] x the bilinear
// colour c of source view v at the projection of sub-ray s's point o + d_s z_k - wv = normalised composite weight x softmax blend
// weight, left by k_render_dense<.., BB = 1> - i.e. utils.py:109-119 applied to nerf.py:110's blend, the two sums exchanged.  One thread
// per (bundle, sub-ray), the reference's own arithmetic per colour tap (world -> camera -> image as two products, IEEE divisions,
// F.grid_sample border / align_corners=False: bundle_sampler.py:327-337, as the operator mirror k_encode_views).  The threads of a
// bundle also move the bundle's other 29 values from the list kernel's packed rows (ld 41: [12 unused | feat (+) rgb 19 | feat_head 8 |
// depth | opacity]) into the output rows (N_b, 3 b^2 + 27) + depth + opacity, or the packed (N_b, 3 b^2 + 29) form.
struct ColArgs {
    DevFrame f;
    const float* wv; const float* tmp;
    float* bf; float* depth; float* opac;   // depth / opac NULL: packed rows
    int ldo;                                // floats per output row: 3 b^2 + 27, or + 2 packed
    int row_begin, nrows;
};
template <int BB>
__global__ void __launch_bounds__(256) k_bundle_colours(ColArgs a) {
    const DevFrame& f = a.f;
    constexpr int b = BB == 1 ? 1 : (BB == 4 ? 2 : 4);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nbs = (long long)f.B * a.nrows * f.W;     // bundles of the strip, all batch items
    const long long bnd = t / BB;
    const int s = (int)(t % BB);
    if (bnd >= nbs) return;
    const int x = (int)(bnd % f.W), rr = (int)(bnd / f.W), row = a.row_begin + rr % a.nrows, bi = rr / a.nrows;
    const size_t gb = ((size_t)bi * f.H + row) * f.W + x;     // bundle index over the whole batch
    const float* tc = tar_cam(f, bi);
    float rng[4];
    load_ranges(f, bi, row, x, rng);
    float n0 = rng[0], f0 = rng[1];
    if (f.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; }
    const int cnt = sample_count(n0, f0, tc[T_MINIV], f.S_max, f.adaptive);
    const float px = (float)(x * b + s % b) + 0.5f, py = (float)(row * b + s / b) + 0.5f;   // sub-ray order by * b + bx   bundle_sampler.py:100
    float d[3];
    ray_dir(tc + T_M, px, py, d);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < cnt; ++k) {
        float z = sample_mid<false>(n0, f0, cnt, k);
        if (f.inv_depth) z = 1.f / z;
        const float p[3] = {tc[T_O] + d[0] * z, tc[T_O + 1] + d[1] * z, tc[T_O + 2] + d[2] * z};   // :255
        const float* wp = a.wv + ((size_t)gb * f.S_max + k) * f.V;
        for (int v = 0; v < f.V; ++v) {
            const float* sc = src_cam(f, bi, v);
            float cam[3], im[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) cam[r] = sc[S_E + 4 * r] * p[0] + sc[S_E + 4 * r + 1] * p[1] + sc[S_E + 4 * r + 2] * p[2] + sc[S_E + 4 * r + 3];
#pragma unroll
            for (int r = 0; r < 3; ++r) im[r] = sc[S_K + 3 * r] * cam[0] + sc[S_K + 3 * r + 1] * cam[1] + sc[S_K + 3 * r + 2] * cam[2];
            const float zc = fmaxf(im[2], 1e-6f);
            const float gx = 2.f * (im[0] / zc) / (float)f.Wo - 1.f, gy = 2.f * (im[1] / zc) / (float)f.Ho - 1.f;
            float rgb[3];
            rgb_fetch(f.src_images + ((size_t)bi * f.V + v) * 3 * f.Ho * f.Wo, f.Ho, f.Wo, gx, gy, rgb);
            const float w = wp[v];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, rgb[c], acc[c]);
        }
    }
    const int Q = 3 * BB + GDB_CFR + GDB_CV;
    float* o = a.bf + gb * (size_t)a.ldo;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * BB + s] = acc[c];
    const float* ti = a.tmp + gb * (size_t)(NOUT + 2) + 12;   // [feat (+) rgb 19 | feat_head 8 | depth | opacity] of the list kernel's row
    for (int i = s; i < GDB_CFR + GDB_CV + 2; i += BB) {
        const float val = ti[i];
        if (i < GDB_CFR + GDB_CV) o[3 * BB + i] = val;
        else if (a.ldo != Q) o[Q + (i - GDB_CFR - GDB_CV)] = val;
        else if (i == GDB_CFR + GDB_CV) a.depth[gb] = val;
        else a.opac[gb] = val;
    }
}

static size_t solo_lds_bytes(int prec, int V);
static bool dense_fits(const GdbFrame& fr);
// The two launches of a bundle_size 1 / 4 render: the list kernel on the bundles' centre rays (dense schedule, two waves per workgroup,
// the walk), then the colours.  f32x operand pairs are not built in this form: that precision runs the fp32 kernel (which is the more
// exact of the two).
template <int PREC, int WPS>
static int render_center_launch(FusedArgs& a, const GdbConfig* cfg, const GdbFrame* fr, const WsLayout& L, const void* ws, bool plan_ready,
                                float* bf, float* depth, float* opac, int ldo, hipStream_t st) {
    if (!dense_fits(*fr)) return gdb_fail(GDB_E_SHAPE, "the dense schedule lists bundles and rows in 16 bits: W = %d, B x H = %lld (both must be < 65536)", fr->W, (long long)fr->B * fr->H);
    if (!plan_ready) {
        int rc = gdb_build_dense_plan(cfg, fr, const_cast<void*>(ws), st);
        if (rc) return rc;
    }
    a.alias = 0;
    a.bf = (float*)((char*)const_cast<void*>(ws) + L.colTmpOff); a.depth = nullptr; a.opac = nullptr; a.ldo = NOUT + 2;
    a.wv = (float*)((char*)const_cast<void*>(ws) + L.colWvOff);
    a.ntiles = a.nrows * a.f.planMW;
    const size_t lds = (solo_lds_bytes(PREC, fr->V) + 15) / 16 * 16;
    a.wave_floats = (int)(lds / sizeof(float));
    if (2 * lds > (size_t)160 * 1024) return gdb_fail(GDB_E_SHAPE, "V=%d needs %zu B of LDS per wave (two waves per workgroup exceed 160 KiB)", fr->V, lds);
    hipError_t e = launch_dense_n<PREC, WPS, 2, 1>(a, lds, st);
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "launch k_render_dense (centre rays): %s", hipGetErrorString(e));
    ColArgs c;
    c.f = a.f; c.wv = a.wv; c.tmp = a.bf; c.bf = bf; c.depth = depth; c.opac = opac; c.ldo = ldo; c.row_begin = a.row_begin; c.nrows = a.nrows;
    const int BB = cfg->bundle_size * cfg->bundle_size;
    const long long threads = (long long)fr->B * a.nrows * fr->W * BB;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (BB == 16) hipLaunchKernelGGL(k_bundle_colours<16>, dim3(grid), dim3(256), 0, st, c);
    else hipLaunchKernelGGL(k_bundle_colours<1>, dim3(grid), dim3(256), 0, st, c);
    e = hipGetLastError();
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "launch k_bundle_colours: %s", hipGetErrorString(e));
    return GDB_OK;
}

// one-wave workgroups a CU holds by LDS (allocated in 1280-byte granules out of 160 KiB)
static size_t waves_by_lds(size_t lds) { return (size_t)(160 * 1024) / ((lds + 1279) / 1280 * 1280); }

// Compute units of the current device, asked once per device ordinal (relaxed; two threads racing just ask twice).
static int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// LDS per wave of the one-wave schedules (segment wave, dense, flat): the views' staging rows, or the output record where that is larger
static size_t solo_lds_bytes(int prec, int V) {
    const size_t sv = prec == GDB_PREC_F16 ? stage_v<GDB_PREC_F16>() : stage_v<GDB_PREC_F32>();
    const size_t per_wave = sizeof(float) * (size_t)V * sv, rec = sizeof(float) * (size_t)(NOUT + 2) * COMP_LD;
    return per_wave > rec ? per_wave : rec;
}
// Shape limits of the two list schedules.  Dense lists bundles and rows in 16 bits.  Flat: its descriptors pack a row (16 bits) with a
// sample offset inside the row (16 bits), its lanes a map index (24 bits) with slot and count (4 + 4 bits), and the sample list is
// addressed with one 32-bit byte offset.
static bool dense_fits(const GdbFrame& fr) { return fr.W < 65536 && (long long)fr.B * fr.H < 65536; }
static bool flat_fits(const GdbConfig& cfg, const GdbFrame& fr, const WsLayout& L) {
    const int S = cfg.max_num_samples;
    return dense_fits(fr) && (long long)fr.W * S < 65536 && (long long)fr.H * fr.W < (1 << 24) && S <= 16 &&
           (unsigned long long)fr.B * fr.H * (unsigned long long)L.smapStride * 4ull < (1ull << 32);
}
// The ONE place that decides what a fused render call runs: `sched` as the caller passed it (GDB_SCHED_AUTO .. GDB_SCHED_FLAT) -> the
// schedule launched (GDB_SCHED_SLOT_WAVES .. GDB_SCHED_FLAT).  gdb_render_info() exports it (bench.py, HotPathEngine: no second copy of
// the rule in Python).  Measured choices (MI355X; profiles/r01/schedules.txt, r03/schedules.txt, r04/ab_*.txt, r05/ab_*.txt):
//  * slot waves   - a workgroup is one segment x S waves, one sample slot each, composite through LDS.  Fixed counts, S_max <= 3.
//  * segment wave - one wave walks all slots of its segment, composite in registers.  Fixed counts with S_max > 3 at f16 or with > 3 views.
//  * dense        - the compacted sample list, one wave per window of whole bundles holding <= 32 samples.  Adaptive counts (the slot
//                   schedules leave lanes idle: c2 80 % of the slot lanes busy, c4 44 %), and fixed counts per gdb_fixed_counts_dense.
//  * flat         - windows of exactly 32 consecutive samples (4-8 % fewer waves; straddling bundles composited by the last arriver).
//                   fp32 on frames of few tiles per wave slot with S_max <= 4 - c2: 8 % fewer tiles (6,112 instead of 6,631); not on
//                   c3 / c4 (greedy windows already fill 96 % of the lanes: 182.5 vs 176.0, 227.8 vs 204.2 us).
static int resolve_schedule(const GdbConfig& cfg, const GdbFrame& fr, const WsLayout& L, int prec, int nrows, int sched) {
    const int S = cfg.max_num_samples, V = fr.V;
    const bool one_wave_fits = solo_lds_bytes(prec, V) <= (size_t)160 * 1024;
    const long long slots = 12LL * device_cus();   // resident wave slots at three waves per SIMD
    // (B <= 2: the flat schedule launches once per batch item, the dense one once for all of them - c2 fp32, us per frame at B = 1 / 2 / 4: flat
    // 93.7 / 94.5 / 95.6, dense 97.6 / 94.5 / 89.1: profiles/r06/xp_frames_per_launch.txt)
    const bool flat_wins = prec == GDB_PREC_F32 && S <= 4 && fr.B <= 2 && ((long long)nrows * fr.W * S + 31) / 32 + 1 <= 3 * slots;   // worst case <= 3 tiles per wave slot
    if (one_wave_fits) {
        if (sched == GDB_SCHED_FLAT || (sched == GDB_SCHED_AUTO && cfg.is_adaptive && flat_fits(cfg, fr, L) && flat_wins)) return GDB_SCHED_FLAT;
        if (sched == GDB_SCHED_DENSE || (sched == GDB_SCHED_AUTO && dense_fits(fr) &&
                                         (cfg.is_adaptive || (prec != GDB_PREC_F16 && gdb_fixed_counts_dense(cfg, V))))) return GDB_SCHED_DENSE;
        if (sched == GDB_SCHED_SEGMENT_WAVE || (sched == GDB_SCHED_AUTO && S > 3)) return GDB_SCHED_SEGMENT_WAVE;
    }
    return GDB_SCHED_SLOT_WAVES;
}

template <int PREC>
static int render_launch(FusedArgs& a, const GdbConfig* cfg, const GdbFrame* fr, const WsLayout& L, const void* ws, int sched, bool plan_ready, hipStream_t st) {
    const int S = cfg->max_num_samples, V = fr->V;
    const size_t per_wave = sizeof(float) * (size_t)V * stage_v<PREC>();
    const size_t lds_max = 160 * 1024;
    const bool rec_fits = per_wave >= sizeof(float) * COMP_REC;
    hipError_t e = hipSuccess;
    const size_t solo_lds = solo_lds_bytes(PREC, V);
    a.ntiles = a.nsegs;
    unsigned grid = (unsigned)((a.ntiles + 7) / 8 * 8);
    size_t pad = 0;
#ifdef GDB_DIAG  // diagnostic build: extra LDS per workgroup lowers the occupancy (un-contended phase times under tools/stamps.py)
    static const size_t env_pad = getenv("GDB_FUSED_LDS_PAD") ? (size_t)atol(getenv("GDB_FUSED_LDS_PAD")) : 0;
    pad = env_pad;
#endif
    const int run = resolve_schedule(*cfg, *fr, L, PREC, a.nrows, sched);
    if (run == GDB_SCHED_FLAT || run == GDB_SCHED_DENSE) {
        if (run == GDB_SCHED_FLAT && !flat_fits(*cfg, *fr, L))
            return gdb_fail(GDB_E_SHAPE, "the flat schedule needs W x S_max < 65536, B x H < 65536, H x W < 2^24 (W = %d, H = %d, B = %d, S_max = %d)", fr->W, fr->H, fr->B, S);
        if (run == GDB_SCHED_DENSE && !dense_fits(*fr))
            return gdb_fail(GDB_E_SHAPE, "the dense schedule lists bundles and rows in 16 bits: W = %d, B x H = %lld (both must be < 65536)", fr->W, (long long)fr->B * fr->H);
        // The plan + sample list are built here, into the plan region of the caller's workspace (a launch of its own on the same
        // stream), unless the caller vouches that gdb_prepare built them from the depth prior as it stands (GDB_SCHED_PLAN_READY).
        if (!plan_ready) {
            int rc = gdb_build_dense_plan(cfg, fr, const_cast<void*>(ws), st);
            if (rc) return rc;
        }
        a.alias = 0;
        // the 3-waves-per-SIMD build (168 registers) wherever LDS admits more than the 8 waves per CU of the 2-wave build.
        // GDB_PREC_F16 (round 5): its list kernels need 142 registers whatever they are allowed (so there is no 2-wave build of them), and
        // fit into the 128 of FOUR waves per SIMD without a spill: that build wherever LDS admits all 16 waves per CU (V <= 4; c2 f16 42.2 ->
        // 39.8 us, c3 71.3 -> 67.0, c4 81.9 -> 78.2.  At V = 5 LDS admits 14, and the 4-wave build with 14 is SLOWER than the 3-wave one with
        // 12: c5 forced onto this schedule 744 -> 777 us, profiles/r05/ab_f16_four_waves.txt).
        const size_t by_lds = waves_by_lds(solo_lds + pad) > 2 * waves_by_lds(2 * (solo_lds + pad)) ? waves_by_lds(solo_lds + pad) : 2 * waves_by_lds(2 * (solo_lds + pad));
        bool three = false;
        if constexpr (PREC == GDB_PREC_F32X) three = by_lds > 8;
        const bool four = PREC == GDB_PREC_F16 && by_lds >= 16;
        const size_t lds = (solo_lds + pad + 15) / 16 * 16;
        if (run == GDB_SCHED_FLAT) {
            const int max_tiles = (int)(((long long)fr->H * fr->W * S + 31) / 32 + 1);
            if constexpr (PREC == GDB_PREC_F16) e = launch_flat<PREC, 3>(a, lds, max_tiles, st);   // (the flat body spills 7 registers at four waves per SIMD)
            else if constexpr (PREC == GDB_PREC_F32X) e = three ? launch_flat<PREC, 3>(a, lds, max_tiles, st) : launch_flat<PREC, 2>(a, lds, max_tiles, st);
            else e = launch_flat<PREC, 2>(a, lds, max_tiles, st);
            if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "launch k_render_flat: %s", hipGetErrorString(e));
            return GDB_OK;
        }
        a.ntiles = a.nrows * a.f.planMW;  // per batch item, worst case (every bundle at S_max)
        grid = (unsigned)((a.ntiles + 7) / 8 * 8);
        if constexpr (PREC == GDB_PREC_F16) e = four ? launch_dense<PREC, 4>(a, lds, st) : launch_dense<PREC, 3>(a, lds, st);
        else if constexpr (PREC == GDB_PREC_F32X) e = three ? launch_dense<PREC, 3>(a, lds, st) : launch_dense<PREC, 2>(a, lds, st);
        else e = launch_dense<PREC, 2>(a, lds, st);
    } else if (run == GDB_SCHED_SEGMENT_WAVE) {  // one wave per segment, all slots in turn
        a.alias = 0;
        // three waves per SIMD only where LDS admits them (12 one-wave workgroups per CU) and the precision's register budget does
        bool three = false;
        if constexpr (PREC == GDB_PREC_F16) three = waves_by_lds(solo_lds + pad) > 8;
        if constexpr (PREC == GDB_PREC_F16) { if (three) e = launch_solo<GDB_PREC_F16, 3>(a, grid, solo_lds + pad, st); }
        if (!three) e = launch_solo<PREC, 2>(a, grid, solo_lds + pad, st);
    } else if (rec_fits && S <= 8 && (size_t)S * per_wave <= lds_max) {  // one wave per slot
        a.alias = 1;
        if (S <= 4) e = launch_fused<false, 4, PREC>(a, grid, S, (size_t)S * per_wave + pad, st);
        else e = launch_fused<false, 8, PREC>(a, grid, S, (size_t)S * per_wave + pad, st);
    } else {  // more slots than waves fit: waves loop over slots, separate composite records
        const size_t fixed = sizeof(float) * (size_t)S * COMP_REC;
        int nw = S < 4 ? S : 4;
        nw = (S + (S + nw - 1) / nw - 1) / ((S + nw - 1) / nw);
        while (nw > 1 && fixed + nw * per_wave > lds_max) --nw;
        const size_t lds = fixed + nw * per_wave;
        if (lds > lds_max) return gdb_fail(GDB_E_SHAPE, "V=%d, S_max=%d needs %zu B of LDS (> %zu)", V, S, lds, lds_max);
        a.alias = 0;
        e = launch_fused<true, 4, PREC>(a, grid, nw, lds, st);
    }
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "launch k_render_fused: %s", hipGetErrorString(e));
    return GDB_OK;
}

// What a fused render call would do for this (config, frame shape, precision, row strip), without launching anything (ABI v6).
extern "C" int gdb_render_info(const GdbConfig* cfg, const GdbFrame* fr, int32_t precision, int32_t row_begin, int32_t row_end, int32_t out[4]) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, fr, false); if (rc) return rc;
    if (!out) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (precision != GDB_PREC_F16 && precision != GDB_PREC_F32 && precision != GDB_PREC_F32X) return gdb_fail(GDB_E_BADARG, "precision %d unsupported", precision);
    if (row_begin < 0 || row_end > fr->H || row_begin > row_end) return gdb_fail(GDB_E_SHAPE, "row strip [%d,%d) outside [0,%d]", row_begin, row_end, fr->H);
    const WsLayout L = ws_layout(*cfg, *fr);
    out[0] = fr->V >= 2 ? 1 : 0;   // (bundle_size 1 / 4 since round 6: the dense list kernel on the centre rays + k_bundle_colours)
    out[1] = !out[0] ? 0 : cfg->bundle_size != 2 ? GDB_SCHED_DENSE : resolve_schedule(*cfg, *fr, L, precision, row_end - row_begin, GDB_SCHED_AUTO);
    out[2] = (cfg->is_adaptive || gdb_fixed_counts_dense(*cfg, fr->V)) ? 1 : 0;   // gdb_prepare builds the plan when the frame carries d_depth_range (gdb_ops.hip prepare_common)
    out[3] = out[0] ? ((out[1] == GDB_SCHED_FLAT || (out[1] == GDB_SCHED_DENSE && fr->B > 1 && !(row_begin == 0 && row_end == fr->H))) ? fr->B : 1) + (cfg->bundle_size != 2 ? 1 : 0) : 0;
    return GDB_OK;
}

static int render_entry(const GdbConfig* cfg, const GdbFrame* fr, const void* ws, const float* pw, int32_t row_begin, int32_t row_end,
                        int32_t precision, int32_t schedule, float* bf, float* depth, float* opac, int ldo, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, fr, true); if (rc) return rc;
    const bool packed = ldo != NOUT;
    if (!ws || !pw || !bf || (!packed && (!depth || !opac))) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    const int b = cfg->bundle_size;
    if (b != 2) ldo = 3 * b * b + GDB_CFR + GDB_CV + (packed ? 2 : 0);   // output rows of bundle_size 1 / 4: Q = 3 b^2 + 27
    if (precision != GDB_PREC_F16 && precision != GDB_PREC_F32 && precision != GDB_PREC_F32X)
        return gdb_fail(GDB_E_BADARG, "precision %d unsupported (0 = f16 MFMA operands with f32 accumulate, 1 = f32 MFMA, 2 = split-f16 operands)", precision);
    const bool plan_ready = (schedule & GDB_SCHED_PLAN_READY) != 0, pyr16_ready = (schedule & GDB_SCHED_PYR16_READY) != 0;
    schedule &= ~(GDB_SCHED_PLAN_READY | GDB_SCHED_PYR16_READY);
    if (schedule < 0 || schedule > 4) return gdb_fail(GDB_E_BADARG, "schedule %d outside 0..4", schedule);
    // the reference's unbiased variance over views (nerf.py:73) is NaN for a single view
    if (fr->V < 2) return gdb_fail(GDB_E_SHAPE, "fused kernel needs at least 2 source views (got %d)", fr->V);
    if (row_begin < 0 || row_end > fr->H || row_begin > row_end) return gdb_fail(GDB_E_SHAPE, "row strip [%d,%d) outside [0,%d]", row_begin, row_end, fr->H);
    if (row_begin == row_end) return GDB_OK;
    WsLayout L = ws_layout(*cfg, *fr);
    // the kernel addresses each of these objects with a 32-bit byte offset from a uniform base
    const size_t lim = (size_t)1 << 32;
    if (L.pyrStride * 4 >= lim || (size_t)3 * fr->Ho * fr->Wo * 4 >= lim || (size_t)GDB_CV * fr->D * fr->H * fr->W * 4 >= lim)
        return gdb_fail(GDB_E_SHAPE, "frame too large for the fused kernel: a per-view pyramid, a source image or a cost volume exceeds 4 GiB");
    // ... and forms row offsets with 24-bit multiplies
    if (fr->Ho >= (1 << 24) || fr->Wo >= (1 << 24) || (size_t)fr->D * fr->H >= ((size_t)1 << 24))
        return gdb_fail(GDB_E_SHAPE, "frame too large for the fused kernel: an extent (or D*H) reaches 2^24");
    FusedArgs a;
    a.f = dev_frame(*cfg, *fr, L, ws);
    a.pw = pw;
    a.row_begin = row_begin; a.nrows = row_end - row_begin;
    a.nseg = (fr->W + 31) / 32;
    a.nsegs = fr->B * a.nrows * a.nseg;
    a.bf = bf; a.depth = depth; a.opac = opac; a.ldo = ldo; a.wv = nullptr;
    a.dbg = nullptr; a.skip = 0; a.wave_floats = 0; a.row_lo = a.row_hi = 0; a.tile_stride = 1; a.flat_base = 0;
    a.xp_stagger = 0; a.xp_per_rank = 0;
#ifdef GDB_DIAG  // diagnostic build: timing-only ablation bits from the environment, stamp buffer
    static const int env_skip = getenv("GDB_FUSED_SKIP") ? atoi(getenv("GDB_FUSED_SKIP")) : 0;
    a.skip = env_skip; a.dbg = g_dbg;
    a.xp_stagger = getenv("GDB_XP_STAGGER") ? atoi(getenv("GDB_XP_STAGGER")) : 0;   // (read per call: the experiment sweeps it inside one process)
    a.xp_per_rank = device_cus() / 8;   // workgroups of one dispatch rank per XCD: one per CU
#endif
    hipStream_t st = (hipStream_t)stream_;
    // GDB_PREC_F16 gathers its feature taps from the half-precision copy of the pyramid: made here (a launch of its own on the same
    // stream, from the fp32 pyramid) unless the caller vouches that gdb_prepare_ex(GDB_PREP_PYR16) made it for this frame
    if (precision == GDB_PREC_F16 && !pyr16_ready) {
        if (L.pyrStride * 2 >= lim) return gdb_fail(GDB_E_SHAPE, "frame too large for the fused kernel: a per-view pyramid exceeds 4 GiB");
        rc = gdb_build_pyr16(cfg, fr, const_cast<void*>(ws), st);
        if (rc) return rc;
    }
    if (b != 2) {
        // bundle_size 1 / 4: the dense list kernel on the bundles' centre rays, then the sub-ray colours (two launches; `schedule` beyond
        // its flags is ignored: this form has the one schedule).  f32x runs the fp32 kernel.
        // (f16: the three-waves-per-SIMD build - with the colour weights the centre-ray body spills four registers at four)
        if (precision == GDB_PREC_F16) return render_center_launch<GDB_PREC_F16, 3>(a, cfg, fr, L, ws, plan_ready, bf, depth, opac, ldo, st);
        return render_center_launch<GDB_PREC_F32, 2>(a, cfg, fr, L, ws, plan_ready, bf, depth, opac, ldo, st);
    }
    if (precision == GDB_PREC_F32) return render_launch<GDB_PREC_F32>(a, cfg, fr, L, ws, schedule, plan_ready, st);
    if (precision == GDB_PREC_F32X) return render_launch<GDB_PREC_F32X>(a, cfg, fr, L, ws, schedule, plan_ready, st);
    return render_launch<GDB_PREC_F16>(a, cfg, fr, L, ws, schedule, plan_ready, st);
}

extern "C" int gdb_render_bundles_fused(const GdbConfig* cfg, const GdbFrame* fr, const void* ws, const float* pw,
                                        int32_t row_begin, int32_t row_end, int32_t precision, int32_t schedule, float* bf,
                                        float* depth, float* opac, void* stream_) {
    return render_entry(cfg, fr, ws, pw, row_begin, row_end, precision, schedule, bf, depth, opac, NOUT, stream_);
}

extern "C" int gdb_render_bundles_packed(const GdbConfig* cfg, const GdbFrame* fr, const void* ws, const float* pw,
                                         int32_t row_begin, int32_t row_end, int32_t precision, int32_t schedule, float* out,
                                         void* stream_) {
    return render_entry(cfg, fr, ws, pw, row_begin, row_end, precision, schedule, out, nullptr, nullptr, NOUT + 2, stream_);
}
