// Next row N1: the residual-dense decoder that turns the 27 non-RGB bundle channels into the full-resolution
// colour residual — `Decoder.forward`, networks/gdb_nerf/decoder_rdn.py:44-81 of the reference (in_conv, 3 ResidualDenseBlocks
// with squeeze-excitation `:24-41`, `:7-21`, up-conv + PixelShuffle, 1x1 out_conv), called at network.py:51,170-175 — as HIP
// kernels for gfx950.
//
// * Layout: channel-last.  The fused hot path already writes bundle rows (N_b, Q) = (H, W, channels), so the decoder reads its
//   27 input channels straight out of `bundle_feat` (no permute / slice copy); activations are (N_b, 64) buffers and the dense
//   block's torch.cat([x, x1, x2]) is "channels 0-63 from P[b], 64-127 from Y": no concatenation copies.
// * 3x3 convolution = implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain, the reference's precision):
//   16 output channels on the MFMA rows, 16 consecutive pixels of an image row on the columns, K = (tap, input channel).  A
//   workgroup = 4 waves = 2 image rows x 32 pixels (1,280 workgroups on a 256x320 map = five per CU, all resident: five waves per
//   SIMD by registers and LDS); a wave owns ONE 16-channel output tile and as many pixels as the layer allows - both rows (32-channel
//   layers) or both rows and both 16-pixel halves (64-channel layers) - because every wave streams its tile's weights from L2 itself
//   and the stream, not the matrix pipe, is what such a layer loses time to (k_conv16's comment has the measurements).  Per
//   32-channel chunk of the input the workgroup stages its rows + halo in LDS (pixel stride 34 floats: the 4-byte B-operand reads of
//   a wave are bank-conflict-free).  Weights are packed on the host in operand order as 16-byte packets (pack_conv16).
// * The up stage is folded on the host: PixelShuffle is a permutation and there is no non-linearity between the 64 -> 256 up
//   convolution and the 1x1 out_conv, so out_conv o PixelShuffle o up = ONE 3x3 convolution 64 -> 12 (4 sub-pixels x 3 colours;
//   products summed in fp64, rounded once): 21x fewer flops for that stage, same function up to fp32 rounding.
// * Split-f16 variant of the convolution (GDB_PREC_F32X, k_conv3x3x): operands as f16 hi + lo pairs (about 22 bits), a product as
//   lo·hi + hi·lo + hi·hi on v_mfma_f32_32x32x16_f16 with fp32 accumulate — 3 x 32 cycles per 16 K instead of 8 x 64: the matrix
//   time of a layer falls 5.3x and the kernel becomes a staging / weight-stream problem.  Activations are split when they are
//   staged into LDS (hi and lo halves of a pixel's 32 channels side by side: the same 128 B as fp32), weights on the host.
// * Squeeze-excitation has no pass of its own: conv3's epilogue leaves per-segment channel sums (a cross-lane tree on registers the
//   wave holds anyway), ONE launch reduces them in a fixed order and runs the two tiny linears + sigmoid (k_se_gate, last-arriver
//   pattern: deterministic), and x += x3 * gate (+ the global residual at the end) happens while the NEXT convolution stages its
//   input (Stager<FUSE>).
#include "gdb_internal.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

int gdb_fail(int code, const char* fmt, ...);
int gdb_check_cfg(const GdbConfig* c);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned U2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

#define DEC_NF 64    // num_feats (network.py:51)
#define DEC_G 32     // growth rate (decoder_rdn.py:27)
#define DEC_PX 34    // pixels per staged row: 32 + halo
#define DEC_CHS 34   // floats per staged pixel: 32 channels of the chunk + 2 (bank spread for the 8-byte reads)
#define DECX_PXD 20  // split-f16 staging: dwords per pixel = 8 (hi halves of a chunk's 16 channels) + 8 (lo halves) + 4 (bank spread for the 16-byte reads)
#define DEC_SE_R 4   // SE bottleneck: 64 / 16

// ---- packed weights -------------------------------------------------------------------------------------------------
// conv_floats(cin, nt): floats of a packed layer of 32 nt output channels (fp32 form pack_conv16 and split-f16 form pack_conv_x alike).
static size_t conv_floats(int cin, int nt) { return (size_t)((cin + 31) / 32) * 9 * 8 * 64 * 2 * nt; }
#define DEC_MAX_LAYERS 16   // dense blocks (decoder_rdn.py:57 takes any count; every config of the reference has 3)
struct DecLayout {
    size_t in_w, in_b, blk[DEC_MAX_LAYERS][5] /* conv1, conv2, conv3, fc0, fc2 */, up_w, up_b;
    size_t in_wx, blkx[DEC_MAX_LAYERS][3], up_wx;  // the same convolutions as split-f16 fragments (pack_conv_x), behind the fp32 ones
    // upscale_factor 4 (bundle_size 4: two up stages, decoder_rdn.py:59-62): the FIRST stage's 64 -> 256 convolution as four 64 -> 64
    // convolutions, one per sub-pixel of its PixelShuffle (fp32 and split-f16 forms, biases); up_w / up_b then hold the SECOND stage folded
    // with out_conv.  Zero-sized at upscale_factor 2.
    size_t u1_w[4], u1_wx[4], u1_b[4];
    size_t total;
};
static DecLayout dec_layout(int nlayers, int bundle_size = 2) {
    DecLayout L{};
    size_t o = 0;
    L.in_w = o; o += conv_floats(27, 2);
    L.in_b = o; o += 64;
    for (int b = 0; b < nlayers; ++b) {
        L.blk[b][0] = o; o += conv_floats(64, 1);
        L.blk[b][1] = o; o += conv_floats(96, 1);
        L.blk[b][2] = o; o += conv_floats(128, 2);
        L.blk[b][3] = o; o += DEC_SE_R * DEC_NF;
        L.blk[b][4] = o; o += DEC_NF * DEC_SE_R;
    }
    L.up_w = o; o += conv_floats(64, 1);
    L.up_b = o; o += 32;
    L.in_wx = o; o += conv_floats(27, 2);
    for (int b = 0; b < nlayers; ++b) {
        L.blkx[b][0] = o; o += conv_floats(64, 1);
        L.blkx[b][1] = o; o += conv_floats(96, 1);
        L.blkx[b][2] = o; o += conv_floats(128, 2);
    }
    L.up_wx = o; o += conv_floats(64, 1);
    if (bundle_size == 4)
        for (int sp = 0; sp < 4; ++sp) {
            L.u1_w[sp] = o; o += conv_floats(64, 2);
            L.u1_wx[sp] = o; o += conv_floats(64, 2);
            L.u1_b[sp] = o; o += 64;
        }
    L.total = (o + 8 * 128 + 63) / 64 * 64;  // + one tap block: the fp32 conv kernel's weight prefetch runs one tap ahead
    return L;
}
// w: (cout, cin, 3, 3) row-major as torch stores it
// A layer for k_conv16 (v_mfma_f32_16x16x4_f32: 16 output rows x 16 pixels x 4 input channels per instruction, the same FLOP rate
// as the 32x32x2 form): [tile mt][32-channel chunk][tap 9][g4 2][lane 64][4] floats; element e of lane l =
// W[16 mt + (l & 15)][32 chunk + 4 (4 g4 + e) + (l >> 4)][tap] - the lane supplies A[row l & 15][k = l >> 4] of output tile mt for
// four consecutive k-groups: a tap of one tile is TWO 16-byte vector loads per wave.  A layer of <= 16 channels packs tile 0 only.
// Four tiles = conv_floats(cin, 2).
static void pack_conv16(const float* w, int cout, int cin, float* out) {
    const int nchunk = (cin + 31) / 32, nmt = (cout + 15) / 16;
    for (int mt = 0; mt < nmt; ++mt)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int g4 = 0; g4 < 2; ++g4)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int co = 16 * mt + (l & 15), ci = 32 * ch + 4 * (4 * g4 + e) + (l >> 4);
                            out[(((((size_t)mt * nchunk + ch) * 9 + tap) * 2 + g4) * 64 + l) * 4 + e] =
                                (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 9 + tap] : 0.f;
                        }
}

// The same layer as split-f16 A-operand fragments of v_mfma_f32_32x32x16_f16: [tile NT][16-channel chunk][tap 9][hi, lo][lane 64][8
// halfs]; element e of lane (i, h) = W[32 tile + i][16 chunk + 8 h + e][tap], hi = f16(w), lo = f16(w - hi).  The chunk count is
// padded to an even number, so the size in floats equals pack_conv's.
static void pack_conv_x(const float* w, int cout, int cin, int nt, float* out) {
    const int nchunk = (cin + 31) / 32 * 2;
    _Float16* o = (_Float16*)out;
    for (int t = 0; t < nt; ++t)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        int i = l & 31, h = l >> 5, co = 32 * t + i, ci = 16 * ch + 8 * h + e;
                        float v = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 9 + tap] : 0.f;
                        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                        const size_t base = (((size_t)t * nchunk + ch) * 9 + tap) * 2;
                        o[((base + 0) * 64 + l) * 8 + e] = hi;
                        o[((base + 1) * 64 + l) * 8 + e] = lo;
                    }
}

static int dec_check(const GdbConfig* cfg, int nlayers) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (cfg->bundle_size != 2 && cfg->bundle_size != 4)
        return gdb_fail(GDB_E_BADARG, "the HIP decoder is built for bundle_size 2 (one up stage) and 4 (two); got %d", cfg->bundle_size);
    if (nlayers < 1 || nlayers > DEC_MAX_LAYERS) return gdb_fail(GDB_E_BADARG, "decoder layers %d outside 1..%d", nlayers, DEC_MAX_LAYERS);
    return GDB_OK;
}

extern "C" int gdb_decoder_packed_floats(const GdbConfig* cfg, int32_t num_layers, size_t* out_floats) {
    int rc = dec_check(cfg, num_layers); if (rc) return rc;
    if (!out_floats) return gdb_fail(GDB_E_BADARG, "out_floats is NULL");
    *out_floats = dec_layout(num_layers, cfg->bundle_size).total;
    return GDB_OK;
}

// h_tensors in state-dict order (decoder_rdn.py:55-65): in_conv.weight (64,27,3,3), in_conv.bias, then per block conv1.weight
// (32,64,3,3), conv2.weight (32,96,3,3), conv3.weight (64,128,3,3), se.fc.0.weight (4,64), se.fc.2.weight (64,4), then
// up.0.weight (256,64,3,3), up.0.bias (256), [bundle_size 4: up.2.weight (256,64,3,3), up.2.bias (256),] out_conv.weight (3,64,1,1),
// out_conv.bias (3): 2 + 5 num_layers + 4 (+ 2) pointers.
extern "C" int gdb_pack_decoder_weights(const GdbConfig* cfg, int32_t num_layers, const float* const* t, float* out) {
    int rc = dec_check(cfg, num_layers); if (rc) return rc;
    if (!t || !out) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    const bool up4 = cfg->bundle_size == 4;
    const int n = 2 + 5 * num_layers + 4 + (up4 ? 2 : 0);
    for (int i = 0; i < n; ++i)
        if (!t[i]) return gdb_fail(GDB_E_BADARG, "decoder tensor %d is NULL", i);
    const DecLayout L = dec_layout(num_layers, cfg->bundle_size);
    memset(out, 0, sizeof(float) * L.total);
    const int cin0 = GDB_CFR + GDB_CV;  // 27
    pack_conv16(t[0], DEC_NF, cin0, out + L.in_w);
    pack_conv_x(t[0], DEC_NF, cin0, 2, out + L.in_wx);
    memcpy(out + L.in_b, t[1], sizeof(float) * DEC_NF);
    for (int b = 0; b < num_layers; ++b) {
        const float* const* q = t + 2 + 5 * b;
        pack_conv16(q[0], DEC_G, DEC_NF, out + L.blk[b][0]);
        pack_conv16(q[1], DEC_G, DEC_NF + DEC_G, out + L.blk[b][1]);
        pack_conv16(q[2], DEC_NF, DEC_NF + 2 * DEC_G, out + L.blk[b][2]);
        pack_conv_x(q[0], DEC_G, DEC_NF, 1, out + L.blkx[b][0]);
        pack_conv_x(q[1], DEC_G, DEC_NF + DEC_G, 1, out + L.blkx[b][1]);
        pack_conv_x(q[2], DEC_NF, DEC_NF + 2 * DEC_G, 2, out + L.blkx[b][2]);
        memcpy(out + L.blk[b][3], q[3], sizeof(float) * DEC_SE_R * DEC_NF);
        memcpy(out + L.blk[b][4], q[4], sizeof(float) * DEC_NF * DEC_SE_R);
    }
    // fold: out_conv o PixelShuffle(2) o up.  PixelShuffle: up channel 4 k + s (s = dy*2 + dx) -> feature k of sub-pixel s.
    // folded output channel c = 3 s + o:  W[c][ci][tap] = sum_k Wout[o][k] Wup[4k + s][ci][tap],  b[c] = sum_k Wout[o][k] bup[4k + s] + bout[o]
    // (upscale_factor 4: there is no non-linearity anywhere in `up` (decoder_rdn.py:59-62, :79-80), so the LAST stage folds with out_conv
    // exactly as the single stage of upscale_factor 2 does; the first stage stays a convolution - four of them, one per sub-pixel)
    const float* wup = t[n - 4]; const float* bup = t[n - 3]; const float* wout = t[n - 2]; const float* bout = t[n - 1];
    if (up4) {
        const float* w0 = t[n - 6]; const float* b0 = t[n - 5];
        std::vector<float> ws((size_t)DEC_NF * DEC_NF * 9);
        for (int sp = 0; sp < 4; ++sp) {   // PixelShuffle: conv channel 4 k + sp -> feature k of sub-pixel sp = dy * 2 + dx
            for (int k = 0; k < DEC_NF; ++k) {
                memcpy(ws.data() + (size_t)k * DEC_NF * 9, w0 + (size_t)(4 * k + sp) * DEC_NF * 9, sizeof(float) * DEC_NF * 9);
                out[L.u1_b[sp] + k] = b0[4 * k + sp];
            }
            pack_conv16(ws.data(), DEC_NF, DEC_NF, out + L.u1_w[sp]);
            pack_conv_x(ws.data(), DEC_NF, DEC_NF, 2, out + L.u1_wx[sp]);
        }
    }
    std::vector<float> wf((size_t)12 * DEC_NF * 9);
    for (int s = 0; s < 4; ++s)
        for (int o = 0; o < 3; ++o) {
            for (int ci = 0; ci < DEC_NF; ++ci)
                for (int tap = 0; tap < 9; ++tap) {
                    double acc = 0;
                    for (int k = 0; k < DEC_NF; ++k) acc += (double)wout[o * DEC_NF + k] * (double)wup[((size_t)(4 * k + s) * DEC_NF + ci) * 9 + tap];
                    wf[((size_t)(3 * s + o) * DEC_NF + ci) * 9 + tap] = (float)acc;
                }
            double acc = bout[o];
            for (int k = 0; k < DEC_NF; ++k) acc += (double)wout[o * DEC_NF + k] * (double)bup[4 * k + s];
            out[L.up_b + 3 * s + o] = (float)acc;
        }
    pack_conv16(wf.data(), 12, DEC_NF, out + L.up_w);
    pack_conv_x(wf.data(), 12, DEC_NF, 1, out + L.up_wx);
    return GDB_OK;
}

// ---- workspace -------------------------------------------------------------------------------------------------------
// Activations are (pixels, 64) buffers: the input x of dense block b (block 0's = in_conv's output, kept to the end as the decoder's
// global residual, in P[0]; block b >= 1's in P[1 + (b - 1) % 2]: a block reads its predecessor's x and writes its own, so two buffers
// alternate whatever the number of blocks), Y = [x1 | x2] of the block in flight, T = conv3's output; the dense block's torch.cat([x, x1, x2]) is "chunks 0-1
// from P[b], chunks 2-3 from Y".  part = per-(row, 32-pixel segment) channel sums of T (written by conv3's epilogue), part2 = their
// sums per group of DEC_SEG segments, gate = the block's 64 gates per batch item, count = one arrival counter per batch item.
#ifndef DEC_SEG
#define DEC_SEG 128   // segments per workgroup of k_se_gate's first stage (256x320, 2,560 segments: 16 / 32 / 64 / 128 / 256 -> 18.4 / 10.8 / 7.9 / 6.6 / 7.8 us: the last arriver's serial tail)
#endif
struct DecWs { size_t P[3], Y, T, part, part2, gate, count, X, U, total; int nseg, ngrp; };   // X, U: upscale_factor 4 only (the blocks' output with the last gate applied; the first up stage's (2H, 2W, 64) map)
static DecWs dec_ws(int B, int H, int W, int bundle_size = 2) {
    DecWs w{};
    const size_t n = (size_t)B * H * W;
    w.nseg = H * ((W + 31) / 32);
    w.ngrp = (w.nseg + DEC_SEG - 1) / DEC_SEG;
    size_t o = 0;
    for (int i = 0; i < 3; ++i) { w.P[i] = o; o = align_up(o + sizeof(float) * n * DEC_NF, 256); }
    w.Y = o; o = align_up(o + sizeof(float) * n * DEC_NF, 256);
    w.T = o; o = align_up(o + sizeof(float) * n * DEC_NF, 256);
    w.part = o; o = align_up(o + sizeof(float) * (size_t)B * w.nseg * DEC_NF, 256);
    w.part2 = o; o = align_up(o + sizeof(float) * (size_t)B * w.ngrp * DEC_NF, 256);
    w.gate = o; o = align_up(o + sizeof(float) * (size_t)B * DEC_NF, 256);
    w.count = o; o = align_up(o + sizeof(unsigned) * (size_t)B, 256);
    w.X = o; w.U = o;
    if (bundle_size == 4) {
        o = align_up(o + sizeof(float) * n * DEC_NF, 256);
        w.U = o; o = align_up(o + sizeof(float) * 4 * n * DEC_NF, 256);
    }
    w.total = o;
    return w;
}
extern "C" int gdb_decoder_workspace_bytes(const GdbConfig* cfg, const GdbFrame* shape, size_t* out_bytes) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!shape || !out_bytes) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (shape->B < 1 || shape->H < 1 || shape->W < 1) return gdb_fail(GDB_E_SHAPE, "non-positive bundle map");
    *out_bytes = dec_ws(shape->B, shape->H, shape->W, cfg->bundle_size).total;
    return GDB_OK;
}

// ---- 3x3 convolution ----------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* in; int in_stride, in_off, cin, nchunk, vec;  // vec: the input rows allow 16-byte loads
    const float* in2; int split;    // 32-channel chunks >= split come from in2 (row stride in_stride, channel 32 (chunk - split))
    const float* w; const float* bias;
    float* out; int out_stride, out_off, cout, relu;
    float* rgb;                     // folded up stage: (B,3,2H,2W) NCHW, channel c = 3 s + o of sub-pixel s = dy*2 + dx
    int up2, up_dy, up_dx;          // first up stage of upscale_factor 4: output pixel (y, x) is written at (2y + up_dy, 2x + up_dx) of a (2H, 2W) map
    // squeeze-excitation folded into the consumer (FUSE kernels; decoder_rdn.py:40,78): the staged input is x = in + fT * gate (+ fS),
    // all (pixels, 64); the workgroup's own pixels of x are also written to fX (the next block's P) when fX is set
    const float* fT; const float* fgate; const float* fS; float* fX;
    float* se_part;                 // SEP kernels (conv3): channel sums of the output per (row, 32-pixel segment)
    unsigned* zero;                 // in_conv: the arrival counters of k_se_gate, cleared for this decode
    int B, H, W, tilesX, tilesY;
};

extern __shared__ float dsmem[];

// Load through (wave-uniform base, 32-bit byte offset): the base stays in SGPRs and the offset is one VGPR instead of a 64-bit
// per-lane address per load (ten registers of a staging plan's five slots, which is what spilled at five waves per SIMD).
template <typename T>
__device__ __forceinline__ T ldu(const void* __restrict__ base, unsigned byte_off) {
    return *(const T*)((const char*)base + byte_off);
}

// Staging plan of a 256-thread workgroup: rows y0-1 .. y0+TR of the 34-pixel column x0-1 .. x0+32, one chunk of 4 NG input channels
// at a time.  The (pixel, 4-channel group) slots a thread fills are the same for every chunk, so their source offsets are computed
// once (the divisions by 34 are not cheap, and VALU cycles are matrix cycles on the fp32 datapath).  The loads are branch-free (a slot
// outside the image or beyond the layer's channels loads a valid address and is zeroed in value()) and run one chunk ahead: a chunk's
// global loads are issued before the previous chunk's MFMAs and consumed after them.
//   VEC:  the input rows allow 16-byte loads and the channel count is a multiple of 4 (every layer but in_conv) - a template parameter,
//         not a run-time branch: as `if (a.vec)` each slot became its own basic block ending in s_waitcnt vmcnt(0), five serial memory
//         round trips at the top of every chunk instead of five loads in flight under the previous chunk's MFMAs.
//   FUSE: the squeeze-excitation apply of the PREVIOUS block happens here (x = in + fT * gate, + fS for the decoder's global residual,
//         rounded exactly as the separate element-wise pass rounded them): the pass itself - 63 MB of traffic, 11 us per block at the
//         DTU size - is gone, and so is the second copy of in_conv's output it needed.
template <int TR, int NG, bool VEC, bool FUSE>
struct Stager {
    static constexpr int CH = 4 * NG;                       // channels per chunk
    static constexpr int NPG = (TR + 2) * DEC_PX * NG;      // (pixel, group) slots of a chunk
    static constexpr int NSLOT = (NPG + 255) / 256;
    static constexpr int LAST = NPG - 256 * (NSLOT - 1);    // threads that own a slot in the last round
    static constexpr int SH = NG == 8 ? 3 : 2;
    unsigned soff[NSLOT];   // byte offset of the slot's pixel inside the input (clamped into the image), channel group excluded
    unsigned inimg, inner;  // bit s: the slot's pixel lies inside the image / is one of the workgroup's own output pixels
    int g4;                 // the slot's 4-channel group inside a chunk: (tid + 256 s) & (NG - 1) = tid & (NG - 1)
    F4 pre[NSLOT], preT[FUSE ? NSLOT : 1], gate4;
    const float* inb; const float* in2b; const float* fTb; const float* fSb; float* fXb;

    __device__ __forceinline__ void init(const ConvArgs& a, int tid, int b, int x0, int y0) {
        const size_t img = (size_t)b * a.H * a.W;
        inb = a.in + img * a.in_stride;   // (a frame's input is < 2^30 floats: checked on the host)
        in2b = a.in2 ? a.in2 + img * a.in_stride : inb;
        if (FUSE) {
            fTb = a.fT + img * DEC_NF; fSb = a.fS ? a.fS + img * DEC_NF : nullptr; fXb = a.fX ? a.fX + img * DEC_NF : nullptr;
            gate4 = F4{0.f, 0.f, 0.f, 0.f};
        }
        g4 = 4 * (tid & (NG - 1));
        inimg = 0; inner = 0;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int idx = tid + 256 * s;
            const int p = idx >> SH, rx = p % DEC_PX, ry = p / DEC_PX;
            const int px = x0 - 1 + rx, py = y0 - 1 + ry;
            const bool slot = idx < NPG;
            const bool in = slot && px >= 0 && px < a.W && py >= 0 && py < a.H;
            inimg |= (in ? 1u : 0u) << s;
            inner |= ((in && rx >= 1 && rx <= 32 && ry >= 1 && ry <= TR) ? 1u : 0u) << s;
            soff[s] = 4u * (unsigned)(in ? (int)(((size_t)py * a.W + px) * a.in_stride + a.in_off) : a.in_off);
        }
    }
    __device__ __forceinline__ bool has(int s, int tid) const { return s < NSLOT - 1 || tid < LAST; }
    // issue the loads of chunk ch (wave-uniform)
    __device__ __forceinline__ void fetch(const ConvArgs& a, int b, int ch) {
        const int split = a.split * (8 / NG);
        const float* src = ch < split ? inb : in2b;
        const int cl = CH * (ch < split ? ch : ch - split) + g4, ci = CH * ch + g4;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            if (VEC) pre[s] = ldu<F4>(src, soff[s] + 4u * (unsigned)(ci + 3 < a.cin ? cl : 0));
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) pre[s][k] = ldu<float>(src, soff[s] + 4u * (unsigned)min(cl + k, a.cin - 1));
            }
            if (FUSE) preT[s] = ldu<F4>(fTb, soff[s] + 4u * (unsigned)cl);
        }
        if (FUSE) gate4 = *(const F4*)(a.fgate + (size_t)b * DEC_NF + cl);
    }
    // the value of slot s of chunk ch as it is staged: zero outside the image (padding = 1) and beyond the layer's channels
    __device__ __forceinline__ F4 value(const ConvArgs& a, int s, int ch) {
        const int ci = CH * ch + g4;
        const bool in = (inimg >> s) & 1;
        F4 v = pre[s];
        if (FUSE) {
            const F4 tg = preT[s] * gate4;
            v = v + tg;
            if (fSb) v = v + ldu<F4>(fSb, soff[s] + 4u * (unsigned)ci);
            if (fXb && ((inner >> s) & 1)) *(F4*)((char*)fXb + soff[s] + 4u * (unsigned)ci) = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (in && (VEC ? ci + 3 : ci + k) < a.cin) ? v[k] : 0.f;
        return v;
    }
};

// Sum of v over the 32 lanes of the caller's half of the wave (fixed tree: deterministic), valid in every lane.
__device__ __forceinline__ float half_wave_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    v += __shfl_xor(v, 16);
    return v;
}

// Sum of v over the 16 lanes of the caller's row group (fixed tree), valid in every lane.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

// Every fp32 convolution of the decoder, on v_mfma_f32_16x16x4_f32.  A workgroup is 2 image rows x 32 pixels = 4 waves; a wave owns
// ONE 16-channel output tile mt and as many of the workgroup's pixels as the layer's tile count leaves it:
//   NPH = 2, NR = 2 (64 output channels: in_conv, conv3): wave = tile mt of four; both rows, both 16-pixel halves = 64 pixels, four
//          accumulators;
//   NPH = 1, NR = 2 (17..32 channels: conv1, conv2):      wave = (pixel half, tile mt of two); both rows = 32 pixels;
//   NPH = 1, NR = 1 (<= 16 channels: the folded up stage): wave = (pixel half, row), tile 0 = 16 pixels.
// Either way 5,120 equal waves on a 256x320 map = five per SIMD, all resident (<= 96 registers, 18.5 KB of LDS per workgroup).
// Why pixels per wave matter: each wave streams its tile's weights from L2 itself (16-byte packets = four k-groups, the next tap's
// requested under the current tap's MFMAs), a packet feeds NPH x NR MFMAs, and that stream - not the matrix pipe, not the staging -
// is what a layer loses time to.  Measured on one box (profiles/r04/decoder_experiments.txt): the 128 -> 64 layer as 32 pixels x 32
// channels per wave on v_mfma_f32_32x32x2_f32 (147 KB of weights per wave) 108.7 us, as 64 pixels x 16 channels (74 KB) 95.7 us, its
// bare MFMA loop ~87; round 3's 8-byte packets with a (row, half) wave on both tiles of a 32-channel layer: the weight loads alone,
// 72 per chunk and wave at 16 addresser cycles each, equalled the layer's matrix time.  The pixels come from LDS, which has the
// bandwidth: rows + halo of a 32-channel chunk, pixel stride 34 floats (lane l reads channel 4 g + (l >> 4) of pixel l & 15: 64
// distinct banks), one ds_read_b32 per MFMA.
// SEP: the epilogue also leaves the segment's per-channel sums of the output for the squeeze-excitation mean (decoder_rdn.py:17-21) -
// a cross-lane tree on registers the wave holds anyway, instead of a pass over T.
template <int NR, bool VEC, bool FUSE, int NPH = 1, bool SEP = false>
__global__ void __launch_bounds__(256, 5) k_conv16(ConvArgs a) {
    constexpr int TR = 2;        // rows per workgroup
    float* lds = dsmem;          // [(TR + 2)][DEC_PX][DEC_CHS]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, kq = lane >> 4;
    const int ph = NPH == 2 ? 0 : wid & 1;                        // this wave's (first) 16-pixel half of the column
    const int mt = NPH == 2 ? wid : (NR == 2 ? wid >> 1 : 0);     // its 16-channel output tile
    const int wrow = NR == 2 ? 0 : wid >> 1;                      // its first row of the workgroup
    const int bx = blockIdx.x % a.tilesX, by = (blockIdx.x / a.tilesX) % a.tilesY, b = blockIdx.x / (a.tilesX * a.tilesY);
    const int x0 = bx * 32, y0 = by * TR;
    if (a.zero && blockIdx.x == 0 && tid < a.B) a.zero[tid] = 0u;
    F4 acc[NPH][NR];   // D layout: register r of lane l = output row 4 (l >> 4) + r of pixel l & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 16 * mt + 4 * kq + r;
        const float bv = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[p][q][r] = bv;
    }
    Stager<TR, 8, VEC, FUSE> st;
    st.init(a, tid, b, x0, y0);
    const int loff0 = (tid >> 3) * DEC_CHS + st.g4;
    const float* wbase = a.w + (size_t)mt * a.nchunk * 9 * 2 * 64 * 4;   // wave-uniform
    const unsigned wlane = 16u * (unsigned)lane;
    const float* brow = lds + (size_t)(wrow * DEC_PX + 16 * ph + pl) * DEC_CHS + kq;
    st.fetch(a, b, 0);
    F4 wn[2];   // the next tap's weights, loaded while the current tap's MFMAs run
#pragma unroll
    for (int g = 0; g < 2; ++g) wn[g] = ldu<F4>(wbase, wlane + 1024u * g);
    for (int ch = 0; ch < a.nchunk; ++ch) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < st.NSLOT; ++s) {
            if (!st.has(s, tid)) continue;
            const F4 v = st.value(a, s, ch);
            F2* dst = (F2*)(lds + loff0 + s * 32 * DEC_CHS);
            dst[0] = F2{v[0], v[1]}; dst[1] = F2{v[2], v[3]};
        }
        __syncthreads();
        if (ch + 1 < a.nchunk) st.fetch(a, b, ch + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            F4 w[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) w[g] = wn[g];
            {   // (the last prefetch of a tile reads the packed buffer's next tile / layer / tail padding: in bounds, never used)
                const float* wnext = wbase + ((size_t)ch * 9 + tap + 1) * 2 * 256;
#pragma unroll
                for (int g = 0; g < 2; ++g) wn[g] = ldu<F4>(wnext, wlane + 1024u * g);
            }
            __builtin_amdgcn_sched_barrier(0);   // the loads stay ahead of this tap's MFMAs (left alone the scheduler sank each next to its use)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int p = 0; p < NPH; ++p)
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const float bv = brow[(size_t)((dy + q) * DEC_PX + dx + 16 * p) * DEC_CHS + 4 * g];
                        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g >> 2][g & 3], bv, acc[p][q], 0, 0, 0);
                    }
            }
        }
    }
    // epilogue: lane (pl, kq) holds output channels 16 mt + 4 kq + (0..3) of pixels (y0 + wrow + q, x0 + 16 (ph + p) + pl)
    const int co = 16 * mt + 4 * kq;
    if (SEP) {   // the segment's channel sums: part[b][y * tilesX + bx][64]; a sum over the 16 lanes of a row group, both pixel halves
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int y = y0 + wrow + q;
            if (y >= a.H) continue;   // (wave-uniform)
            F4 sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = 0.f;
#pragma unroll
                for (int p = 0; p < NPH; ++p) v += (x0 + 16 * (ph + p) + pl < a.W) ? acc[p][q][k] : 0.f;
                sum[k] = row16_sum(v);
            }
            if (pl == 0) *(F4*)(a.se_part + (((size_t)b * a.H + y) * a.tilesX + bx) * DEC_NF + co) = sum;
        }
    }
    if (co >= a.cout) return;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
        const int x = x0 + 16 * (ph + p) + pl;
        if (x >= a.W) continue;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int y = y0 + wrow + q;
            if (y >= a.H) continue;
            const size_t pix = a.up2 ? ((size_t)b * 2 * a.H + 2 * y + a.up_dy) * (2 * (size_t)a.W) + 2 * x + a.up_dx
                                     : (size_t)b * a.H * a.W + (size_t)y * a.W + x;
            F4 v = acc[p][q];
            if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (a.rgb) {  // folded up stage: channel c = 3 s + o -> rgb[b][o][2y + (s >> 1)][2x + (s & 1)]
                const int Ho = 2 * a.H, Wo = 2 * a.W;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = co + k;
                    if (c < 12) {
                        const int sp = c / 3, o = c - 3 * sp;
                        a.rgb[(((size_t)b * 3 + o) * Ho + 2 * y + (sp >> 1)) * Wo + 2 * x + (sp & 1)] = v[k];
                    }
                }
            } else {
                *(F4*)(a.out + pix * a.out_stride + a.out_off + co) = v;
            }
        }
    }
}


// Split-f16 variant (GDB_PREC_F32X).  Workgroup = 4 waves = 4 image rows of one 32-pixel column, all on the SAME 32-channel
// output tile (a 64-channel layer launches its two tiles as separate workgroups).  Per 16-channel chunk of the input (one MFMA
// K-step) the workgroup stages in LDS (a) its rows + halo, each pixel's channels as 16 hi halves then 16 lo halves (pixel stride
// 20 dwords: the 16-byte B-operand reads of a lane group fall on distinct banks), converted from fp32 on the way in, and (b) the
// chunk's 18 KiB of weight fragments [9 taps][hi, lo][64 lanes][16 B]; a tap is then four ds_read_b128 (A hi / lo, B hi / lo) and
// three MFMAs.  34 KB of LDS per workgroup: four workgroups per CU, so one's staging runs under the others' MFMAs.  The next
// chunk's global loads (pixels and weights) are issued before the chunk's MFMAs and converted / stored after them.
// Measured alternatives (profiles/r02/decoder_split_f16_variants.txt): weights streamed per wave from L2 into a register ring, 1
// or 2 rows per wave — every wave then pulls its weights through the CU's texture path (64 B/clk) and that stream, not the
// matrix pipe, sets the time (0.43-0.49 ms); 32-channel chunks (66 KB of LDS, two workgroups per CU) 0.37-0.42 ms.  What is left
// is not matrix time (one MFMA instead of three per product: 0.36 -> 0.32 ms) nor the global loads (none after the first chunk:
// 0.33 ms) but the two-barrier staging cycle per chunk; see DESIGN.md section 8.
constexpr int DECX_ROWS = 1;   // image rows per wave
#ifndef DECX_PF
#define DECX_PF 1              // taps of LDS operand reads in flight ahead of the MFMAs (1, 2, 3 measured the same)
#endif
// Staging plan, VEC, FUSE: Stager<> above (16-channel chunks, four groups per pixel); SEP as in k_conv3x3.
template <bool VEC, bool FUSE, bool SEP>
__global__ void __launch_bounds__(256) k_conv3x3x(ConvArgs a) {
    constexpr int R = DECX_ROWS, TR = 4 * R;
    constexpr int PIXD = (TR + 2) * DEC_PX * DECX_PXD;   // dwords of the pixel image
    constexpr int WFRAG = 9 * 2 * 64;                    // half8 per (tile, chunk) of weights
    unsigned* lds = (unsigned*)dsmem;
    unsigned* wl = lds + PIXD;                           // 9 * 2 * 64 * 4 dwords
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int wrow0 = wid * R;
    const int nt = (a.cout + 31) / 32;
    const int nchunk = 2 * a.nchunk;                     // 16-channel chunks (a.nchunk counts 32-channel ones)
    int blk = blockIdx.x;
    const int t = blk % nt; blk /= nt;
    const int bx = blk % a.tilesX, by = (blk / a.tilesX) % a.tilesY, b = blk / (a.tilesX * a.tilesY);
    const int x0 = bx * 32, y0 = by * TR;
    if (a.zero && blockIdx.x == 0 && tid < a.B) a.zero[tid] = 0u;
    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float bv = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q][r] = bv;
    }
    Stager<TR, 4, VEC, FUSE> st;
    st.init(a, tid, b, x0, y0);
    const int loff0 = (tid >> 2) * DECX_PXD + (st.g4 >> 1);   // dword offset of slot 0's hi halves in LDS (lo halves 8 dwords further); slot s: 64 pixels on
    constexpr int NW = (WFRAG + 255) / 256;
    const half8* wsrc = (const half8*)a.w + (size_t)t * nchunk * WFRAG;
    half8 wpre[NW];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NW; ++k) wpre[k] = wsrc[(size_t)ch * WFRAG + min(tid + 256 * k, WFRAG - 1)];
        st.fetch(a, b, ch);
    };
    fetch(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();  // the previous chunk's reads are done
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (tid + 256 * k < WFRAG) *(half8*)(wl + (size_t)(tid + 256 * k) * 4) = wpre[k];
#pragma unroll
        for (int s = 0; s < st.NSLOT; ++s) {
            if (!st.has(s, tid)) continue;
            const F4 v = st.value(a, s, ch);
            unsigned* dst = lds + loff0 + s * 64 * DECX_PXD;
            const half2v h01 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
            const half2v h23 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]));
            const U2 hi = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
            const U2 lo = {__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h01.x, v[1] - (float)h01.y)),
                           __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h23.x, v[3] - (float)h23.y))};
            *(U2*)dst = hi;
            *(U2*)(dst + 8) = lo;
        }
        __syncthreads();
        if (ch + 1 < nchunk) fetch(ch + 1);   // flies under this chunk's MFMAs
        // operand ring: the four ds_read_b128 of tap + DECX_PF are issued before the MFMAs of a tap (left to the compiler each
        // read sat directly in front of its MFMA behind an lgkmcnt wait: an LDS round trip per 96 cycles of matrix work)
        half8 opA[DECX_PF + 1][2], opB[DECX_PF + 1][R][2];
        auto load_ops = [&](int tap, int slot) {
            const int dy = tap / 3, dx = tap % 3;
            opA[slot][0] = *(const half8*)(wl + (size_t)((2 * tap) * 64 + lane) * 4);
            opA[slot][1] = *(const half8*)(wl + (size_t)((2 * tap + 1) * 64 + lane) * 4);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const unsigned* bp = lds + (size_t)((wrow0 + q + dy) * DEC_PX + j + dx) * DECX_PXD + 4 * h;
                opB[slot][q][0] = *(const half8*)bp;
                opB[slot][q][1] = *(const half8*)(bp + 8);
            }
        };
#pragma unroll
        for (int tap = 0; tap < DECX_PF; ++tap) load_ops(tap, tap % (DECX_PF + 1));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + DECX_PF < 9) load_ops(tap + DECX_PF, (tap + DECX_PF) % (DECX_PF + 1));
            __builtin_amdgcn_sched_barrier(0);   // keeps the reads above the MFMAs below (a compiler-only memory fence did not)
            const int cur = tap % (DECX_PF + 1);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                acc[q] = MFMA16(opA[cur][1], opB[cur][q][0], acc[q]);
                acc[q] = MFMA16(opA[cur][0], opB[cur][q][1], acc[q]);
                acc[q] = MFMA16(opA[cur][0], opB[cur][q][0], acc[q]);
            }
        }
    }
    const int x = x0 + j;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int y = y0 + wrow0 + q;
        if (y >= a.H) continue;   // (wave-uniform)
        const bool valid = x < a.W;
        if (SEP) {   // the segment's channel sums: part[b][y * tilesX + bx][64]
            float* dst = a.se_part + (((size_t)b * a.H + y) * a.tilesX + bx) * DEC_NF + 32 * t + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                F4 sum;
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[k] = half_wave_sum(valid ? acc[q][4 * g + k] : 0.f);
                if (j == 0) *(F4*)(dst + 8 * g) = sum;
            }
        }
        if (!valid) continue;
        const size_t pix = a.up2 ? ((size_t)b * 2 * a.H + 2 * y + a.up_dy) * (2 * (size_t)a.W) + 2 * x + a.up_dx
                                 : (size_t)b * a.H * a.W + (size_t)y * a.W + x;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * t + 8 * g + 4 * h;
            if (co >= a.cout) continue;
            F4 v = {acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
            if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (a.rgb) {
                const int Ho = 2 * a.H, Wo = 2 * a.W;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = co + k;
                    if (c < 12) {
                        const int s = c / 3, o = c - 3 * s;
                        a.rgb[(((size_t)b * 3 + o) * Ho + 2 * y + (s >> 1)) * Wo + 2 * x + (s & 1)] = v[k];
                    }
                }
            } else {
                *(F4*)(a.out + pix * a.out_stride + a.out_off + co) = v;
            }
        }
    }
}

// ---- squeeze-excitation ------------------------------------------------------------------------------------------------
// gate[b][c] = sigmoid(W2 relu(W1 mean_c))   decoder_rdn.py:17-21, from the per-segment channel sums conv3's epilogue left in `part`
// ([b][segment][64]).  One launch, two stages: a workgroup sums DEC_SEG segments (4 thread groups x 64 channels, each group its
// segments in order, then a fixed tree over the groups) into part2[b][group]; the workgroup that arrives LAST at the batch item's
// counter (release / acquire fences around a device-scope atomic) sums the groups in index order and runs the two tiny linears.  Every
// sum has a fixed shape whatever the arrival order: bit-reproducible.  The counter is left at zero for the next block.
__global__ void __launch_bounds__(256) k_se_gate(const float* __restrict__ part, int nseg, int ngrp, float inv_hw, const float* __restrict__ w1,
                                                 const float* __restrict__ w2, float* __restrict__ part2, unsigned* __restrict__ count,
                                                 float* __restrict__ gate) {
    __shared__ float red[256];
    __shared__ float mean[DEC_NF], hid[DEC_SE_R];
    __shared__ unsigned ticket;
    const int b = blockIdx.x / ngrp, grp = blockIdx.x % ngrp;
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float* src = part + (size_t)b * nseg * DEC_NF;
    const int s0 = grp * DEC_SEG, s1 = min(nseg, s0 + DEC_SEG);
    float s = 0.f;
#pragma unroll 8
    for (int k = s0 + q; k < s1; k += 4) s += src[(size_t)k * DEC_NF + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0) part2[((size_t)b * ngrp + grp) * DEC_NF + c] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
    __threadfence();   // this workgroup's part2 row is visible device-wide before its arrival is
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(count + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(ngrp - 1)) return;
    __threadfence();   // every other workgroup's row is visible here
    const float* p2 = part2 + (size_t)b * ngrp * DEC_NF;
    s = 0.f;
#pragma unroll 4
    for (int k = q; k < ngrp; k += 4) s += __builtin_nontemporal_load(p2 + (size_t)k * DEC_NF + c);
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0) mean[c] = ((red[c] + red[64 + c]) + (red[128 + c] + red[192 + c])) * inv_hw;
    __syncthreads();
    if (threadIdx.x < DEC_SE_R) {
        float acc = 0.f;
        for (int k = 0; k < DEC_NF; ++k) acc += w1[threadIdx.x * DEC_NF + k] * mean[k];
        hid[threadIdx.x] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (q == 0) {
        float acc = 0.f;
        for (int r = 0; r < DEC_SE_R; ++r) acc += w2[c * DEC_SE_R + r] * hid[r];
        gate[(size_t)b * DEC_NF + c] = 1.f / (1.f + expf(-acc));
    }
    if (threadIdx.x == 0) count[b] = 0u;
}

// upscale_factor 4: the dense blocks' output with the last squeeze-excitation gate and the global residual applied,
// x = P + T * gate + S (decoder_rdn.py:40, :78), as its own (tiny: the b = 4 bundle map is a sixteenth of the image) element-wise pass -
// the four sub-pixel convolutions of the first up stage all read it.  Rounded as the fused apply of the bundle_size 2 path rounds.
__global__ void __launch_bounds__(256) k_apply_gate(const float* __restrict__ P, const float* __restrict__ T, const float* __restrict__ gate,
                                                    const float* __restrict__ S, float* __restrict__ X, size_t npix_per_item, int B) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one float4 of 64 channels: 16 per pixel
    if (i >= (size_t)B * npix_per_item * 16) return;
    const size_t pix = i >> 4; const int c4 = (int)(i & 15) * 4;
    const int b = (int)(pix / npix_per_item);
    const F4 p = *(const F4*)(P + pix * DEC_NF + c4), t = *(const F4*)(T + pix * DEC_NF + c4), g = *(const F4*)(gate + (size_t)b * DEC_NF + c4);
    const F4 sres = *(const F4*)(S + pix * DEC_NF + c4);
    F4 v = p + t * g;
    v = v + sres;
    *(F4*)(X + pix * DEC_NF + c4) = v;
}

// ---- entry ---------------------------------------------------------------------------------------------------------------
static hipError_t launch_conv(const ConvArgs& a, hipStream_t st) {   // 64 output channels
    const size_t lds = sizeof(float) * (size_t)(2 + 2) * DEC_PX * DEC_CHS;
    const dim3 grid((unsigned)(a.B * a.tilesX * a.tilesY));
    if (a.se_part) hipLaunchKernelGGL((k_conv16<2, true, false, 2, true>), grid, dim3(256), lds, st, a);     // conv3
    else if (a.vec) hipLaunchKernelGGL((k_conv16<2, true, false, 2, false>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_conv16<2, false, false, 2, false>), grid, dim3(256), lds, st, a);             // in_conv
    return hipGetLastError();
}

template <int NR>
static hipError_t launch_conv16(const ConvArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(2 + 2) * DEC_PX * DEC_CHS;
    const dim3 grid((unsigned)(a.B * a.tilesX * a.tilesY));
    if (a.fT) hipLaunchKernelGGL((k_conv16<NR, true, true>), grid, dim3(256), lds, st, a);
    else if (a.vec) hipLaunchKernelGGL((k_conv16<NR, true, false>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_conv16<NR, false, false>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

static hipError_t launch_convx(const ConvArgs& a, int nt, hipStream_t st) {
    const size_t lds = sizeof(unsigned) * ((size_t)(4 * DECX_ROWS + 2) * DEC_PX * DECX_PXD + 9 * 2 * 64 * 4);
    const dim3 grid((unsigned)(a.B * a.tilesX * a.tilesY * nt));
    if (a.fT) hipLaunchKernelGGL((k_conv3x3x<true, true, false>), grid, dim3(256), lds, st, a);
    else if (a.se_part) hipLaunchKernelGGL((k_conv3x3x<true, false, true>), grid, dim3(256), lds, st, a);
    else if (a.vec) hipLaunchKernelGGL((k_conv3x3x<true, false, false>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_conv3x3x<false, false, false>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

extern "C" int gdb_decode(const GdbConfig* cfg, const GdbFrame* shape, const float* d_bundle_feat, int32_t ld_bundle_feat,
                          const float* d_packed, int32_t num_layers, int32_t precision, void* d_ws, size_t ws_bytes, float* d_rgb_c,
                          void* stream_) {
    int rc = dec_check(cfg, num_layers); if (rc) return rc;
    if (precision != GDB_PREC_F32 && precision != GDB_PREC_F32X)
        return gdb_fail(GDB_E_BADARG, "decoder precision %d unsupported (1 = fp32 MFMA, 2 = split-f16 operand pairs)", precision);
    const bool split = precision == GDB_PREC_F32X;
    if (!shape || !d_bundle_feat || !d_packed || !d_ws || !d_rgb_c) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    const int B = shape->B, H = shape->H, W = shape->W;
    if (B < 1 || H < 1 || W < 1) return gdb_fail(GDB_E_SHAPE, "non-positive bundle map");
    if (B > 256) return gdb_fail(GDB_E_SHAPE, "decoder batch %d > 256", B);
    const int Q = 3 * cfg->bundle_size * cfg->bundle_size + GDB_CFR + GDB_CV, n_rgb = 3 * cfg->bundle_size * cfg->bundle_size;
    if (ld_bundle_feat < Q) return gdb_fail(GDB_E_SHAPE, "bundle_feat row stride %d < %d channels", ld_bundle_feat, Q);
    const bool up4 = cfg->bundle_size == 4;
    const DecWs ws = dec_ws(B, H, W, cfg->bundle_size);
    if (ws_bytes < ws.total) return gdb_fail(GDB_E_WORKSPACE, "decoder workspace %zu B < required %zu B", ws_bytes, ws.total);
    if ((size_t)H * W * (up4 ? 4 : 1) * (size_t)(ld_bundle_feat > DEC_NF ? ld_bundle_feat : DEC_NF) >= ((size_t)1 << 30))
        return gdb_fail(GDB_E_SHAPE, "bundle map too large for the decoder's 32-bit staging byte offsets");
    const DecLayout L = dec_layout(num_layers, cfg->bundle_size);
    hipStream_t st = (hipStream_t)stream_;
    float* Pbuf[3] = {(float*)((char*)d_ws + ws.P[0]), (float*)((char*)d_ws + ws.P[1]), (float*)((char*)d_ws + ws.P[2])};
    auto Px = [&](int b) -> float* { return b == 0 ? Pbuf[0] : Pbuf[1 + ((b - 1) & 1)]; };   // the input x of dense block b
    float* Y = (float*)((char*)d_ws + ws.Y); float* T = (float*)((char*)d_ws + ws.T);
    float* part = (float*)((char*)d_ws + ws.part); float* part2 = (float*)((char*)d_ws + ws.part2);
    float* gate = (float*)((char*)d_ws + ws.gate); unsigned* count = (unsigned*)((char*)d_ws + ws.count);
    static std::atomic<unsigned long long> attr_done{0};
    if (split) {   // per device, once: the split kernel's workgroup may take more than the default 64 KiB of dynamic LDS
        const size_t lds = sizeof(unsigned) * ((size_t)(4 * DECX_ROWS + 2) * DEC_PX * DECX_PXD + 9 * 2 * 64 * 4);
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
        if (lds > 64 * 1024 && !(attr_done.load() >> dev & 1)) {
            const void* fns[] = {(const void*)k_conv3x3x<true, true, false>, (const void*)k_conv3x3x<true, false, true>,
                                 (const void*)k_conv3x3x<true, false, false>, (const void*)k_conv3x3x<false, false, false>};
            for (const void* fn : fns) {
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            }
            attr_done.fetch_or(1ull << dev);
        }
    }
    int cH = H, cW = W;   // the map the next convolution runs on ((2H, 2W) for the last stage of upscale_factor 4)
    auto conv = [&](ConvArgs a, int nt) -> hipError_t {
        const int H = cH, W = cW;
        a.B = B; a.H = H; a.W = W; a.tilesX = (W + 31) / 32;
        a.nchunk = (a.cin + 31) / 32;
        if (!a.in2) a.split = a.nchunk;
        a.vec = (a.in_stride % 4 == 0) && (a.in_off % 4 == 0) && (a.cin % 4 == 0) && ((uintptr_t)a.in % 16 == 0);
        if (split) {
            a.tilesY = (H + 4 * DECX_ROWS - 1) / (4 * DECX_ROWS);
            return launch_convx(a, nt, st);
        }
        a.tilesY = (H + 1) / 2;   // every fp32 kernel: 2 rows x 32 pixels per workgroup
        if (nt == 2) return launch_conv(a, st);
        return a.cout > 16 ? launch_conv16<2>(a, st) : launch_conv16<1>(a, st);
    };
    hipError_t e;
#define CK(x) do { e = (x); if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "decoder launch: %s", hipGetErrorString(e)); } while (0)
    {   // shallow = in_conv(bundle channels n_rgb..Q-1)   decoder_rdn.py:76: block 0's x, and the global residual at the end
        ConvArgs a{};
        a.in = d_bundle_feat; a.in_stride = ld_bundle_feat; a.in_off = n_rgb; a.cin = Q - n_rgb;
        a.w = d_packed + (split ? L.in_wx : L.in_w); a.bias = d_packed + L.in_b;
        a.out = Px(0); a.out_stride = DEC_NF; a.out_off = 0; a.cout = DEC_NF; a.relu = 0; a.zero = count;
        CK(conv(a, 2));
    }
    for (int b = 0; b < num_layers; ++b) {   // ResidualDenseBlock.forward   decoder_rdn.py:35-41
        ConvArgs a{};
        a.in_stride = DEC_NF; a.in_off = 0; a.out = Y; a.out_stride = DEC_NF; a.relu = 1;
        // conv1; from the second block on it first forms its own input x_b = x_{b-1} + x3_{b-1} * gate_{b-1} (:40) and leaves it in P[b]
        a.in = b ? Px(b - 1) : Px(0);
        if (b) { a.fT = T; a.fgate = gate; a.fX = Px(b); }
        a.cin = DEC_NF; a.w = d_packed + (split ? L.blkx[b][0] : L.blk[b][0]); a.out_off = 0; a.cout = DEC_G;
        CK(conv(a, 1));
        a.fT = nullptr; a.fgate = nullptr; a.fX = nullptr;
        a.in = Px(b); a.in2 = Y; a.split = 2;
        a.cin = DEC_NF + DEC_G; a.w = d_packed + (split ? L.blkx[b][1] : L.blk[b][1]); a.out_off = DEC_G;
        CK(conv(a, 1));
        a.cin = DEC_NF + 2 * DEC_G; a.w = d_packed + (split ? L.blkx[b][2] : L.blk[b][2]); a.out = T; a.out_off = 0; a.cout = DEC_NF; a.relu = 0;
        a.se_part = part;
        CK(conv(a, 2));
        hipLaunchKernelGGL(k_se_gate, dim3((unsigned)(B * ws.ngrp)), dim3(256), 0, st, part, ws.nseg, ws.ngrp, 1.f / (float)((size_t)H * W),
                           d_packed + L.blk[b][3], d_packed + L.blk[b][4], part2, count, gate);
        CK(hipGetLastError());
    }
    if (!up4) {   // out_conv(PixelShuffle(up(x + shallow))) as one folded 64 -> 12 convolution on x = x_{L-1} + x3 * gate + shallow   :40,78-80
        ConvArgs a{};
        a.in = Px(num_layers - 1); a.in_stride = DEC_NF; a.in_off = 0; a.cin = DEC_NF;
        a.fT = T; a.fgate = gate; a.fS = Px(0);
        a.w = d_packed + (split ? L.up_wx : L.up_w); a.bias = d_packed + L.up_b;
        a.cout = 12; a.relu = 0; a.rgb = d_rgb_c;
        CK(conv(a, 1));
    } else {
        // upscale_factor 4 (bundle_size 4; decoder_rdn.py:59-62): x = x_{L-1} + x3 * gate + shallow as its own pass, the first up stage as
        // four 64 -> 64 convolutions (one per sub-pixel of its PixelShuffle) into the (2H, 2W, 64) map U, then the second stage folded with
        // out_conv as ONE 64 -> 12 convolution on U, written pixel-shuffled to the (4H, 4W) image - no non-linearity anywhere in between.
        float* X = (float*)((char*)d_ws + ws.X); float* U = (float*)((char*)d_ws + ws.U);
        const size_t npix = (size_t)H * W;
        hipLaunchKernelGGL(k_apply_gate, dim3((unsigned)(((size_t)B * npix * 16 + 255) / 256)), dim3(256), 0, st, Px(num_layers - 1), T, gate, Px(0), X, npix, B);
        CK(hipGetLastError());
        for (int sp = 0; sp < 4; ++sp) {
            ConvArgs a{};
            a.in = X; a.in_stride = DEC_NF; a.in_off = 0; a.cin = DEC_NF;
            a.w = d_packed + (split ? L.u1_wx[sp] : L.u1_w[sp]); a.bias = d_packed + L.u1_b[sp];
            a.out = U; a.out_stride = DEC_NF; a.out_off = 0; a.cout = DEC_NF; a.relu = 0;
            a.up2 = 1; a.up_dy = sp >> 1; a.up_dx = sp & 1;
            CK(conv(a, 2));
        }
        cH = 2 * H; cW = 2 * W;
        ConvArgs a{};
        a.in = U; a.in_stride = DEC_NF; a.in_off = 0; a.cin = DEC_NF;
        a.w = d_packed + (split ? L.up_wx : L.up_w); a.bias = d_packed + L.up_b;
        a.cout = 12; a.relu = 0; a.rgb = d_rgb_c;
        CK(conv(a, 1));
    }
#undef CK
    return GDB_OK;
}
