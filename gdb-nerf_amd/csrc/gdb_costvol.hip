// "Next" rows of SURVEY.md §8(f) on the same bar as the hot path:
//   N2  build_feature_volume  (reference networks/gdb_nerf/depth_net.py:424-476): plane-sweep homography
//       warp of the source feature maps onto the target frustum + biased variance over views;
//   N4  depth_regression      (depth_net.py:479-514): soft-argmax depth and confidence interval.
// Exact fp32 (-ffp-contract=off), gather-bound: one lane per voxel column x, so the NCHW source maps are
// read along x (8-byte x-pair loads) and the NCDHW volume is written in 256-B row segments.
#include "gdb_internal.h"
#include <cstdlib>

int gdb_fail(int code, const char* fmt, ...);

#define LAUNCH_CHECK(name)                                                                    \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return gdb_fail(GDB_E_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// pixel(target) -> pixel(source) map of every (batch, view): P_src * inverse([P_tar; 0 0 0 1])   :449-453
__device__ __forceinline__ void inv4_rowmajor_f64(const double* m, double* o) {
    double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    double inv = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    o[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * inv;   o[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * inv;
    o[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * inv; o[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * inv;
    o[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * inv;  o[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * inv;
    o[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * inv; o[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * inv;
    o[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * inv;   o[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * inv;
    o[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * inv; o[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * inv;
    o[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * inv; o[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * inv;
    o[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * inv; o[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * inv;
}

__global__ void k_costvol_proj(int B, int V, const float* __restrict__ src_exts, const float* __restrict__ src_ints,
                               const float* __restrict__ tar_exts, const float* __restrict__ tar_ints, float* __restrict__ proj) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * V) return;
    int b = t / V;
    double Pt[16], Pti[16], Ps[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f, q = 0.f;  // the reference multiplies fp32 tensors: round the 3x4 products to fp32 first
            for (int k = 0; k < 3; ++k) {
                s += tar_ints[b * 9 + i * 3 + k] * tar_exts[b * 16 + k * 4 + j];
                q += src_ints[(size_t)t * 9 + i * 3 + k] * src_exts[(size_t)t * 16 + k * 4 + j];
            }
            Pt[i * 4 + j] = s; Ps[i * 4 + j] = q;
        }
    Pt[12] = 0; Pt[13] = 0; Pt[14] = 0; Pt[15] = 1;
    inv4_rowmajor_f64(Pt, Pti);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Ps[i * 4 + k] * (double)(float)Pti[k * 4 + j];
            proj[(size_t)t * 12 + i * 4 + j] = (float)s;
        }
}

struct F2c { float x, y; } __attribute__((packed, aligned(4)));

struct CostVolArgs {
    int B, V, C, Hs, Ws, D, Ht, Wt, inv_depth, cpt, tiles, nblk;  // cpt: channels per thread
    const float* feat; const float* proj; const float* depth_values; float* out;
};

// Channel-pair re-layout of the source maps for the PAIR form of k_costvol: (B*V, C, Hs, Ws) -> (B*V, C / 2, Hs, Ws, 2), so that one
// 16-byte load at (y, x) holds the x pair of TWO channels.  One thread per (map, channel pair, y, x): two coalesced 4-byte loads, one
// 8-byte store.  15.7 MB at the 256x320 stage: ~5 us, a third of what it saves the sweep (profiles/r04/costvol_pair_layout.txt).
__global__ void __launch_bounds__(256) k_costvol_pairs(const float* __restrict__ src, float2* __restrict__ dst, size_t plane, size_t npairs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs * plane) return;
    const size_t pc = t / plane, p = t - pc * plane;   // pc = map * (C / 2) + channel pair
    dst[t] = make_float2(src[(2 * pc) * plane + p], src[(2 * pc + 1) * plane + p]);
}

// PAIR: a.feat is the channel-pair-interleaved copy (k_costvol_pairs; C even): per (row, view) ONE 16-byte load serves two channels.
template <int VT, bool PAIR>  // VT = number of source views: the per-view tap state lives in registers
__global__ void __launch_bounds__(256) k_costvol(CostVolArgs a) {
    // 1-D grid over (batch, tile of 256 voxels of the flattened (y,x) plane, depth plane, channel group), depth
    // innermost, remapped so that each XCD (blocks b, b+8, ...) walks one contiguous band: the D planes of a tile
    // and the neighbouring tiles re-read the same source rows out of that XCD's L2.  (With depth as a grid
    // dimension the 16 MB of source maps were fetched from HBM ~16x: 383 MB FETCH_SIZE at the 256x320 stage.)
    const int groups = (a.C + a.cpt - 1) / a.cpt;
    const int chunk = (a.nblk + 7) >> 3;
    int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (lb >= a.nblk) return;
    const int g = lb % groups; lb /= groups;
    const int d = lb % a.D; lb /= a.D;
    const int tile = lb % a.tiles, b = lb / a.tiles;
    const int c_begin = g * a.cpt, c_end = min(c_begin + a.cpt, a.C);
    const int t = tile * blockDim.x + threadIdx.x;  // the (y,x) plane flattened: no ragged-row waste
    if (t >= a.Ht * a.Wt) return;
    const int y = t / a.Wt, x = t - y * a.Wt;
    const size_t vox = ((size_t)d * a.Ht + y) * a.Wt + x;
    float depth = a.depth_values[(size_t)b * a.D * a.Ht * a.Wt + vox];
    if (a.inv_depth) depth = 1.f / depth;                                                     // :445-446
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    // per view: two clamped row offsets and the four tap weights of the x pair (zeros padding = weight 0)
    unsigned off0[VT], off1[VT];
    float w00[VT], w01[VT], w10[VT], w11[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        {
            const float* P = a.proj + ((size_t)b * a.V + v) * 12;
            float p[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = (P[4 * r] * px + P[4 * r + 1] * py + P[4 * r + 2]) * depth + P[4 * r + 3];  // :466
            float z = fmaxf(p[2], 1e-6f);
            float gx = 2.f * (p[0] / z) / (float)a.Ws - 1.f, gy = 2.f * (p[1] / z) / (float)a.Hs - 1.f;   // :467-468
            float ix = ((gx + 1.f) * (float)a.Ws - 1.f) / 2.f, iy = ((gy + 1.f) * (float)a.Hs - 1.f) / 2.f;
            // keep far-away coordinates representable as int (they are outside anyway)
            ix = fminf(fmaxf(ix, -4.f), (float)a.Ws + 4.f); iy = fminf(fmaxf(iy, -4.f), (float)a.Hs + 4.f);
            bool finite = (p[0] == p[0]) && (p[1] == p[1]) && (p[2] == p[2]);
            float xf = floorf(ix), yf = floorf(iy);
            float fx = ix - xf, fy = iy - yf;
            int x0 = (int)xf, y0 = (int)yf;
            // x pair (xs, xs+1) covers columns x0, x0+1 where they exist
            int xs = min(max(x0, 0), a.Ws - 2);
            float ea = (x0 >= 0 && x0 <= a.Ws - 1) ? 1.f - fx : 0.f, eb = (x0 + 1 >= 0 && x0 + 1 <= a.Ws - 1) ? fx : 0.f;
            float e0 = (x0 == xs ? ea : 0.f) + (x0 + 1 == xs ? eb : 0.f);
            float e1 = (x0 == xs + 1 ? ea : 0.f) + (x0 + 1 == xs + 1 ? eb : 0.f);
            float ra = (y0 >= 0 && y0 <= a.Hs - 1) ? 1.f - fy : 0.f, rb = (y0 + 1 >= 0 && y0 + 1 <= a.Hs - 1) ? fy : 0.f;
            if (!finite) { e0 = e1 = 0.f; }
            int ya = min(max(y0, 0), a.Hs - 1), yb = min(max(y0 + 1, 0), a.Hs - 1);
            off0[v] = (unsigned)(ya * a.Ws + xs); off1[v] = (unsigned)(yb * a.Ws + xs);
            w00[v] = e0 * ra; w01[v] = e1 * ra; w10[v] = e0 * rb; w11[v] = e1 * rb;
        }
    }
    const size_t plane = (size_t)a.Hs * a.Ws, ovol = (size_t)a.D * a.Ht * a.Wt;
    const float invV = 1.f / (float)a.V;
    if constexpr (PAIR) {
        // [c / 2][y][x][2]: the 16 bytes at (y, xs) are (c @ xs, c + 1 @ xs, c @ xs + 1, c + 1 @ xs + 1); same products, same order of
        // sums as the planar form below: bit-identical results
        struct F4c { float x, y, z, w; } __attribute__((packed, aligned(8)));
        for (int c = c_begin; c < c_end; c += 2) {
            float val[2][VT], mean[2] = {0.f, 0.f};
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const float* pl = a.feat + (((size_t)b * a.V + v) * a.C + c) * plane;   // start of the pair's interleaved plane (2 plane floats)
                const F4c t0 = *(const F4c*)(pl + 2 * (size_t)off0[v]), t1 = *(const F4c*)(pl + 2 * (size_t)off1[v]);
                val[0][v] = t0.x * w00[v] + t0.z * w01[v] + t1.x * w10[v] + t1.z * w11[v];
                val[1][v] = t0.y * w00[v] + t0.w * w01[v] + t1.y * w10[v] + t1.w * w11[v];
                mean[0] += val[0][v]; mean[1] += val[1][v];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float m = mean[e] * invV;
                float var = 0.f;
#pragma unroll
                for (int v = 0; v < VT; ++v) { float d = val[e][v] - m; var += d * d; }
                a.out[((size_t)b * a.C + c + e) * ovol + vox] = var * invV;
            }
        }
        return;
    }
    for (int c = c_begin; c < c_end; ++c) {
        float val[VT], mean = 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            {
                const float* pl = a.feat + (((size_t)b * a.V + v) * a.C + c) * plane;
#ifndef GDB_XP_CV_SCALAR   // one 8-byte load per x pair (any 4-byte alignment costs the same 16 TA cycles per wave instruction:
                           // tools/ubench/ta_rate.hip); two dword loads per pair measured slower: 115 vs 72 us at the 256x320 stage
                F2c t0 = *(const F2c*)(pl + off0[v]), t1 = *(const F2c*)(pl + off1[v]);
                val[v] = t0.x * w00[v] + t0.y * w01[v] + t1.x * w10[v] + t1.y * w11[v];      // :472
#else
                const float a0 = pl[off0[v]], a1 = pl[off0[v] + 1], b0 = pl[off1[v]], b1 = pl[off1[v] + 1];
                val[v] = a0 * w00[v] + a1 * w01[v] + b0 * w10[v] + b1 * w11[v];              // :472
#endif
                mean += val[v];
            }
        }
        mean *= invV;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v) { float e = val[v] - mean; var += e * e; }
        a.out[((size_t)b * a.C + c) * ovol + vox] = var * invV;                               // :474 (unbiased=False)
    }
}

static int build_feature_volume(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                                const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                                int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                                int32_t inv_depth, float* d_proj_ws, float* d_pair_ws, float* d_out, void* stream_);

extern "C" int gdb_build_feature_volume(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                                        const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                                        int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                                        int32_t inv_depth, float* d_proj_ws, float* d_out, void* stream_) {
    return build_feature_volume(d_src_feat, d_src_exts, d_src_ints, d_tar_exts, d_tar_ints, d_depth_values, B, V, C, Hs, Ws, D, Ht, Wt, inv_depth,
                                d_proj_ws, nullptr, d_out, stream_);
}

extern "C" int gdb_build_feature_volume_ws(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                                           const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                                           int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                                           int32_t inv_depth, float* d_proj_ws, float* d_pair_ws, float* d_out, void* stream_) {
    return build_feature_volume(d_src_feat, d_src_exts, d_src_ints, d_tar_exts, d_tar_ints, d_depth_values, B, V, C, Hs, Ws, D, Ht, Wt, inv_depth,
                                d_proj_ws, d_pair_ws, d_out, stream_);
}

static int build_feature_volume(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                                const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                                int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                                int32_t inv_depth, float* d_proj_ws, float* d_pair_ws, float* d_out, void* stream_) {
    if (!d_src_feat || !d_src_exts || !d_src_ints || !d_tar_exts || !d_tar_ints || !d_depth_values || !d_proj_ws || !d_out)
        return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (B < 1 || V < 1 || C < 1 || Hs < 1 || Ws < 2 || D < 1 || Ht < 1 || Wt < 1) return gdb_fail(GDB_E_SHAPE, "bad cost-volume shape");
    if (V > GDB_MAX_VIEWS) return gdb_fail(GDB_E_SHAPE, "V=%d exceeds %d views", V, GDB_MAX_VIEWS);
    if ((size_t)C * Hs * Ws >= ((size_t)1 << 32)) return gdb_fail(GDB_E_SHAPE, "source feature map too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_costvol_proj, dim3((B * V + 63) / 64), dim3(64), 0, st, B, V, d_src_exts, d_src_ints, d_tar_exts, d_tar_ints, d_proj_ws);
    LAUNCH_CHECK("k_costvol_proj");
    // channels per thread: all of them (splitting channels over more threads measured no faster on MI355X:
    // 44 / 118 us at the two DTU stage shapes for cpt = 32, 8, 4); GDB_COSTVOL_CPT overrides in the diagnostic build
    int cpt = C;
#ifdef GDB_DIAG  // diagnostic build only: the product entry reads no environment
    if (getenv("GDB_COSTVOL_CPT")) cpt = atoi(getenv("GDB_COSTVOL_CPT")) > 0 ? atoi(getenv("GDB_COSTVOL_CPT")) : C;
#endif
    const int tiles = (Ht * Wt + 255) / 256, groups = (C + cpt - 1) / cpt;
    if ((size_t)B * tiles * D * groups >= ((size_t)1 << 31)) return gdb_fail(GDB_E_SHAPE, "cost volume too large for the launch grid");
    const int nblk = B * tiles * D * groups;
    // The channel-pair form: given C * Hs * Ws * V * B floats of scratch (gdb_build_feature_volume_ws) and an even channel count, the
    // source maps are re-laid once ([c / 2][y][x][2], k_costvol_pairs) and the sweep loads 16 bytes per (channel PAIR, row, view)
    // instead of 8 per (channel, row, view): half the load instructions of a kernel the texture addresser bounds.  Bit-identical.
    const bool pair = d_pair_ws != nullptr && (C % 2) == 0 && (cpt % 2) == 0 && V <= 4;   // (5..8 views: the doubled tap state leaves the registers)
    if (pair) {
        const size_t plane = (size_t)Hs * Ws, npairs = (size_t)B * V * (C / 2);
        hipLaunchKernelGGL(k_costvol_pairs, dim3((unsigned)((npairs * plane + 255) / 256)), dim3(256), 0, st, d_src_feat, (float2*)d_pair_ws, plane, npairs);
        LAUNCH_CHECK("k_costvol_pairs");
    }
    CostVolArgs a{B, V, C, Hs, Ws, D, Ht, Wt, inv_depth, cpt, tiles, nblk, pair ? d_pair_ws : d_src_feat, d_proj_ws, d_depth_values, d_out};
    const dim3 grid((nblk + 7) / 8 * 8), blk(256);
    if (pair) {
        switch (V) {
            case 1: hipLaunchKernelGGL((k_costvol<1, true>), grid, blk, 0, st, a); break;
            case 2: hipLaunchKernelGGL((k_costvol<2, true>), grid, blk, 0, st, a); break;
            case 3: hipLaunchKernelGGL((k_costvol<3, true>), grid, blk, 0, st, a); break;
            default: hipLaunchKernelGGL((k_costvol<4, true>), grid, blk, 0, st, a); break;
        }
    } else {
        switch (V) {
            case 1: hipLaunchKernelGGL((k_costvol<1, false>), grid, blk, 0, st, a); break;
            case 2: hipLaunchKernelGGL((k_costvol<2, false>), grid, blk, 0, st, a); break;
            case 3: hipLaunchKernelGGL((k_costvol<3, false>), grid, blk, 0, st, a); break;
            case 4: hipLaunchKernelGGL((k_costvol<4, false>), grid, blk, 0, st, a); break;
            case 5: hipLaunchKernelGGL((k_costvol<5, false>), grid, blk, 0, st, a); break;
            case 6: hipLaunchKernelGGL((k_costvol<6, false>), grid, blk, 0, st, a); break;
            case 7: hipLaunchKernelGGL((k_costvol<7, false>), grid, blk, 0, st, a); break;
            default: hipLaunchKernelGGL((k_costvol<8, false>), grid, blk, 0, st, a); break;
        }
    }
    LAUNCH_CHECK("k_costvol");
    return GDB_OK;
}

// ---- N4  depth_regression ---------------------------------------------------------------------
__global__ void k_depth_regression(int B, int D, size_t HW, float ci_scale, int inv_depth, const float* __restrict__ dv,
                                   const float* __restrict__ prob, float* __restrict__ depth, float* __restrict__ ci) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * HW) return;
    size_t b = t / HW, p = t - b * HW;
    const float* dvb = dv + b * D * HW + p; const float* pb = prob + b * D * HW + p;
    float mean = 0.f;
    for (int d = 0; d < D; ++d) mean += pb[d * HW] * dvb[d * HW];                          // :495
    float var = 0.f;
    for (int d = 0; d < D; ++d) { float e = dvb[d * HW] - mean; var += pb[d * HW] * (e * e); }  // :496
    float half = ci_scale * sqrtf(fmaxf(var, 1e-12f));                                       // :497-500
    float first = dvb[0], last = dvb[(size_t)(D - 1) * HW];
    if (inv_depth) {                                                                          // :502-507
        ci[(b * 2) * HW + p] = 1.f / fminf(mean + half, first);
        ci[(b * 2 + 1) * HW + p] = 1.f / fmaxf(mean - half, last);
        depth[t] = 1.f / mean;
    } else {                                                                                  // :508-512
        ci[(b * 2) * HW + p] = fmaxf(mean - half, first);
        ci[(b * 2 + 1) * HW + p] = fminf(mean + half, last);
        depth[t] = mean;
    }
}

extern "C" int gdb_depth_regression(const float* d_depth_values, const float* d_depth_prob, int32_t B, int32_t D, int32_t H, int32_t W,
                                    float ci_scale, int32_t inv_depth, float* d_depth, float* d_ci, void* stream_) {
    if (!d_depth_values || !d_depth_prob || !d_depth || !d_ci) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (B < 1 || D < 1 || H < 1 || W < 1) return gdb_fail(GDB_E_SHAPE, "bad shape");
    size_t HW = (size_t)H * W, n = (size_t)B * HW;
    hipLaunchKernelGGL(k_depth_regression, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, B, D, HW, ci_scale,
                       inv_depth, d_depth_values, d_depth_prob, d_depth, d_ci);
    LAUNCH_CHECK("k_depth_regression");
    return GDB_OK;
}
