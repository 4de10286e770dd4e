// Operator mirrors of the GDB-NeRF hot path for gfx950: per-frame preparation (camera block,
// feature mip pyramid) and one kernel family per reference method (build_rays, sample,
// encode, NeRF.forward, composite).  These materialise the same intermediates as the
// reference and exist for drop-in use of the individual operators and for bisecting the
// fused kernel (gdb_fused.hip), which is the production path.
//
// Compiled with -ffp-contract=off: the reference evaluates every torch op separately, so
// a*b+c is two roundings unless written as fmaf() here on purpose.
#include "gdb_internal.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdarg>

// ============================================================================================
// error plumbing
// ============================================================================================
static thread_local char g_err[512] = "";

int gdb_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return gdb_fail(GDB_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCH_CHECK(name)                                                                    \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return gdb_fail(GDB_E_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

extern "C" int gdb_abi_version(void) { return GDB_ABI_VERSION; }
extern "C" const char* gdb_last_error(void) { return g_err; }

int gdb_check_cfg(const GdbConfig* c) {
    if (!c) return gdb_fail(GDB_E_BADARG, "cfg is NULL");
    int b = c->bundle_size;
    // network.py:33-34 raises ValueError('`Bundle size` must be a power of 2.')
    if (b <= 0 || (b & (b - 1)) != 0) return gdb_fail(GDB_E_BADARG, "`Bundle size` must be a power of 2.");
    if (b > 4) return gdb_fail(GDB_E_BADARG, "bundle_size %d unsupported (1, 2 or 4)", b);
    if (c->max_num_samples < 1 || c->max_num_samples > GDB_MAX_SAMPLES)
        return gdb_fail(GDB_E_BADARG, "max_num_samples %d outside 1..%d", c->max_num_samples, GDB_MAX_SAMPLES);
    if (c->max_mipmap_level < 0 || c->max_mipmap_level > GDB_MAX_MIP)
        return gdb_fail(GDB_E_BADARG, "max_mipmap_level %d outside 0..%d", c->max_mipmap_level, GDB_MAX_MIP);
    if (c->global_num_depth < 1) return gdb_fail(GDB_E_BADARG, "global_num_depth must be positive");
    if (c->feat_dim != GDB_CF || c->voxel_dim != GDB_CV || c->hid_dim != GDB_HID)
        return gdb_fail(GDB_E_BADARG, "kernels are built for feat_dim %d, voxel_dim %d, nerf_hidden_dims %d (got %d, %d, %d)",
                        GDB_CF, GDB_CV, GDB_HID, c->feat_dim, c->voxel_dim, c->hid_dim);
    return GDB_OK;
}

int gdb_check_frame(const GdbConfig* c, const GdbFrame* f, bool need_ptrs) {
    if (!f) return gdb_fail(GDB_E_BADARG, "frame is NULL");
    if (f->B < 1 || f->V < 1 || f->Ho < 1 || f->Wo < 1 || f->D < 1) return gdb_fail(GDB_E_SHAPE, "non-positive frame size");
    if (f->V > GDB_MAX_VIEWS) return gdb_fail(GDB_E_SHAPE, "V=%d exceeds %d views", f->V, GDB_MAX_VIEWS);
    if (f->Ho % c->bundle_size || f->Wo % c->bundle_size)
        return gdb_fail(GDB_E_SHAPE, "image %dx%d not divisible by bundle_size %d", f->Ho, f->Wo, c->bundle_size);
    if (f->H != f->Ho / c->bundle_size || f->W != f->Wo / c->bundle_size)
        return gdb_fail(GDB_E_SHAPE, "bundle map %dx%d != image/bundle_size %dx%d", f->H, f->W, f->Ho / c->bundle_size, f->Wo / c->bundle_size);
    if ((size_t)f->B * f->H * f->W * c->max_num_samples >= (size_t)1 << 31)
        return gdb_fail(GDB_E_SHAPE, "more than 2^31 sample slots");
    // (d_img_feat is read by gdb_prepare only: afterwards the feature pyramid in the workspace stands for it)
    if (need_ptrs && (!f->d_src_images || !f->d_feat_volume || !f->d_depth_range || !f->d_vol_range ||
                      !f->d_src_exts || !f->d_src_ints || !f->d_tar_exts || !f->d_tar_ints || !f->d_near_far))
        return gdb_fail(GDB_E_BADARG, "frame has a NULL device pointer");
    return GDB_OK;
}

extern "C" int gdb_workspace_bytes(const GdbConfig* cfg, const GdbFrame* shape, size_t* out_bytes) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!out_bytes) return gdb_fail(GDB_E_BADARG, "out_bytes is NULL");
    *out_bytes = ws_layout(*cfg, *shape).total;
    return GDB_OK;
}

extern "C" int gdb_dense_plan_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[3]) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!out) return gdb_fail(GDB_E_BADARG, "out is NULL");
    const WsLayout L = ws_layout(*cfg, *shape);
    out[0] = L.planOff; out[1] = (size_t)L.planMW + 2; out[2] = (size_t)L.planL;
    return GDB_OK;
}

extern "C" int gdb_dense_map_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[2]) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!out) return gdb_fail(GDB_E_BADARG, "out is NULL");
    const WsLayout L = ws_layout(*cfg, *shape);
    out[0] = L.smapOff; out[1] = (size_t)L.smapStride;
    return GDB_OK;
}

extern "C" int gdb_pyramid_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[7]) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!out) return gdb_fail(GDB_E_BADARG, "out is NULL");
    const WsLayout L = ws_layout(*cfg, *shape);
    out[0] = L.pyrOff; out[1] = L.pyrStride; out[2] = (size_t)L.levels;
    for (int l = 0; l <= GDB_MAX_MIP; ++l) out[3 + l] = l <= L.levels ? L.lvlOff[l] : 0;
    return GDB_OK;
}

extern "C" int gdb_pyramid16_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[7]) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!out) return gdb_fail(GDB_E_BADARG, "out is NULL");
    const WsLayout L = ws_layout(*cfg, *shape);
    out[0] = L.pyr16Off; out[1] = 2 * L.pyrStride; out[2] = (size_t)L.levels;   // byte offset, bytes per (batch, view), levels
    for (int l = 0; l <= GDB_MAX_MIP; ++l) out[3 + l] = l <= L.levels ? 2 * L.lvlOff[l] : 0;  // byte offset of each level
    return GDB_OK;
}

// ============================================================================================
// camera block
// ============================================================================================
// Closed-form fp64 inverses, fully in registers (a Gauss-Jordan with indexed scratch arrays made this
// single thread the critical path of k_prepare).  4x4 by 2x2 sub-determinants, 3x3 by cofactors.
__device__ __forceinline__ void invert4_f64(const double* m, double* o) {
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
__device__ __forceinline__ void invert3_f64(const double* m, double* o) {
    double a = m[4] * m[8] - m[5] * m[7], b = m[5] * m[6] - m[3] * m[8], c = m[3] * m[7] - m[4] * m[6];
    double inv = 1.0 / (m[0] * a + m[1] * b + m[2] * c);
    o[0] = a * inv; o[1] = (m[2] * m[7] - m[1] * m[8]) * inv; o[2] = (m[1] * m[5] - m[2] * m[4]) * inv;
    o[3] = b * inv; o[4] = (m[0] * m[8] - m[2] * m[6]) * inv; o[5] = (m[2] * m[3] - m[0] * m[5]) * inv;
    o[6] = c * inv; o[7] = (m[1] * m[6] - m[0] * m[7]) * inv; o[8] = (m[0] * m[4] - m[1] * m[3]) * inv;
}

__device__ void cam_prep_one(int t, int B, int V, int b, int inv_depth, int gnd, const float* __restrict__ tar_exts,
                           const float* __restrict__ tar_ints, const float* __restrict__ src_exts,
                           const float* __restrict__ src_ints, const float* __restrict__ near_far,
                           float* __restrict__ cams) {
    int per = V + 1;
    if (t >= B * per) return;
    int bi = t / per, v = t % per - 1;
    float* tar = cams + (size_t)bi * (TAR_STRIDE + V * SRC_STRIDE);
    const float PI_F = 3.14159265358979323846f;
    if (v < 0) {
        double E[16], Ei[16], K[9], Ki[9];
        for (int i = 0; i < 16; ++i) E[i] = tar_exts[bi * 16 + i];
        for (int i = 0; i < 9; ++i) K[i] = tar_ints[bi * 9 + i];
        invert4_f64(E, Ei);
        invert3_f64(K, Ki);
        for (int i = 0; i < 3; ++i) { tar[T_O + i] = (float)Ei[i * 4 + 3]; tar[T_Z + i] = (float)Ei[i * 4 + 2]; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;  // product of the float-rounded inverses, as the reference multiplies fp32 tensors
                for (int k = 0; k < 3; ++k) s += (double)(float)Ei[i * 4 + k] * (double)(float)Ki[k * 3 + j];
                tar[T_M + 3 * i + j] = (float)s;
            }
        float fx = tar_ints[bi * 9 + 0], fy = tar_ints[bi * 9 + 4];
        float pr = 1.f / sqrtf(fx * fy * PI_F);
        tar[T_PIXR] = pr;
        float nr = near_far[bi * 2], fr = near_far[bi * 2 + 1];
        tar[T_NEAR] = nr; tar[T_FAR] = fr;
        tar[T_MINIV] = inv_depth ? (1.f / nr - 1.f / fr) / (float)gnd : (fr - nr) / (float)gnd;
        tar[T_DISK] = (float)b * pr;
        for (int i = T_DISK + 1; i < TAR_STRIDE; ++i) tar[i] = 0.f;
    } else {
        float* s = tar + TAR_STRIDE + v * SRC_STRIDE;
        const float* Ef = src_exts + ((size_t)bi * V + v) * 16;
        const float* Kf = src_ints + ((size_t)bi * V + v) * 9;
        double E[16], Ei[16];
        for (int i = 0; i < 16; ++i) E[i] = Ef[i];
        invert4_f64(E, Ei);
        for (int i = 0; i < 12; ++i) s[S_E + i] = Ef[i];
        for (int i = 0; i < 9; ++i) { s[S_K + i] = Kf[i]; s[S_KS + i] = (i < 6) ? Kf[i] / (float)b : Kf[i]; }
        for (int i = 0; i < 3; ++i) s[S_C + i] = (float)Ei[i * 4 + 3];
        s[S_PIXR] = 1.f / sqrtf(s[S_KS + 0] * s[S_KS + 4] * PI_F);
        s[S_IPIXR] = 1.f / s[S_PIXR];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += (double)Kf[3 * r + k] * (double)Ef[4 * k + c];
                s[S_P + 4 * r + c] = (float)acc;
            }
        s[47] = 0.f;
    }
}

// ============================================================================================
// feature pyramid: NCHW (B*V, C_f+3, H, W) -> channel-last, 20-float texels, + box mips
// ============================================================================================
// One launch for the whole per-frame preparation.  Workgroups 0..ntiles-1 each take a 32x8 tile of
// one (batch, view) feature map: read NCHW along x (128-B row segments), write level 0 straight from
// registers in the chunk-planar pyramid layout [level][chunk 0..4][y][x] of float4 (a wave's 32
// neighbouring texels of one chunk are 512 contiguous bytes), park the tile in LDS, and after ONE barrier let
// one thread per (chunk, 4x4 block) build the box-filtered levels: its four level-1 texels and its level-2
// texel from LDS, level 3 from three lane shuffles.  Each level is the (a+b+c+d)*0.25 of the level below,
// as nvdiffrast's mip construction.  The last workgroup computes the camera block.
#define PT_W 32
#define PT_H 8
struct PrepArgs {
    int B, V, H, W, levels, tilesX, tilesY, ntiles, b, inv_depth, gnd;
    int lvlH[GDB_MAX_MIP + 1], lvlW[GDB_MAX_MIP + 1];
    unsigned lvlOff[GDB_MAX_MIP + 1];
    unsigned pyrStride;
    const float* img_feat; float* pyr;
    char* pyr16;  // half-precision copy of the pyramid (gdb_internal.h PYR16_*), or NULL: not asked for
    const float* src_images; int Ho, Wo, fpn;  // fpn: img_feat holds C_f channels only; the 3 colours are resampled here (N3)
    const float* tar_exts; const float* tar_ints; const float* src_exts; const float* src_ints; const float* near_far;
    float* cams;
    // dense-schedule plan: workgroups ntiles+1 .. ntiles+nplan, one wave per bundle-map row
    int nplan, S_max, adaptive, planL, planMW;
    int plan_r0, plan_nr;   // the plan covers the bundle-map rows [plan_r0, plan_r0 + plan_nr) of every batch item (gdb_prepare_rows: one rank's strip)
    const float* depth_range; int* plan;
    unsigned* smap; int smapStride; int* nwin; int* nsamp;
    char* img16; int nimg;           // half-precision RGBA copy of the source images (gdb_internal.h IMG16_*): nimg workgroups of 512 pixels, or 0
    const int* bounds;               // gdb_prepare_rows on a partial strip: per (batch, view) the tiles / image rows to build (k_strip_bounds), else NULL
    int* flat_cnt; int n_flat_cnt;   // flat schedule: the window boundaries' arrival counters, zeroed here once per frame (gdb_fused.hip: hand-off)
};
// The plan workgroups also clear the flat schedule's arrival counters (a render leaves them at zero again: the last arriver of a
// boundary resets it; this is what bounds the damage of a render that was aborted half-way).  nwg workgroups of 256 threads share the array.
__device__ __forceinline__ void zero_flat_counters(const PrepArgs& a, int wg, int nwg) {
    const int per = (a.n_flat_cnt + nwg - 1) / nwg, lo = wg * per, hi = min(a.n_flat_cnt, lo + per);
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) a.flat_cnt[i] = 0;
}

// One wave per bundle-map row: per-bundle sample counts (bundle_sampler.py:179), their exclusive prefix along the row, the row's
// compacted sample list (WsLayout::smapOff: entry s = [bundle | slot << 16 | count << 24], bundle-major / sample-minor as
// bundle_sampler.py:182-189 orders it) and its cut into WINDOWS of whole bundles holding at most 32 samples each - one window = one
// wave of k_render_dense, lane = sample.  Row record: [number of windows, first sample offset of window 0 .. nwin-1, total].
// The row is walked in rounds of 64 bundles, one per lane.
//   * Greedy cut (rows of PLAN_GREEDY_MIN .. PLAN_LDS_ROW - 1 sample offsets): every window takes bundles until the next one would not fit, so a
//     wave idles (count of the bundle that did not fit) - 1 lanes at most: ~97 % of the lanes carry a sample at S_max 3, ~94 % at 6.
//     The cut is a chain - window w + 1 starts at the last bundle start within 32 offsets of window w's - and it sits on
//     k_prepare's critical path, so it is walked in SCALAR code over a bit per sample offset (1 = a bundle starts here; OR-ed into
//     LDS words by the lanes, then held one word per lane): a hop is two v_readlane, a 64-bit shift and a find-last-bit, ~70 cycles
//     (a byte per offset in LDS read by one lane: ~180 per hop, 2.2 us of k_prepare on c2).
//   * Fixed cut (shorter and longer rows): window w = the bundles whose first sample offset falls into [planL w, planL (w + 1)), planL = 33 -
//     S_max: needs no chain, fills (planL + ~1) / 32 of the lanes (round 2's plan).
// Either way a row has at most planMW = ceil(W S_max / planL) windows.
#ifndef PLAN_GREEDY_MIN
#define PLAN_GREEDY_MIN 1024
#endif
#ifndef PLAN_LDS_ROW
#define PLAN_LDS_ROW 4096   // sample offsets of a row the greedy cut handles: 128 words of start bits, two per lane
#endif
__device__ void plan_row(const PrepArgs& a, int rowid, int lane, unsigned* __restrict__ words) {
    const int bi = rowid / a.H, row = rowid % a.H;
    const float nr = a.near_far[bi * 2], fr = a.near_far[bi * 2 + 1];
    const float miniv = a.inv_depth ? (1.f / nr - 1.f / fr) / (float)a.gnd : (fr - nr) / (float)a.gnd;  // = T_MINIV of the camera block
    const size_t hw = (size_t)a.H * a.W;
    const float* nearp = a.depth_range + ((size_t)bi * 2) * hw + (size_t)row * a.W;
    const float* farp = nearp + hw;
    int* rec = a.plan + (size_t)rowid * (a.planMW + 2);
    unsigned* sm = a.smap + (size_t)rowid * a.smapStride;
    if (!a.adaptive) {
        // Fixed counts (gdb_fixed_counts_dense): every bundle holds S_max samples, so the plan is closed-form - windows of floor(32 /
        // S_max) whole bundles, the best any cut can do - and needs neither the depth prior nor the chain (which made k_prepare 30 us
        // longer at S_max 8: 2,560 sample offsets per row, 80 hops).
        const int S = a.S_max, bpw = max(32 / S, 1), nwin = (a.W + bpw - 1) / bpw, total = a.W * S;
        for (int x = lane; x < a.W; x += 64)
            for (int k = 0; k < S; ++k) sm[x * S + k] = (unsigned)x | ((unsigned)k << 16) | ((unsigned)S << 24);
        for (int e = total + lane; e < a.smapStride; e += 64) sm[e] = 0xFFFFFFFFu;
        for (int w = lane; w < nwin; w += 64) rec[1 + w] = w * bpw * S;
        if (lane == 0) { rec[0] = nwin; rec[1 + nwin] = total; a.nwin[rowid] = nwin; a.nsamp[rowid] = total; }
        return;
    }
    // The greedy chain costs ~0.1 us per window of the row on k_prepare's critical path (c2: +2.5 us) and saves idle lanes in the
    // render (c2, 960 sample offsets per row, S_max 3: 1 us; c3, 1440: 9 us; c4, 2400, S_max 6: 23 us): taken from 1024 offsets up.
    const bool greedy = a.W * a.S_max < PLAN_LDS_ROW && a.W * a.S_max >= PLAN_GREEDY_MIN;
    if (greedy) { words[lane] = 0u; words[64 + lane] = 0u; __builtin_amdgcn_wave_barrier(); }
    // Rounds of 64 consecutive bundles, one per lane (coalesced loads and stores).  The row is ONE latency chain inside k_prepare,
    // so the ranges of the first PLAN_PRE rounds (512 bundles) are all requested up front: under k_prepare's pyramid traffic a
    // memory round trip costs microseconds, and the row pays one of them, not one per round.  (Measured, k_prepare on c2 with the
    // fixed cut: 12.9 us with a chain of dependent loads, 11.8 us this way; 14.5 us with cpl consecutive bundles per lane and a
    // single scan - strided loads, longer divergent store loops.)
    int base = 0, cprev_carry = 0;   // samples before this round; count of the bundle just before it
    constexpr int PLAN_PRE = 8;
    float npre[PLAN_PRE], fpre[PLAN_PRE];
#pragma unroll
    for (int r = 0; r < PLAN_PRE; ++r) {
        const int x = 64 * r + lane;
        npre[r] = 0.f; fpre[r] = 0.f;
        if (x < a.W) { npre[r] = nearp[x]; fpre[r] = farp[x]; }
    }
    for (int c0 = 0; c0 < a.W; c0 += 64) {
        const int x = c0 + lane;
        float n0 = 0.f, f0 = 0.f;
        const int r = c0 >> 6;
        if (r < PLAN_PRE) {
#pragma unroll
            for (int q = 0; q < PLAN_PRE; ++q) if (q == r) { n0 = npre[q]; f0 = fpre[q]; }
        } else if (x < a.W) { n0 = nearp[x]; f0 = farp[x]; }
        if (a.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; }  // bundle_sampler.py:224-226, as load_bundle does
        const int c = x < a.W ? sample_count(n0, f0, miniv, a.S_max, a.adaptive) : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int off = base + incl - c;            // sample offset of this lane's bundle
        int cprev = __shfl_up(c, 1);                // count of the bundle just before it
        if (lane == 0) cprev = cprev_carry;
        for (int k = 0; k < c; ++k)                 // the row's sample list: bundle-major, sample-minor (bundle_sampler.py:182-189)
            sm[off + k] = (unsigned)x | ((unsigned)k << 16) | ((unsigned)c << 24);
        if (greedy && x < a.W) atomicOr(&words[off >> 5], 1u << (off & 31));   // a bundle starts at offset `off`
        if (!greedy && x < a.W) {
            // consecutive bundle offsets differ by at most S_max <= planL, so every window up to the one the LAST bundle starts in
            // has a first bundle; windows beyond that one hold no bundle start (the last bundle's samples may reach into the next
            // window of offsets - they still belong to the window the bundle starts in)
            const int w = off / a.planL;
            if (x == 0 || (off - cprev) / a.planL != w) rec[1 + w] = off;
            if (x == a.W - 1) { rec[0] = w + 1; rec[2 + w] = off + c; a.nwin[rowid] = w + 1; }
        }
        base += __shfl(incl, 63);
        cprev_carry = __shfl(c, 63);
    }
    const int total = base;
    if (lane == 0) a.nsamp[rowid] = total;   // (the flat schedule reads the rows' lists as one list)
    for (int s = total + lane; s < a.smapStride; s += 64) sm[s] = 0xFFFFFFFFu;  // past the row's last sample
    if (greedy) {
        if (lane == 0) atomicOr(&words[total >> 5], 1u << (total & 31));   // sentinel: the row ends where a bundle would start
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every lane's bits are in LDS before the words are read back
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned w_lo = words[lane], w_hi = words[64 + lane];      // lane i holds words i and 64 + i of the row's start bits
        auto word = [&](int i) -> unsigned {                              // wave-uniform i
            const unsigned lo = __builtin_amdgcn_readlane(w_lo, i & 63), hi = __builtin_amdgcn_readlane(w_hi, i & 63);
            return i < 64 ? lo : (i < 128 ? hi : 0u);
        };
        int start = 0, w = 0;
        const int tot = __builtin_amdgcn_readfirstlane(total);
        while (start < tot && w < a.planMW) {
            if (lane == 0) rec[1 + w] = start;
            ++w;
            // the next window starts at the LAST bundle start among the offsets start + 1 .. start + 32 (bundles hold <= 16
            // samples, so there is one; the sentinel ends the row)
            const int q = start + 1, idx = q >> 5, sh = q & 31;
            const unsigned long long both = ((unsigned long long)word(idx + 1) << 32) | word(idx);
            const unsigned bits = (unsigned)(both >> sh);
            start = __builtin_amdgcn_readfirstlane(q + 31 - __builtin_clz(bits | 1u));
        }
        if (lane == 0) { rec[0] = w; rec[1 + w] = total; a.nwin[rowid] = w; }
    }
}

__device__ __forceinline__ float4 box4(const float4 A, const float4 B, const float4 C, const float4 D) {
    return make_float4((A.x + B.x + C.x + D.x) * 0.25f, (A.y + B.y + C.y + D.y) * 0.25f,
                       (A.z + B.z + C.z + D.z) * 0.25f, (A.w + B.w + C.w + D.w) * 0.25f);
}
__device__ __forceinline__ float4 shfl_xor4(const float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}

// ---- half-precision pyramid stores (GDB_PREP_PYR16) ------------------------------------------------------------------------------
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h4v to_h4(const float4 q) { return h4v{(_Float16)q.x, (_Float16)q.y, (_Float16)q.z, (_Float16)q.w}; }
// fp32 chunk c (channels 4c..4c+3) of texel p of a level of hw texels at byte offset lvlB: plane c & 1, second 8 bytes of the texel
// for c >= 2; chunk 4 = plane 2 (8 bytes per texel)
__device__ __forceinline__ void store16_chunk(char* __restrict__ base, unsigned lvlB, unsigned hw, unsigned p, int c, const float4 q) {
    const unsigned off = c == 4 ? lvlB + PYR16_PLANE2(hw) + 8u * p : lvlB + (unsigned)(c & 1) * 16u * hw + 16u * p + (unsigned)(c >> 1) * 8u;
    *(h4v*)(base + off) = to_h4(q);
}

// The half-precision RGBA copy of the source images (GDB_PREC_F16 renders take their colour taps from it): one thread per x pair of
// pixels - three 8-byte loads (planar fp32), one 16-byte store (r, g, b, 0 | r, g, b, 0 as halves).  Wo is even (Wo = b W).
__device__ __forceinline__ void img16_block(const float* __restrict__ src_images, char* __restrict__ img16, int nbv, int Ho, int Wo, int blk,
                                            const int* __restrict__ bounds = nullptr) {
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    const size_t plane = (size_t)Ho * Wo, pairs = plane / 2;
    const size_t q = (size_t)blk * 256 + threadIdx.x;
    if (q >= pairs * (size_t)nbv) return;
    const size_t bv = q / pairs, p = (q - bv * pairs) * 2;   // first pixel of the pair inside the image
    if (bounds) {   // a rank's row strip: only the image rows its samples' colour taps can reach (k_strip_bounds)
        const int* bb = bounds + bv * STRIP_BOUNDS;
        const int y = (int)(p / (size_t)Wo);
        if (!bb[6] && (y < bb[4] || y > bb[5])) return;
    }
    const float* im = src_images + bv * 3 * plane + p;
    const float2 r = *(const float2*)im, g = *(const float2*)(im + plane), b = *(const float2*)(im + 2 * plane);
    const h8v o = {(_Float16)r.x, (_Float16)g.x, (_Float16)b.x, (_Float16)0.f, (_Float16)r.y, (_Float16)g.y, (_Float16)b.y, (_Float16)0.f};
    *(h8v*)(img16 + (bv * plane + p) * 8) = o;
}
__global__ void __launch_bounds__(256) k_img16(const float* __restrict__ src_images, char* __restrict__ img16, int nbv, int Ho, int Wo) {
    img16_block(src_images, img16, nbv, Ho, Wo, (int)blockIdx.x);
}

__global__ void __launch_bounds__(256) k_prepare(PrepArgs a) {
    __shared__ float4 tile4[PT_W * PT_H * (GDB_CP / 4)];  // level 0 of the tile, [chunk][y][x]
    static_assert(sizeof(tile4) >= 4 * 128 * sizeof(unsigned) && PLAN_LDS_ROW <= 128 * 32, "the plan rows borrow the tile's LDS: 128 words of start bits per row");
    // Grid: [camera block | plan workgroups (4 bundle-map rows each) | pyramid tiles].  The serial pieces come FIRST: the camera
    // block is one short chain of fp64 inverses, a plan row one wave walking a latency chain (strided loads, IEEE divisions, a scan,
    // scattered stores); dispatched last they ran on after the tiles had drained (k_prepare 10.7 -> 13.7 us when every adaptive
    // frame got a plan), dispatched first they hide under the tiles.
    if (blockIdx.x == 0) {
        // (the x-pair loads of the half-precision pyramid's last texel reach PYR16_PAD bytes past it: kept finite)
        if (a.pyr16 && threadIdx.x < PYR16_PAD / 4) ((unsigned*)(a.pyr16 + (size_t)2 * a.pyrStride * a.B * a.V))[threadIdx.x] = 0u;
        for (int t = threadIdx.x; t < a.B * (a.V + 1); t += blockDim.x)
            if (t % (a.V + 1) == 0 || a.src_exts)
                cam_prep_one(t, a.B, a.V, a.b, a.inv_depth, a.gnd, a.tar_exts, a.tar_ints, a.src_exts, a.src_ints, a.near_far, a.cams);
        return;
    }
    if ((int)blockIdx.x <= a.nplan) {
        const int idx = ((int)blockIdx.x - 1) * 4 + (int)(threadIdx.x >> 6);
        zero_flat_counters(a, (int)blockIdx.x - 1, a.nplan);
        // (the plan rows borrow the tile's LDS: 128 words of bundle-start bits per row)
        if (idx < a.B * a.plan_nr) plan_row(a, (idx / a.plan_nr) * a.H + a.plan_r0 + idx % a.plan_nr, threadIdx.x & 63, (unsigned*)tile4 + (size_t)(threadIdx.x >> 6) * 128);
        return;
    }
    const int blk = (int)blockIdx.x - 1 - a.nplan;
    if (blk >= a.ntiles) { img16_block(a.src_images, a.img16, a.B * a.V, a.Ho, a.Wo, blk - a.ntiles, a.bounds); return; }
    const int tx = blk % a.tilesX, ty = (blk / a.tilesX) % a.tilesY, bv = blk / (a.tilesX * a.tilesY);
    if (a.bounds) {   // a rank's row strip (gdb_prepare_rows): this tile only if the strip's samples can reach it (k_strip_bounds, an earlier launch)
        typedef const int __attribute__((address_space(4))) kint;
        kint* bb = (kint*)(a.bounds + (size_t)bv * STRIP_BOUNDS);
        if (!bb[6] && (tx < bb[0] || tx > bb[1] || ty < bb[2] || ty > bb[3])) return;
    }
    const int x0 = tx * PT_W, y0 = ty * PT_H;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool in0 = gx < a.W && gy < a.H;
    const int nch = a.fpn ? GDB_CF : GDB_CFR;  // channels in the input map
    const float* src = a.img_feat + (size_t)bv * nch * a.H * a.W + (size_t)min(gy, a.H - 1) * a.W + min(gx, a.W - 1);
    float v[GDB_CP];
#pragma unroll
    for (int c = 0; c < GDB_CF; ++c) v[c] = in0 ? src[(size_t)c * a.H * a.W] : 0.f;
    if (a.fpn) {
        // network.py:159-164: the source image resampled to the bundle map, F.interpolate(mode='bilinear',
        // align_corners=False): src = (dst + 0.5) * (n_in / n_out) - 0.5 (>= 0 here), taps i0, min(i0 + 1, n_in - 1)
        const float sy = fmaxf(((float)min(gy, a.H - 1) + 0.5f) * ((float)a.Ho / (float)a.H) - 0.5f, 0.f);
        const float sx = fmaxf(((float)min(gx, a.W - 1) + 0.5f) * ((float)a.Wo / (float)a.W) - 0.5f, 0.f);
        const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        const int y1 = min(y0 + 1, a.Ho - 1), x1 = min(x0 + 1, a.Wo - 1);
        const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* im = a.src_images + (size_t)bv * 3 * a.Ho * a.Wo;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = im + (size_t)c * a.Ho * a.Wo;
            const float top = lx0 * pl[(size_t)y0 * a.Wo + x0] + lx1 * pl[(size_t)y0 * a.Wo + x1];
            const float bot = lx0 * pl[(size_t)y1 * a.Wo + x0] + lx1 * pl[(size_t)y1 * a.Wo + x1];
            v[GDB_CF + c] = in0 ? ly0 * top + ly1 * bot : 0.f;
        }
    } else {
#pragma unroll
        for (int c = GDB_CF; c < GDB_CFR; ++c) v[c] = in0 ? src[(size_t)c * a.H * a.W] : 0.f;
    }
    v[GDB_CFR] = 0.f;
    float4* pyr4 = a.pyr ? (float4*)(a.pyr + (size_t)bv * a.pyrStride) : nullptr;   // nullptr: GDB_PREP_PYR16_ONLY
    char* p16 = a.pyr16 ? a.pyr16 + (size_t)bv * 2 * a.pyrStride : nullptr;
#pragma unroll
    for (int c = 0; c < GDB_CP / 4; ++c) {
        const float4 q = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
        tile4[(c * PT_H + ly) * PT_W + lx] = q;                           // texels outside the map are zero
        if (in0 && pyr4) pyr4[((size_t)c * a.H + gy) * a.W + gx] = q;
    }
    if (p16 && in0) {  // level 0 of the half-precision copy: the texel's two 16-byte planes and its 8-byte plane, straight from registers
        typedef _Float16 h8v __attribute__((ext_vector_type(8)));
        const unsigned hw = (unsigned)(a.H * a.W), p = (unsigned)(gy * a.W + gx);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            h8v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (_Float16)v[4 * hh + e]; o[4 + e] = (_Float16)v[8 + 4 * hh + e]; }
            *(h8v*)(p16 + (size_t)hh * 16u * hw + 16u * p) = o;
        }
        *(h4v*)(p16 + PYR16_PLANE2(hw) + 8u * p) = h4v{(_Float16)v[16], (_Float16)v[17], (_Float16)v[18], (_Float16)v[19]};
    }
    if (a.levels < 1) return;
    __syncthreads();
    // one thread per (chunk, 4x4 block of the tile): 5 x (8 x 2) = 80 threads, 16 consecutive lanes per chunk
    const int t = threadIdx.x;
    if (t >= (GDB_CP / 4) * 16) return;
    const int ch = t >> 4, bx = t & 7, by = (t >> 3) & 1;
    const float4* T = tile4 + ch * PT_H * PT_W;
    float4 l1[2][2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int sx = 4 * bx + 2 * i2, sy = 4 * by + 2 * j2;
            l1[j2][i2] = box4(T[sy * PT_W + sx], T[sy * PT_W + sx + 1], T[(sy + 1) * PT_W + sx], T[(sy + 1) * PT_W + sx + 1]);
            const int px = (x0 >> 1) + 2 * bx + i2, py = (y0 >> 1) + 2 * by + j2;
            if (px < a.lvlW[1] && py < a.lvlH[1]) {
                if (pyr4) pyr4[(a.lvlOff[1] >> 2) + ((size_t)ch * a.lvlH[1] + py) * a.lvlW[1] + px] = l1[j2][i2];
                if (p16) store16_chunk(p16, 2u * a.lvlOff[1], (unsigned)(a.lvlH[1] * a.lvlW[1]), (unsigned)(py * a.lvlW[1] + px), ch, l1[j2][i2]);
            }
        }
    if (a.levels < 2) return;
    const float4 l2 = box4(l1[0][0], l1[0][1], l1[1][0], l1[1][1]);
    {
        const int px = (x0 >> 2) + bx, py = (y0 >> 2) + by;
        if (px < a.lvlW[2] && py < a.lvlH[2]) {
            if (pyr4) pyr4[(a.lvlOff[2] >> 2) + ((size_t)ch * a.lvlH[2] + py) * a.lvlW[2] + px] = l2;
            if (p16) store16_chunk(p16, 2u * a.lvlOff[2], (unsigned)(a.lvlH[2] * a.lvlW[2]), (unsigned)(py * a.lvlW[2] + px), ch, l2);
        }
    }
    if (a.levels < 3) return;
    // the level-2 neighbours (bx^1, by), (bx, by^1), (bx^1, by^1) sit in lanes t^1, t^8, t^9 of the same 16-lane group
    const float4 nb = shfl_xor4(l2, 1), nc = shfl_xor4(l2, 8), nd = shfl_xor4(l2, 9);
    if ((bx & 1) == 0 && by == 0) {
        const float4 l3 = box4(l2, nb, nc, nd);
        const int px = (x0 >> 3) + (bx >> 1), py = y0 >> 3;
        if (px < a.lvlW[3] && py < a.lvlH[3]) {
            if (pyr4) pyr4[(a.lvlOff[3] >> 2) + ((size_t)ch * a.lvlH[3] + py) * a.lvlW[3] + px] = l3;
            if (p16) store16_chunk(p16, 2u * a.lvlOff[3], (unsigned)(a.lvlH[3] * a.lvlW[3]), (unsigned)(py * a.lvlW[3] + px), ch, l3);
        }
    }
}

// The half-precision copy alone, from the fp32 pyramid an earlier gdb_prepare built (a GDB_PREC_F16 render that is not told the
// copy exists - GDB_SCHED_PYR16_READY - makes it first): one thread per (texel of any level, fp32 chunk).
__global__ void __launch_bounds__(256) k_pyr16(const float* __restrict__ pyr, char* __restrict__ pyr16, unsigned pyrStride, int nbv, int levels,
                                               int H, int W, unsigned lo1, unsigned lo2, unsigned lo3) {
    const unsigned per = pyrStride / 4;                       // float4 chunks per (batch, view)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)per * nbv) return;
    const unsigned bv = (unsigned)(t / per), q = (unsigned)(t % per);
    // which level: chunk-planar blocks of 5 hw_l float4s at float4 offset lvlOff[l] / 4
    int l = 0; unsigned lo = 0;
    if (levels >= 1 && q >= lo1 / 4) { l = 1; lo = lo1; }
    if (levels >= 2 && q >= lo2 / 4) { l = 2; lo = lo2; }
    if (levels >= 3 && q >= lo3 / 4) { l = 3; lo = lo3; }
    const unsigned hw = (unsigned)((H >> l) * (W >> l)), r = q - lo / 4, c = r / hw, p = r % hw;
    if (c >= 5) return;
    const float4 v = ((const float4*)(pyr + (size_t)bv * pyrStride))[q];
    store16_chunk(pyr16 + (size_t)bv * 2 * pyrStride, 2u * lo, hw, p, (int)c, v);
}
int gdb_build_pyr16(const GdbConfig* cfg, const GdbFrame* f, void* ws, hipStream_t st) {
    WsLayout L = ws_layout(*cfg, *f);
    const size_t n = (L.pyrStride / 4) * (size_t)f->B * f->V;
    hipLaunchKernelGGL(k_pyr16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)((char*)ws + L.pyrOff), (char*)ws + L.pyr16Off,
                       (unsigned)L.pyrStride, f->B * f->V, L.levels, f->H, f->W, (unsigned)L.lvlOff[1], (unsigned)L.lvlOff[2], (unsigned)L.lvlOff[3]);
    LAUNCH_CHECK("k_pyr16");
    hipMemsetAsync((char*)ws + L.pyr16Off + (size_t)2 * L.pyrStride * f->B * f->V, 0, PYR16_PAD, st);
    if (!f->d_src_images) return gdb_fail(GDB_E_BADARG, "a GDB_PREC_F16 render needs frame->d_src_images");
    if (cfg->bundle_size != 2) return GDB_OK;   // (the x-pair image copy is the bundle_size 2 kernels': k_bundle_colours reads the fp32 images)
    const size_t npair = (size_t)f->B * f->V * f->Ho * f->Wo / 2;
    hipLaunchKernelGGL(k_img16, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, st, f->d_src_images, (char*)ws + L.img16Off, f->B * f->V, f->Ho, f->Wo);
    LAUNCH_CHECK("k_img16");
    return GDB_OK;
}

// ---- gdb_prepare_rows: what of the source views a row strip's samples can reach (round 6; SURVEY.md 8(e), VERDICT r05 task 4) --------------
// A rank that renders the bundle-map rows [r0, r1) needs, of every source view, only the pyramid texels (all levels) and image pixels its
// samples project onto.  Every point the strip samples - sub-ray points o + d(x, y) z and their bundle means - lies in the truncated
// pyramid { o + M [x, y, 1]^T z : (x, y) in [0, Wo] x [r0 b, r1 b], z in [z_min, z_max] } (bundle_sampler.py:67-71, :254-256), z_min / z_max =
// the extremes of the strip's depth prior (a sample's depth lies between its bundle's near and far, in depth and in disparity sampling:
// :122-191).  That body is convex, so while its 8 vertices lie in front of a source camera (K (E p + t) has z >= a positive floor) its
// perspective image is the convex hull of theirs and the bounding box of the 8 projections bounds every tap coordinate.  Margins: a
// bilinear tap pair at mip level l reaches 1.5 * 2^l level-0 texels from the level-0 coordinate (12 at level 3), + 2 for the rounding of
// this fp32 estimate against the render's own arithmetic; colour taps 1 pixel + 1.  Anything the construction cannot vouch for - a vertex
// behind or near a source camera's plane, a non-finite or non-positive depth in the strip - sets `whole`: that view is built in full.
// One workgroup per batch item: the strip's depth extremes by a block reduction, then one thread per view.  All fp32, closed-form
// inverses; the camera block proper (fp64) is k_prepare's.
struct BoundsArgs {
    int B, V, H, W, Ho, Wo, b, r0, r1, tilesX, tilesY;
    const float* depth_range; const float* tar_exts; const float* tar_ints; const float* src_exts; const float* src_ints;
    int* bounds;
};
#define SB_THREADS 1024
#define SB_UNROLL 8
__global__ void __launch_bounds__(SB_THREADS) k_strip_bounds(BoundsArgs a) {
    __shared__ float s_lo[SB_THREADS / 64], s_hi[SB_THREADS / 64];
    __shared__ int s_bad[SB_THREADS / 64];
    const int bi = (int)blockIdx.x, t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    const size_t hw = (size_t)a.H * a.W;
    const float* nearp = a.depth_range + ((size_t)bi * 2) * hw + (size_t)a.r0 * a.W;
    const float* farp = nearp + hw;
    const int n = (a.r1 - a.r0) * a.W;
    float lo = INFINITY, hi = -INFINITY; int bad = 0;
    // (this workgroup is a latency chain ahead of k_prepare: 2 SB_UNROLL loads per thread in flight per round trip - c5's strip of 75 rows
    // x 800 bundles is 8 round trips this way, 470 with one load at a time in 256 threads)
    for (int i0 = t; i0 < n; i0 += SB_THREADS * SB_UNROLL) {
        float x[SB_UNROLL], y[SB_UNROLL];
#pragma unroll
        for (int u = 0; u < SB_UNROLL; ++u) {
            const int i = i0 + u * SB_THREADS;
            x[u] = i < n ? nearp[i] : 1.f; y[u] = i < n ? farp[i] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < SB_UNROLL; ++u) {
            if (i0 + u * SB_THREADS < n) {
                if (!(x[u] > 0.f) || !(y[u] > 0.f) || !(fabsf(x[u]) < INFINITY) || !(fabsf(y[u]) < INFINITY)) bad = 1;   // (NaN fails every comparison)
                lo = fminf(lo, fminf(x[u], y[u])); hi = fmaxf(hi, fmaxf(x[u], y[u]));
            }
        }
    }
    for (int d = 32; d; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d)); hi = fmaxf(hi, __shfl_xor(hi, d)); bad |= __shfl_xor(bad, d); }
    if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_bad[wv] = bad; }
    __syncthreads();
    if (t >= a.V) return;
    float zmin = INFINITY, zmax = -INFINITY; int anybad = 0;
    for (int i = 0; i < SB_THREADS / 64; ++i) { zmin = fminf(zmin, s_lo[i]); zmax = fmaxf(zmax, s_hi[i]); anybad |= s_bad[i]; }
    int whole = anybad || !(zmin > 0.f) || !(zmax >= zmin) || n <= 0;
    int* out = a.bounds + ((size_t)bi * a.V + t) * STRIP_BOUNDS;
    // target camera: c2w = E^-1, ray matrix M = R_c2w K^-1 (fp32 closed forms; k_prepare's camera block does the same in fp64)
    const float* E = a.tar_exts + bi * 16; const float* K = a.tar_ints + bi * 9;
    double Ed[16], Ei[16], Kd[9], Ki[9];
    for (int i = 0; i < 16; ++i) Ed[i] = E[i];
    for (int i = 0; i < 9; ++i) Kd[i] = K[i];
    invert4_f64(Ed, Ei); invert3_f64(Kd, Ki);
    float M[9], o[3];
    for (int i = 0; i < 3; ++i) {
        o[i] = (float)Ei[i * 4 + 3];
        for (int j = 0; j < 3; ++j) M[3 * i + j] = (float)(Ei[i * 4] * Ki[j] + Ei[i * 4 + 1] * Ki[3 + j] + Ei[i * 4 + 2] * Ki[6 + j]);
    }
    const float* Es = a.src_exts + ((size_t)bi * a.V + t) * 16; const float* Ks = a.src_ints + ((size_t)bi * a.V + t) * 9;
    float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    const float floor_z = 1e-3f * zmin;   // a vertex this close to the camera plane (or behind it): no bound
    for (int c = 0; c < 8 && !whole; ++c) {
        const float px = (c & 1) ? (float)a.Wo : 0.f, py = (c & 2) ? (float)(a.r1 * a.b) : (float)(a.r0 * a.b), z = (c & 4) ? zmax : zmin;
        float p[3], cam[3], im[3];
        for (int i = 0; i < 3; ++i) p[i] = o[i] + (M[3 * i] * px + M[3 * i + 1] * py + M[3 * i + 2]) * z;
        for (int r = 0; r < 3; ++r) cam[r] = Es[4 * r] * p[0] + Es[4 * r + 1] * p[1] + Es[4 * r + 2] * p[2] + Es[4 * r + 3];
        for (int r = 0; r < 3; ++r) im[r] = Ks[3 * r] * cam[0] + Ks[3 * r + 1] * cam[1] + Ks[3 * r + 2] * cam[2];
        if (!(im[2] > floor_z) || !(fabsf(im[0]) < INFINITY) || !(fabsf(im[1]) < INFINITY)) { whole = 1; break; }
        const float u = im[0] / im[2], v = im[1] / im[2];
        xlo = fminf(xlo, u); xhi = fmaxf(xhi, u); ylo = fminf(ylo, v); yhi = fmaxf(yhi, v);
    }
    if (!whole && !(fabsf(xlo) < 1e9f && fabsf(xhi) < 1e9f && fabsf(ylo) < 1e9f && fabsf(yhi) < 1e9f)) whole = 1;
    if (whole) { out[0] = 0; out[1] = a.tilesX - 1; out[2] = 0; out[3] = a.tilesY - 1; out[4] = 0; out[5] = a.Ho - 1; out[6] = 1; out[7] = 0; return; }
    // source pixels -> level-0 texel coordinates of the feature pyramid: pixel / b - 0.5 (S_KS: rows 0, 1 of K divided by b; :311-312, :351-353)
    // The margins go AROUND THE CLAMPED coordinate (clamp-to-edge / border addressing clamps the coordinate first and then reads texel
    // floor(c) and its neighbour floor(c) + 1 - with weight 0 at the edge, but it is read - at every level): a body that projects wholly
    // outside a map still reads the map's first / last texels and their neighbours.
    const float ib = 1.f / (float)a.b, mt = 14.f;
    auto clampf = [](float v, float hi) { return fminf(fmaxf(v, 0.f), hi); };
    const float tx0 = fmaxf(clampf(xlo * ib - 0.5f, (float)(a.W - 1)) - mt, 0.f), tx1 = fminf(clampf(xhi * ib - 0.5f, (float)(a.W - 1)) + mt, (float)(a.W - 1));
    const float ty0 = fmaxf(clampf(ylo * ib - 0.5f, (float)(a.H - 1)) - mt, 0.f), ty1 = fminf(clampf(yhi * ib - 0.5f, (float)(a.H - 1)) + mt, (float)(a.H - 1));
    out[0] = (int)floorf(tx0) / PT_W; out[1] = min((int)floorf(tx1) / PT_W, a.tilesX - 1);
    out[2] = (int)floorf(ty0) / PT_H; out[3] = min((int)floorf(ty1) / PT_H, a.tilesY - 1);
    out[4] = (int)fmaxf(floorf(clampf(ylo - 0.5f, (float)(a.Ho - 1))) - 2.f, 0.f);
    out[5] = (int)fminf(ceilf(clampf(yhi - 0.5f, (float)(a.Ho - 1))) + 2.f, (float)(a.Ho - 1));
    out[6] = 0; out[7] = 0;
}

static int prepare_common(const GdbConfig* cfg, const GdbFrame* f, const float* fpn_feat, int flags, void* ws, size_t ws_bytes, void* stream_, int row_begin = 0, int row_end = -1);

// The dense plan alone (for a render call that asks for GDB_SCHED_DENSE on a frame whose prepare did not build it).
__global__ void __launch_bounds__(256) k_plan(PrepArgs a) {
    __shared__ unsigned words[4 * 128];
    const int idx = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    zero_flat_counters(a, (int)blockIdx.x, (int)gridDim.x);
    if (idx < a.B * a.plan_nr) plan_row(a, (idx / a.plan_nr) * a.H + a.plan_r0 + idx % a.plan_nr, threadIdx.x & 63, words + (size_t)(threadIdx.x >> 6) * 128);
}
int gdb_build_dense_plan(const GdbConfig* cfg, const GdbFrame* f, void* ws, hipStream_t st) {
    WsLayout L = ws_layout(*cfg, *f);
    PrepArgs a{};
    a.B = f->B; a.H = f->H; a.W = f->W; a.inv_depth = cfg->inv_depth; a.gnd = cfg->global_num_depth;
    a.near_far = f->d_near_far; a.S_max = cfg->max_num_samples; a.adaptive = cfg->is_adaptive; a.planL = L.planL; a.planMW = L.planMW;
    a.depth_range = f->d_depth_range; a.plan = (int*)((char*)ws + L.planOff);
    a.smap = (unsigned*)((char*)ws + L.smapOff); a.smapStride = L.smapStride; a.nwin = (int*)((char*)ws + L.nwinOff); a.nsamp = (int*)((char*)ws + L.nsampOff);
    a.flat_cnt = (int*)((char*)ws + L.sideHdrOff); a.n_flat_cnt = L.flatMaxTiles + 1;
    a.plan_r0 = 0; a.plan_nr = f->H;
    hipLaunchKernelGGL(k_plan, dim3((f->B * f->H + 3) / 4), dim3(256), 0, st, a);
    LAUNCH_CHECK("k_plan");
    return GDB_OK;
}

extern "C" int gdb_prepare(const GdbConfig* cfg, const GdbFrame* f, void* ws, size_t ws_bytes, void* stream_) {
    return prepare_common(cfg, f, nullptr, 0, ws, ws_bytes, stream_);
}

extern "C" int gdb_prepare_ex(const GdbConfig* cfg, const GdbFrame* f, const float* d_fpn_feat, int32_t flags, void* ws, size_t ws_bytes, void* stream_) {
    if (flags & ~GDB_PREP_ALL) return gdb_fail(GDB_E_BADARG, "gdb_prepare_ex: unknown flag bits 0x%x", (unsigned)(flags & ~GDB_PREP_ALL));
    if ((flags & GDB_PREP_PYR16_ONLY) && !(flags & GDB_PREP_PYR16)) return gdb_fail(GDB_E_BADARG, "gdb_prepare_ex: GDB_PREP_PYR16_ONLY needs GDB_PREP_PYR16");
    if (d_fpn_feat && (!f || !f->d_src_images)) return gdb_fail(GDB_E_BADARG, "gdb_prepare_ex with d_fpn_feat resamples frame->d_src_images: it is NULL");
    return prepare_common(cfg, f, d_fpn_feat, flags, ws, ws_bytes, stream_);
}

// gdb_prepare_ex for a rank that renders ONE row strip of the frame: camera block and feature pyramid as ever (a strip's
// samples project anywhere into the source views), the list schedules' plan for the bundle-map rows [row_begin, row_end) of every batch
// item only.
extern "C" int gdb_prepare_rows(const GdbConfig* cfg, const GdbFrame* f, const float* d_fpn_feat, int32_t flags, int32_t row_begin, int32_t row_end,
                                void* ws, size_t ws_bytes, void* stream_) {
    if (flags & ~GDB_PREP_ALL) return gdb_fail(GDB_E_BADARG, "gdb_prepare_rows: unknown flag bits 0x%x", (unsigned)(flags & ~GDB_PREP_ALL));
    if ((flags & GDB_PREP_PYR16_ONLY) && !(flags & GDB_PREP_PYR16)) return gdb_fail(GDB_E_BADARG, "gdb_prepare_rows: GDB_PREP_PYR16_ONLY needs GDB_PREP_PYR16");
    if (d_fpn_feat && (!f || !f->d_src_images)) return gdb_fail(GDB_E_BADARG, "gdb_prepare_rows with d_fpn_feat resamples frame->d_src_images: it is NULL");
    if (row_end < 0) return gdb_fail(GDB_E_SHAPE, "row strip [%d,%d) is negative", row_begin, row_end);
    if ((flags & GDB_PREP_STRIP_REACH) && (flags & GDB_PREP_STRIP_WHOLE)) return gdb_fail(GDB_E_BADARG, "gdb_prepare_rows: GDB_PREP_STRIP_REACH and GDB_PREP_STRIP_WHOLE exclude each other");
    return prepare_common(cfg, f, d_fpn_feat, flags, ws, ws_bytes, stream_, row_begin, row_end);
}

extern "C" int gdb_prepare_fpn(const GdbConfig* cfg, const GdbFrame* f, const float* d_fpn_feat, void* ws, size_t ws_bytes, void* stream_) {
    if (!d_fpn_feat) return gdb_fail(GDB_E_BADARG, "d_fpn_feat is NULL");
    if (!f || !f->d_src_images) return gdb_fail(GDB_E_BADARG, "gdb_prepare_fpn resamples frame->d_src_images: it is NULL");
    return prepare_common(cfg, f, d_fpn_feat, 0, ws, ws_bytes, stream_);
}

static int prepare_common(const GdbConfig* cfg, const GdbFrame* f, const float* fpn_feat, int flags, void* ws, size_t ws_bytes, void* stream_, int row_begin, int row_end) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, f, false); if (rc) return rc;
    if (!ws) return gdb_fail(GDB_E_BADARG, "workspace is NULL");
    // target camera + scene range are always needed; the source side (views, feature map) may be absent
    // when only build_rays / sample will follow (the reference has no source views at that point either)
    if (!f->d_tar_exts || !f->d_tar_ints || !f->d_near_far) return gdb_fail(GDB_E_BADARG, "frame has a NULL target-camera pointer");
    if ((f->d_src_exts == nullptr) != (f->d_src_ints == nullptr)) return gdb_fail(GDB_E_BADARG, "source extrinsics and intrinsics must come together");
    WsLayout L = ws_layout(*cfg, *f);
    if (ws_bytes < L.total) return gdb_fail(GDB_E_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream_;
    PrepArgs a{};
    a.B = f->B; a.V = f->V; a.H = f->H; a.W = f->W; a.levels = L.levels;
    if (a.levels > 3) return gdb_fail(GDB_E_BADARG, "max_mipmap_level > 3 unsupported by the tile kernel");
    a.tilesX = (f->W + PT_W - 1) / PT_W; a.tilesY = (f->H + PT_H - 1) / PT_H;
    a.ntiles = (f->d_img_feat || fpn_feat) ? a.tilesX * a.tilesY * f->B * f->V : 0;
    // GDB_PREP_PYR16 with a feature map also copies the source images to half precision (the GDB_PREC_F16 kernels read their colour
    // taps from that copy): without them a later f16 render told GDB_SCHED_PYR16_READY would read uninitialised workspace
    if ((flags & GDB_PREP_PYR16) && a.ntiles && !f->d_src_images)
        return gdb_fail(GDB_E_BADARG, "GDB_PREP_PYR16 copies frame->d_src_images to half precision beside the pyramid: it is NULL");
    // GDB_PREP_SOURCES_READY (ABI v7): the caller vouches that the products that depend on the SOURCE views alone - the feature pyramid(s)
    // and the half-precision image copy the other flags name - are in this workspace from an earlier prepare of the same source tensors,
    // contents unchanged: only the camera block and the plan are rebuilt (a sweep of target views over fixed source views).
    const bool sources_ready = (flags & GDB_PREP_SOURCES_READY) != 0;
    if (sources_ready) a.ntiles = 0;
    a.b = cfg->bundle_size; a.inv_depth = cfg->inv_depth; a.gnd = cfg->global_num_depth;
    for (int i = 0; i <= GDB_MAX_MIP; ++i) { a.lvlH[i] = L.lvlH[i]; a.lvlW[i] = L.lvlW[i]; a.lvlOff[i] = (unsigned)L.lvlOff[i]; }
    a.pyrStride = (unsigned)L.pyrStride;
    a.img_feat = fpn_feat ? fpn_feat : f->d_img_feat; a.pyr = (float*)((char*)ws + L.pyrOff);
    a.pyr16 = (flags & GDB_PREP_PYR16) ? (char*)ws + L.pyr16Off : nullptr;
    if (flags & GDB_PREP_PYR16_ONLY) a.pyr = nullptr;   // the fp32 pyramid is not written
    // the half-precision copy is addressed with 32-bit byte offsets inside one (batch, view) block
    if (a.pyr16 && (size_t)2 * L.pyrStride >= ((size_t)1 << 32)) return gdb_fail(GDB_E_SHAPE, "feature map too large for the half-precision pyramid");
    a.fpn = fpn_feat != nullptr; a.src_images = f->d_src_images; a.Ho = f->Ho; a.Wo = f->Wo;
    a.tar_exts = f->d_tar_exts; a.tar_ints = f->d_tar_ints; a.src_exts = f->d_src_exts; a.src_ints = f->d_src_ints;
    a.near_far = f->d_near_far; a.cams = (float*)((char*)ws + L.camsOff);
    // the dense schedule's plan needs the depth prior; a frame prepared without it (build_rays / sample only) has none
    a.S_max = cfg->max_num_samples; a.adaptive = cfg->is_adaptive; a.planL = L.planL; a.planMW = L.planMW;
    a.depth_range = f->d_depth_range; a.plan = (int*)((char*)ws + L.planOff);
    a.smap = (unsigned*)((char*)ws + L.smapOff); a.smapStride = L.smapStride; a.nwin = (int*)((char*)ws + L.nwinOff); a.nsamp = (int*)((char*)ws + L.nsampOff);
    a.flat_cnt = (int*)((char*)ws + L.sideHdrOff); a.n_flat_cnt = L.flatMaxTiles + 1;
    // Built here (inside this launch, ~1 us) for adaptive configs (and the fixed-count ones the dense schedule takes:
    // gdb_fixed_counts_dense) whenever the frame carries its depth prior: a render call that
    // is told so (GDB_SCHED_PLAN_READY) uses it as it stands; any other dense render builds the plan itself (gdb_build_dense_plan,
    // a launch of its own on the same stream).
    if (row_end < 0) row_end = f->H;
    if (row_begin < 0 || row_end > f->H || row_begin > row_end) return gdb_fail(GDB_E_SHAPE, "row strip [%d,%d) outside [0,%d]", row_begin, row_end, f->H);
    a.plan_r0 = row_begin; a.plan_nr = row_end - row_begin;
    a.nplan = (f->d_depth_range && a.plan_nr > 0 && (cfg->is_adaptive || gdb_fixed_counts_dense(*cfg, f->V))) ? (f->B * a.plan_nr + 3) / 4 : 0;
    // the half-precision RGBA copy of the source images rides in the same launch, behind the pyramid tiles
    // (the copy is made of x PAIRS of pixels, for the fused kernels, which are built for bundle_size 2: an even pixel count; other
    // configs - the operator mirrors' - never read it)
    a.img16 = (a.pyr16 && f->d_src_images && a.ntiles && cfg->bundle_size == 2 && (((size_t)f->Ho * f->Wo) & 1) == 0) ? (char*)ws + L.img16Off : nullptr;
    a.nimg = a.img16 ? (int)(((size_t)f->B * f->V * f->Ho * f->Wo / 2 + 255) / 256) : 0;
    // A rank's row strip (gdb_prepare_rows on fewer rows than the frame has): only the pyramid tiles / image rows the strip's samples can
    // reach are built - k_strip_bounds, a launch of its own ahead of this one (the tiles read its result), decides per (batch, view).
    // Renders of rows OUTSIDE [row_begin, row_end) are then invalid on this workspace until a prepare of the whole frame.
    a.bounds = nullptr;
    bool reach = a.ntiles && a.plan_nr > 0 && a.plan_nr < f->H && f->d_depth_range && f->d_src_exts && !(flags & GDB_PREP_STRIP_WHOLE);
    if (reach && !(flags & GDB_PREP_STRIP_REACH)) {
        // by size: the bound is a launch of its own ahead of the tiles (a one-workgroup latency chain + a kernel boundary, ~8 us of stream
        // time) - worth it once the whole-frame tile work moves >= 128 MB (per texel 76 B read, 106 B of fp32 pyramid, 53 B of
        // half-precision pyramid; per source pixel 20 B for the half-precision image copy)
        const size_t texels = (size_t)f->B * f->V * f->H * f->W, px = (size_t)f->B * f->V * f->Ho * f->Wo;
        const size_t est = texels * (76 + (a.pyr ? 106 : 0) + (a.pyr16 ? 53 : 0)) + (a.img16 ? px * 20 : 0);
        reach = est >= ((size_t)128 << 20);
    }
    if (reach) {
        BoundsArgs b{};
        b.B = f->B; b.V = f->V; b.H = f->H; b.W = f->W; b.Ho = f->Ho; b.Wo = f->Wo; b.b = cfg->bundle_size; b.r0 = row_begin; b.r1 = row_end;
        b.tilesX = a.tilesX; b.tilesY = a.tilesY;
        b.depth_range = f->d_depth_range; b.tar_exts = f->d_tar_exts; b.tar_ints = f->d_tar_ints; b.src_exts = f->d_src_exts; b.src_ints = f->d_src_ints;
        b.bounds = (int*)((char*)ws + L.boundsOff);
        hipLaunchKernelGGL(k_strip_bounds, dim3(f->B), dim3(SB_THREADS), 0, st, b);
        LAUNCH_CHECK("k_strip_bounds");
        a.bounds = b.bounds;
    }
    hipLaunchKernelGGL(k_prepare, dim3(a.ntiles + 1 + a.nplan + a.nimg), dim3(256), 0, st, a);
    LAUNCH_CHECK("k_prepare");
    return GDB_OK;
}

// ============================================================================================
// A1  build_rays (materialised, for the operator mirror)
// ============================================================================================
__global__ void k_build_rays(DevFrame f, float* __restrict__ rays_d, float* __restrict__ uv, float* __restrict__ rays_o,
                             float* __restrict__ z_axis, float* __restrict__ pixr) {
    size_t n = (size_t)f.B * f.Ho * f.Wo;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (size_t)f.B * 3) {
        int bi = (int)(t / 3), i = (int)(t % 3);
        rays_o[t] = tar_cam(f, bi)[T_O + i];
        z_axis[t] = tar_cam(f, bi)[T_Z + i];
        if (i == 0) pixr[bi] = tar_cam(f, bi)[T_PIXR];
    }
    if (t >= n) return;
    int px = (int)(t % f.Wo); size_t r = t / f.Wo;
    int py = (int)(r % f.Ho); int bi = (int)(r / f.Ho);
    float x = (float)px + 0.5f, y = (float)py + 0.5f;
    float d[3];
    ray_dir(tar_cam(f, bi) + T_M, x, y, d);
    rays_d[t * 3 + 0] = d[0]; rays_d[t * 3 + 1] = d[1]; rays_d[t * 3 + 2] = d[2];
    if (bi == 0) {
        uv[((size_t)py * f.Wo + px) * 2 + 0] = 2.f * x / (float)f.Wo - 1.f;
        uv[((size_t)py * f.Wo + px) * 2 + 1] = 2.f * y / (float)f.Ho - 1.f;
    }
}

extern "C" int gdb_build_rays(const GdbConfig* cfg, const GdbFrame* f, const void* ws, float* rays_d, float* uv,
                              float* rays_o, float* z_axis, float* pixr, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, f, false); if (rc) return rc;
    if (!ws || !rays_d || !uv || !rays_o || !z_axis || !pixr) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    WsLayout L = ws_layout(*cfg, *f);
    DevFrame d = dev_frame(*cfg, *f, L, ws);
    size_t n = (size_t)f->B * f->Ho * f->Wo;
    hipLaunchKernelGGL(k_build_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, d, rays_d, uv,
                       rays_o, z_axis, pixr);
    LAUNCH_CHECK("k_build_rays");
    return GDB_OK;
}

// ============================================================================================
// A2+A3  sample
// ============================================================================================
template <int BB>
__global__ void k_counts(DevFrame f, int32_t* __restrict__ cnt) {
    size_t nb = (size_t)f.B * f.H * f.W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    int w = (int)(t % f.W); size_t r = t / f.W; int h = (int)(r % f.H); int bi = (int)(r / f.H);
    size_t hw = (size_t)f.H * f.W, p = (size_t)h * f.W + w;
    float n0 = f.depth_range[((size_t)bi * 2) * hw + p], f0 = f.depth_range[((size_t)bi * 2 + 1) * hw + p];
    if (f.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; }
    cnt[t] = sample_count(n0, f0, tar_cam(f, bi)[T_MINIV], f.S_max, f.adaptive);
}

// Exclusive scan of int32 counts: per-block scan, serial scan of the block sums, add-back.
__global__ void k_scan_block(size_t n, const int32_t* __restrict__ in, int32_t* __restrict__ out, int32_t* __restrict__ bsum) {
    __shared__ int32_t wsum[4];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    int32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int32_t wo = 0;
    for (int i = 0; i < wid; ++i) wo += wsum[i];
    int32_t ex = wo + inc - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (threadIdx.x == 255) bsum[blockIdx.x] = wo + inc;
}

__global__ void k_scan_sums(int nblk, int32_t* __restrict__ bsum) {  // one thread; nblk <= a few thousand
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t acc = 0;
        for (int i = 0; i < nblk; ++i) { int32_t t = bsum[i]; bsum[i] = acc; acc += t; }
        bsum[nblk] = acc;
    }
}

__global__ void k_scan_add(size_t n, const int32_t* __restrict__ bsum, int32_t* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] += bsum[t / SCAN_BLOCK];
}

template <int BB>
__global__ void k_sample(DevFrame f, const int32_t* __restrict__ offs, const int32_t* __restrict__ bsum, int nblk,
                         float* __restrict__ rays_xyz, float* __restrict__ uvd, float* __restrict__ z_vals,
                         float* __restrict__ ball_radii, int64_t* __restrict__ indices, int64_t* __restrict__ per_batch,
                         int32_t* __restrict__ spb, int64_t* __restrict__ total) {
    size_t nb = (size_t)f.B * f.H * f.W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *total = bsum[nblk];
    if (t >= nb) return;
    int w = (int)(t % f.W); size_t r = t / f.W; int h = (int)(r % f.H); int bi = (int)(r / f.H);
    Bundle<BB> q;
    load_bundle<BB>(f, bi, h, w, q);
    spb[t] = q.count;
    int32_t off = offs[t];
    size_t hw = (size_t)f.H * f.W;
    if ((size_t)h * f.W + w == hw - 1) {  // last bundle of the batch item closes its sample range
        per_batch[bi] = (int64_t)off + q.count;  // inclusive end; k_per_batch turns ends into counts
    }
    for (int k = 0; k < q.count; ++k) {
        float z, dn, ball, xyz[BB][3], ctr[3];
        bundle_sample<BB>(f, q, k, z, dn, xyz, ctr, ball);
        size_t i = (size_t)off + k;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int s = 0; s < BB; ++s) rays_xyz[(i * 3 + c) * BB + s] = xyz[s][c];
        uvd[i * 3 + 0] = q.u; uvd[i * 3 + 1] = q.v; uvd[i * 3 + 2] = dn;
        z_vals[i] = z; ball_radii[i] = ball; indices[i] = (int64_t)t;
    }
}

// per_batch holds inclusive end offsets after k_sample; turn them into per-batch counts.
__global__ void k_per_batch(int B, int64_t* __restrict__ per_batch) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t prev = 0;
        for (int i = 0; i < B; ++i) { int64_t e = per_batch[i]; per_batch[i] = e - prev; prev = e; }
    }
}

template <int BB>
static int sample_impl(const DevFrame& d, const WsLayout& L, void* ws, float* rays_xyz, float* uvd, float* z_vals,
                       float* ball, int64_t* indices, int64_t* per_batch, int32_t* spb, int64_t* total, hipStream_t st) {
    size_t nb = (size_t)d.B * d.H * d.W;
    int32_t* cnt = (int32_t*)((char*)ws + L.cntOff);
    int32_t* offs = (int32_t*)((char*)ws + L.offOff);
    int32_t* bsum = (int32_t*)((char*)ws + L.bsumOff);
    unsigned g = (unsigned)((nb + 255) / 256);
    hipLaunchKernelGGL(k_counts<BB>, dim3(g), dim3(256), 0, st, d, cnt);
    LAUNCH_CHECK("k_counts");
    hipLaunchKernelGGL(k_scan_block, dim3(L.nBlocksScan), dim3(256), 0, st, nb, cnt, offs, bsum);
    LAUNCH_CHECK("k_scan_block");
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, st, L.nBlocksScan, bsum);
    LAUNCH_CHECK("k_scan_sums");
    hipLaunchKernelGGL(k_scan_add, dim3(g), dim3(256), 0, st, nb, bsum, offs);
    LAUNCH_CHECK("k_scan_add");
    hipLaunchKernelGGL(k_sample<BB>, dim3(g), dim3(256), 0, st, d, offs, bsum, L.nBlocksScan, rays_xyz, uvd, z_vals, ball,
                       indices, per_batch, spb, total);
    LAUNCH_CHECK("k_sample");
    hipLaunchKernelGGL(k_per_batch, dim3(1), dim3(64), 0, st, d.B, per_batch);
    LAUNCH_CHECK("k_per_batch");
    return GDB_OK;
}

extern "C" int gdb_sample(const GdbConfig* cfg, const GdbFrame* f, void* ws, float* rays_xyz, float* uvd, float* z_vals,
                          float* ball, int64_t* indices, int64_t* per_batch, int32_t* spb, int64_t* total, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, f, false); if (rc) return rc;
    if (!f->d_depth_range || !f->d_vol_range) return gdb_fail(GDB_E_BADARG, "frame has a NULL depth_range / vol_range pointer");
    if (!ws || !rays_xyz || !uvd || !z_vals || !ball || !indices || !per_batch || !spb || !total)
        return gdb_fail(GDB_E_BADARG, "NULL pointer");
    WsLayout L = ws_layout(*cfg, *f);
    DevFrame d = dev_frame(*cfg, *f, L, ws);
    hipStream_t st = (hipStream_t)stream_;
    switch (cfg->bundle_size) {
        case 1: return sample_impl<1>(d, L, ws, rays_xyz, uvd, z_vals, ball, indices, per_batch, spb, total, st);
        case 2: return sample_impl<4>(d, L, ws, rays_xyz, uvd, z_vals, ball, indices, per_batch, spb, total, st);
        default: return sample_impl<16>(d, L, ws, rays_xyz, uvd, z_vals, ball, indices, per_batch, spb, total, st);
    }
}

// ============================================================================================
// A4  encode
// ============================================================================================
// Batch item of compacted sample i, from the per-batch counts (bundle_sampler.py:318-319,370).
__device__ __forceinline__ int batch_of(int B, const int64_t* __restrict__ per_batch, int64_t i) {
    int64_t acc = 0;
    for (int bi = 0; bi < B; ++bi) { acc += per_batch[bi]; if (i < acc) return bi; }
    return B - 1;
}

// Voxel feature: 5-D grid_sample, trilinear / border / align_corners=False.  :322-324
// One thread per (sample, channel): consecutive samples sit on consecutive bundles, so the
// NCDHW volume is read along x.
__global__ void k_encode_vox(DevFrame f, const float* __restrict__ uvd, const int64_t* __restrict__ per_batch,
                             const int64_t* __restrict__ total, float* __restrict__ vox) {
    int64_t n = *total;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / GDB_CV; int c = (int)(t % GDB_CV);
    if (i >= n) return;
    int bi = batch_of(f.B, per_batch, i);
    float x = gs_coord(uvd[i * 3 + 0], f.W), y = gs_coord(uvd[i * 3 + 1], f.H), z = gs_coord(uvd[i * 3 + 2], f.D);
    float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    float wx = x - xf, wy = y - yf, wz = z - zf;
    int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
    const float* vol = f.feat_volume + ((size_t)bi * GDB_CV + c) * f.D * f.H * f.W;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                bool in = xx <= f.W - 1 && yy <= f.H - 1 && zz <= f.D - 1;
                float wgt = (dx ? wx : 1.f - wx) * (dy ? wy : 1.f - wy) * (dz ? wz : 1.f - wz);
                float val = vol[((size_t)min(zz, f.D - 1) * f.H + min(yy, f.H - 1)) * f.W + min(xx, f.W - 1)];
                acc += in ? val * wgt : 0.f;
            }
    vox[i * GDB_CV + c] = acc;
}

// Mip-mapped feature at texture coordinate (u,v) in [0,1]^2, level-of-detail `level`:
// restatement of nvdiffrast.torch.texture(..., mip_level_bias=level, boundary_mode='clamp',
// max_mip_level=L) as called at bundle_sampler.py:355-359.  Writes GDB_CFR floats.
__device__ __forceinline__ void tex_level(const DevFrame& f, const float* __restrict__ pyr, int l, float u, float v,
                                          float4 out[GDB_CP / 4]) {
    int W = f.lvlW[l], H = f.lvlH[l];
    int x0, x1, y0, y1; float fx, fy;
    tex_coord(u, W, x0, x1, fx);
    tex_coord(v, H, y0, y1, fy);
    const float4* base = (const float4*)(pyr + f.lvlOff[l]);  // chunk-planar: [chunk][y][x]
#pragma unroll
    for (int c = 0; c < GDB_CP / 4; ++c) {
        const float4* pl = base + (size_t)c * H * W;
        float4 top = lerp4(pl[(size_t)y0 * W + x0], pl[(size_t)y0 * W + x1], fx);
        float4 bot = lerp4(pl[(size_t)y1 * W + x0], pl[(size_t)y1 * W + x1], fx);
        out[c] = lerp4(top, bot, fy);
    }
}

__device__ __forceinline__ void tex_fetch(const DevFrame& f, const float* __restrict__ pyr, float u, float v, float level,
                                          float4 out[GDB_CP / 4]) {
    int l0, l1; float frac;
    mip_select(level, f.levels, l0, l1, frac);
    tex_level(f, pyr, l0, u, v, out);
    if (frac > 0.f) {
        float4 o1[GDB_CP / 4];
        tex_level(f, pyr, l1, u, v, o1);
#pragma unroll
        for (int c = 0; c < GDB_CP / 4; ++c) out[c] = lerp4(out[c], o1[c], frac);
    }
}

// One (view, sample): per-sub-ray RGB, mip-mapped feature, view-direction code.  :327-369
template <int BB>
__global__ void k_encode_views(DevFrame f, const float* __restrict__ rays_xyz, const float* __restrict__ ball_radii,
                               const int64_t* __restrict__ per_batch, const int64_t* __restrict__ total, int64_t n_alloc,
                               float* __restrict__ out) {
    constexpr int P = 3 * BB + GDB_CFR + 4;
    int64_t n = *total;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = blockIdx.y;
    if (i >= n) return;
    int bi = batch_of(f.B, per_batch, i);
    const float* sc = src_cam(f, bi, v);
    const float* tc = tar_cam(f, bi);
    float* o = out + ((size_t)v * n_alloc + i) * P;
    const float* img = f.src_images + ((size_t)bi * f.V + v) * 3 * f.Ho * f.Wo;

    float cw[3] = {0.f, 0.f, 0.f}, cc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < BB; ++s) {
        float p[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = rays_xyz[((size_t)i * 3 + c) * BB + s]; cw[c] += p[c]; }
        float cam[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            cam[r] = sc[S_E + 4 * r] * p[0] + sc[S_E + 4 * r + 1] * p[1] + sc[S_E + 4 * r + 2] * p[2] + sc[S_E + 4 * r + 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) cc[c] += cam[c];
        float im[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) im[r] = sc[S_K + 3 * r] * cam[0] + sc[S_K + 3 * r + 1] * cam[1] + sc[S_K + 3 * r + 2] * cam[2];
        float zc = fmaxf(im[2], 1e-6f);
        float gx = 2.f * (im[0] / zc) / (float)f.Wo - 1.f, gy = 2.f * (im[1] / zc) / (float)f.Ho - 1.f;
        float rgb[3];
        rgb_fetch(img, f.Ho, f.Wo, gx, gy, rgb);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * BB + s] = rgb[c];  // channel order c*b²+sub  :337
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { cw[c] = cw[c] / (float)BB; cc[c] = cc[c] / (float)BB; }

    float level = mip_level(cc[0], cc[1], cc[2], ball_radii[i], sc[S_PIXR]);
    float ci[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) ci[r] = sc[S_KS + 3 * r] * cc[0] + sc[S_KS + 3 * r + 1] * cc[1] + sc[S_KS + 3 * r + 2] * cc[2];
    float zc = fmaxf(ci[2], 1e-6f);
    float tu = ci[0] / zc / (float)f.W, tv = ci[1] / zc / (float)f.H;
    float4 ft[GDB_CP / 4];
    tex_fetch(f, f.pyr + ((size_t)bi * f.V + v) * f.pyrStride, tu, tv, level, ft);
    const float* ff = (const float*)ft;
#pragma unroll
    for (int c = 0; c < GDB_CFR; ++c) o[3 * BB + c] = ff[c];

    float td[3], sd[3], a[3], dif[3], dn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = cw[c] - tc[T_O + c];
    normalize3(a, td);
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = cw[c] - sc[S_C + c];
    normalize3(a, sd);
#pragma unroll
    for (int c = 0; c < 3; ++c) dif[c] = td[c] - sd[c];
    normalize3(dif, dn);
    o[3 * BB + GDB_CFR + 0] = dn[0]; o[3 * BB + GDB_CFR + 1] = dn[1]; o[3 * BB + GDB_CFR + 2] = dn[2];
    o[3 * BB + GDB_CFR + 3] = td[0] * sd[0] + td[1] * sd[1] + td[2] * sd[2];
}

extern "C" int gdb_encode(const GdbConfig* cfg, const GdbFrame* f, const void* ws, const float* rays_xyz, const float* uvd,
                          const float* ball, const int64_t* per_batch, const int64_t* total, int64_t n_alloc, float* out,
                          float* vox, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, f, true); if (rc) return rc;
    if (!ws || !rays_xyz || !uvd || !ball || !per_batch || !total || !out || !vox) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (n_alloc < 1 || n_alloc > (int64_t)f->B * f->H * f->W * cfg->max_num_samples)
        return gdb_fail(GDB_E_SHAPE, "n_alloc %lld outside 1..B*H*W*S_max", (long long)n_alloc);
    WsLayout L = ws_layout(*cfg, *f);
    DevFrame d = dev_frame(*cfg, *f, L, ws);
    hipStream_t st = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_encode_vox, dim3((unsigned)((n_alloc * GDB_CV + 255) / 256)), dim3(256), 0, st, d, uvd, per_batch, total, vox);
    LAUNCH_CHECK("k_encode_vox");
    dim3 g((unsigned)((n_alloc + 127) / 128), f->V);
    switch (cfg->bundle_size) {
        case 1: hipLaunchKernelGGL(k_encode_views<1>, g, dim3(128), 0, st, d, rays_xyz, ball, per_batch, total, n_alloc, out); break;
        case 2: hipLaunchKernelGGL(k_encode_views<4>, g, dim3(128), 0, st, d, rays_xyz, ball, per_batch, total, n_alloc, out); break;
        default: hipLaunchKernelGGL(k_encode_views<16>, g, dim3(128), 0, st, d, rays_xyz, ball, per_batch, total, n_alloc, out); break;
    }
    LAUNCH_CHECK("k_encode_views");
    return GDB_OK;
}
