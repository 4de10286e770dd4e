// Internal layouts and device helpers shared by the HIP translation units.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gdb_nerf_hip.h"

// ---- fixed network sizes (every config of the reference: dtu_pretrain.yaml:17-42) ----------
#define GDB_CF 16            // fpn.feat_dims[feat_level]
#define GDB_CFR (GDB_CF + 3) // feature ⊕ rgb channels of img_feat
#define GDB_CP 20            // GDB_CFR padded to a multiple of 4 floats: five 16-B chunks per texel; the pyramid stores
                             // each level chunk-planar, [chunk][y][x] of float4 (consecutive lanes read consecutive 16 B)
#define GDB_CV 8             // mvs.voxel_dim
#define GDB_HID 64           // nerf.nerf_hidden_dims
#define GDB_GF 32            // width of global_fc / agg
#define GDB_IM 16            // width of fc
#define GDB_HD (GDB_CV + GDB_IM)              // 24: [vox | im]
#define GDB_FV (GDB_CFR + 4)                  // 23: per-view tail [feat | rgb | dir]
#define GDB_W0IN (GDB_HID + GDB_HD + GDB_FV)  // 111

// ---- camera block in the workspace ------------------------------------------------------
// Per batch item: one target record, then V source records.
#define TAR_STRIDE 24
#define T_O 0      // rays_o (3)                      bundle_sampler.py:69
#define T_Z 3      // camera z axis in world (3)      :68
#define T_M 6      // R_c2w * K^-1, row-major 3x3     :70
#define T_PIXR 15  // 1/sqrt(fx fy pi)                :74
#define T_NEAR 16
#define T_FAR 17
#define T_MINIV 18 // minimum sample interval         :227-229
#define T_DISK 19  // b * pixel radius                :106
#define SRC_STRIDE 48
#define S_E 0      // w2c rows 0..2 (3x4)
#define S_K 12     // intrinsics 3x3
#define S_KS 21    // intrinsics with rows 0,1 divided by b   :311-312
#define S_C 30     // camera centre in world (3)              :305
#define S_PIXR 33  // 1/sqrt(fx/b fy/b pi)                    :313
#define S_IPIXR 34 // its reciprocal (fused kernel)
#define S_P 35     // K * w2c rows 0..2 (3x4), product formed in fp64 (fused kernel: world -> pixel in one step)


// ---- packed MLP weights, fp32 section (float offsets; every block 4-float aligned) --------
constexpr int pw_al(int x) { return (x + 3) / 4 * 4; }
constexpr int PW_VIEW_W = 0;                                   // (19,4)
constexpr int PW_VIEW_B = pw_al(PW_VIEW_W + GDB_CFR * 4);
constexpr int PW_GLOB_W = pw_al(PW_VIEW_B + GDB_CFR);          // (32,57)
constexpr int PW_GLOB_B = pw_al(PW_GLOB_W + GDB_GF * 3 * GDB_CFR);
constexpr int PW_AGG_W = pw_al(PW_GLOB_B + GDB_GF);            // (1,32)
constexpr int PW_AGG_B = pw_al(PW_AGG_W + GDB_GF);
constexpr int PW_FC_W = pw_al(PW_AGG_B + 1);                   // (16,32)
constexpr int PW_FC_B = pw_al(PW_FC_W + GDB_IM * GDB_GF);
constexpr int PW_LR0_W = pw_al(PW_FC_B + GDB_IM);              // (64,24)
constexpr int PW_LR0_B = pw_al(PW_LR0_W + GDB_HID * GDB_HD);
constexpr int PW_SIG_W = pw_al(PW_LR0_B + GDB_HID);            // (1,64)
constexpr int PW_SIG_B = pw_al(PW_SIG_W + GDB_HID);
constexpr int PW_W0_W = pw_al(PW_SIG_B + 1);                   // (64,111)
constexpr int PW_W0_B = pw_al(PW_W0_W + GDB_HID * GDB_W0IN);
constexpr int PW_W2_W = pw_al(PW_W0_B + GDB_HID);              // (1,64)
constexpr int PW_W2_B = pw_al(PW_W2_W + GDB_HID);
constexpr int PW_FH_W = pw_al(PW_W2_B + 1);                    // (8,64)
constexpr int PW_FH_B = pw_al(PW_FH_W + GDB_CV * GDB_HID);
constexpr int PW_FP32_FLOATS = (PW_FH_B + GDB_CV + 63) / 64 * 64;

struct WsLayout {
    int levels;             // mip levels built beyond level 0 (<= max_mipmap_level)
    int lvlH[GDB_MAX_MIP + 1], lvlW[GDB_MAX_MIP + 1];
    size_t lvlOff[GDB_MAX_MIP + 1];  // float offset of each level inside one (b,v) pyramid
    size_t pyrStride;                // floats per (b,v) pyramid
    size_t camsOff, pyrOff, cntOff, offOff, bsumOff, planOff;  // byte offsets in the workspace
    size_t total;
    int nBlocksScan;
    int planL, planMW;  // dense schedule: sample-offset window length, windows per bundle-map row at most (plan row = planMW + 2 ints)
    size_t smapOff;     // dense schedule: the compacted sample list itself, one uint32 per sample offset of a row
    int smapStride;     // entries per bundle-map row (planMW * planL + 32: a window's 32 lanes never read past it)
    size_t nwinOff;     // dense schedule: windows in use per bundle-map row (int32 per row, padded to whole int4s): the render waves
                        // turn a dense tile index into (row, window) with one scan of it, so the grid holds no empty windows between rows
    size_t nsampOff;    // flat schedule: samples per bundle-map row (int32 per row, padded to whole int4s), written by plan_row
    size_t sideOff, sideHdrOff;  // flat schedule: per window boundary the samples of the bundle that straddles it + a header (FLAT_* below)
    int flatMaxTiles;   // flat schedule: windows of 32 consecutive samples at most (every bundle at S_max), per launch
    size_t pyr16Off;    // the half-precision copy of the feature pyramid the GDB_PREC_F16 render gathers from (PYR16_* below):
                        // 2 bytes per float of the fp32 pyramid, same (batch, view) stride and level offsets in elements
    size_t img16Off;    // the half-precision RGBA copy of the source images (IMG16_* below) behind it: at IMG16_REL(...) from pyr16Off
    size_t boundsOff;   // gdb_prepare_rows: per (batch, view) the pyramid tiles / image rows a row strip's samples can reach (STRIP_BOUNDS ints)
    size_t colTmpOff, colWvOff;  // bundle_size 1 / 4 (the two-launch fused path): the list kernel's packed rows (N_b x 41) and its per
                                 // (sample slot, view) colour weights (N_b x S_max x V), read by k_bundle_colours; size 0 at bundle_size 2
};

// ---- half-precision pyramid (GDB_PREC_F16 only) ------------------------------------------------------------------------------
// One (batch, view) block is pyrStride halves, laid out per level l (element offset lvlOff[l], i.e. byte offset 2 lvlOff[l]) as three
// planes over the level's texels: plane 0 = 16 B per texel, halves of channels [0..3, 8..11]; plane 1 = 16 B, channels [4..7,
// 12..15]; plane 2 = 8 B, channels [16..19] (19 is padding).  Lane half h of the fused kernel owns channels 4h..4h+3, 8+4h..8+4h+3
// and 16+2h, 17+2h (the accumulator rows it owns): ONE 16-byte load of plane h and ONE 4-byte load of plane 2 per tap, against
// two 16-byte loads and an 8-byte load of the fp32 pyramid - a third fewer load instructions, half the bytes through L1.
// Values are the fp32 pyramid's, rounded to nearest half once (the mip levels are box-filtered in fp32 first).
#define PYR16_PLANE2(hw) (32u * (unsigned)(hw))   // byte offset of plane 2 inside a level of hw texels (plane 1 sits at 16 hw)
// Plane 2 is read as x PAIRS (one 16-byte load = channels 16..19 of texels x, x + 1: two loads per level instead of four): the pair of
// the last texel of the last level reaches 8 bytes past the pyramid, so the region ends in PYR16_PAD bytes that gdb_prepare zeroes.
#define PYR16_PAD 16
// ---- half-precision source images (GDB_PREC_F16 only) -------------------------------------------------------------------------
// Per (batch, view) Ho x Wo pixels of 8 bytes = halves (r, g, b, 0): an x pair of a colour tap is ONE 16-byte load where the planar
// fp32 images (B,V,3,Ho,Wo) take three 8-byte ones (12 -> 4 colour loads per sample, view and lane: the f16 kernels are bound by the
// texture addresser, profiles/r05/pmc_c5_f16_baseline.txt).  Values = the source colours rounded to nearest half once (they are in
// [0, 1]: <= 2.4e-4), written by k_prepare (GDB_PREP_PYR16).  It sits behind the half-precision pyramid, at a distance both the host
// and the kernels derive from the frame's sizes (no pointer of its own in the kernel arguments).
#define IMG16_REL(pyrStride, BV) ((((size_t)2 * (size_t)(pyrStride) * (size_t)(BV)) + PYR16_PAD + 255) / 256 * 256)

#define SCAN_BLOCK 1024
// gdb_prepare_rows: ints per (batch, view) of the strip's reach into a source view: [tile x lo, hi, tile y lo, hi (inclusive, 32 x 8-texel
// pyramid tiles), image row lo, hi (inclusive, source pixels), whole (1: the bound could not be formed - build everything), unused]
#define STRIP_BOUNDS 8
// flat schedule: floats per (sample, lane half) record of a straddling bundle: 20 pre-weight values (16 blended + 4 feat_head),
// alpha, the depth term, the boundary's header (map index, sample count) = 24 floats, padded to 32: ONE 128-byte line per record
// (the region is 256-byte aligned), so that no line of the side buffer is shared between two records - i.e. between two lanes, two
// waves or two window boundaries: a line is written by one lane (six 16-byte write-through stores) and read by one wave (the last
// arriver of its boundary), once per launch each, the read after the write has drained (at 96 bytes a
// record shared its lines with the next boundary's, which another wave may have pulled into its L2 before they were written)
#ifndef FLAT_REC
#define FLAT_REC 32
#endif

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Fixed sample counts (is_adaptive = 0) for which GDB_SCHED_AUTO takes the DENSE schedule at fp32 / split-f16 (the sample list is then
// S_max entries per bundle: the plan does not depend on the depth prior's values): more than 3 samples per bundle (or exactly 2) and at
// most 3 source views.  Measured at 512x640, V = 3 (profiles/r04/ab_fixed_counts_dense_vs_solo.txt, dense / segment wave, us): fp32 S 4 162 / 177,
// S 6 241 / 261, S 8 284 / 347; split-f16 S 8 190 / 235; but f16 S 8 139 / 137, S 4 84 / 71, and V = 5 (c5) fp32 2,429 / 1,870.
// gdb_prepare builds the plan for such configs too, so that a render told GDB_SCHED_PLAN_READY need not.
// (S_max 2: dense 77.3 / slot waves 84.7; S_max 3 stays on the slot waves: 117.1 / dense 121.2)
static inline bool gdb_fixed_counts_dense(const GdbConfig& c, int V) { return !c.is_adaptive && (c.max_num_samples > 3 || c.max_num_samples == 2) && V <= 3; }


static inline WsLayout ws_layout(const GdbConfig& c, const GdbFrame& f) {
    WsLayout L{};
    L.lvlH[0] = f.H; L.lvlW[0] = f.W; L.lvlOff[0] = 0;
    size_t acc = (size_t)f.H * f.W * GDB_CP;
    int lv = 0;
    while (lv < c.max_mipmap_level && lv < GDB_MAX_MIP) {
        int h = L.lvlH[lv], w = L.lvlW[lv];
        if (h < 2 || w < 2 || (h & 1) || (w & 1)) break;
        ++lv;
        L.lvlH[lv] = h / 2; L.lvlW[lv] = w / 2; L.lvlOff[lv] = acc;
        acc += (size_t)(h / 2) * (w / 2) * GDB_CP;
    }
    L.levels = lv;
    L.pyrStride = acc;
    size_t nb = (size_t)f.B * f.H * f.W;
    L.nBlocksScan = (int)((nb + SCAN_BLOCK - 1) / SCAN_BLOCK);
    size_t off = 0;
    L.camsOff = off; off = align_up(off + sizeof(float) * (size_t)f.B * (TAR_STRIDE + (size_t)f.V * SRC_STRIDE), 256);
    L.pyrOff = off;  off = align_up(off + sizeof(float) * L.pyrStride * f.B * f.V, 256);
    L.cntOff = off;  off = align_up(off + sizeof(int32_t) * nb, 256);
    L.offOff = off;  off = align_up(off + sizeof(int32_t) * nb, 256);
    L.bsumOff = off; off = align_up(off + sizeof(int32_t) * (L.nBlocksScan + 1), 256);
    // Dense-schedule plan (k_prepare writes it, k_render_dense reads it): per bundle-map row the first bundle of every window of
    // planL consecutive sample offsets.  A bundle holds at most S_max <= planL samples, so every window has a first bundle, and a
    // window's bundles hold at most planL + S_max - 1 = 32 samples: one wave's lanes.
    L.planL = 33 - c.max_num_samples;
    L.planMW = (int)(((size_t)f.W * c.max_num_samples + L.planL - 1) / L.planL);
    L.planOff = off; off = align_up(off + sizeof(int32_t) * (size_t)f.B * f.H * (L.planMW + 2), 256);
    // ... and the sample list of every row (bundle_sampler.py:182-189: bundle-major, sample-minor): entry s of a row names the
    // sample at offset s = [bundle x (16 bits) | slot k (8) | the bundle's count (8)], 0xFFFFFFFF past the row's last sample.
    // Window w's wave reads entries [planL w, planL w + 32): lane = sample, no per-wave count / scan / LDS map.
    L.smapStride = L.planMW * L.planL + 32;
    L.smapOff = off; off = align_up(off + sizeof(uint32_t) * (size_t)f.B * f.H * L.smapStride, 256);
    L.nwinOff = off; off = align_up(off + sizeof(int32_t) * ((size_t)f.B * f.H + 8), 256);
    // Flat schedule (GDB_SCHED_FLAT): the rows' sample lists read as ONE list, a wave = 32 consecutive samples of it; a bundle may
    // straddle two windows, and then both waves leave its samples' records here for the fix-up launch (k_flat_fix).
    L.nsampOff = off; off = align_up(off + sizeof(int32_t) * ((size_t)f.B * f.H + 8), 256);
    L.flatMaxTiles = (int)((size_t)f.B * (((size_t)f.H * f.W * c.max_num_samples + 31) / 32 + 2));   // per batch item: tiles + 1 boundaries
    L.sideHdrOff = off; off = align_up(off + sizeof(int32_t) * 2 * ((size_t)L.flatMaxTiles + 1), 256);
    L.sideOff = off; off = align_up(off + sizeof(float) * FLAT_REC * 2 * (size_t)c.max_num_samples * ((size_t)L.flatMaxTiles + 1), 256);
    L.pyr16Off = off; off += IMG16_REL(L.pyrStride, (size_t)f.B * f.V);
    L.img16Off = off; off = align_up(off + (size_t)8 * f.B * f.V * f.Ho * f.Wo + 16, 256);
    L.boundsOff = off; off = align_up(off + sizeof(int32_t) * STRIP_BOUNDS * (size_t)f.B * f.V, 256);
    L.colTmpOff = off; if (c.bundle_size != 2) off = align_up(off + sizeof(float) * nb * 41, 256);
    L.colWvOff = off;  if (c.bundle_size != 2) off = align_up(off + sizeof(float) * nb * (size_t)c.max_num_samples * f.V, 256);
    L.total = off;
    return L;
}

// Device-side view of one frame's prepared data.
struct DevFrame {
    int B, V, Ho, Wo, H, W, D, b;
    int S_max, adaptive, inv_depth, levels;
    int lvlH[GDB_MAX_MIP + 1], lvlW[GDB_MAX_MIP + 1];
    unsigned lvlOff[GDB_MAX_MIP + 1];
    unsigned pyrStride;
    float invW, invH;  // 1/W, 1/H of the bundle map (uniform reciprocals the fused kernel would otherwise recompute per view)
    int planL, planMW;
    const int* plan;   // dense-schedule plan rows: [nwin, first bundle of window 0..nwin-1, W]
    const unsigned* smap; int smapStride;  // dense-schedule sample list, smapStride entries per row
    const int* nwin;   // dense-schedule windows in use per row
    const int* nsamp;  // flat schedule: samples per row
    float* side; int* side_hdr;  // flat schedule: straddling-bundle records and headers (written by the render, read by the fix-up)
    const float* cams;
    const float* pyr;
    const void* pyr16;  // half-precision pyramid (GDB_PREC_F16 gathers from it; built by gdb_prepare_ex or by the render call)
    const float* src_images;
    const float* feat_volume;
    const float* depth_range;
    const float* vol_range;
};

static inline DevFrame dev_frame(const GdbConfig& c, const GdbFrame& f, const WsLayout& L, const void* ws) {
    DevFrame d{};
    d.B = f.B; d.V = f.V; d.Ho = f.Ho; d.Wo = f.Wo; d.H = f.H; d.W = f.W; d.D = f.D; d.b = c.bundle_size;
    d.S_max = c.max_num_samples; d.adaptive = c.is_adaptive; d.inv_depth = c.inv_depth; d.levels = L.levels;
    for (int i = 0; i <= GDB_MAX_MIP; ++i) { d.lvlH[i] = L.lvlH[i]; d.lvlW[i] = L.lvlW[i]; d.lvlOff[i] = (unsigned)L.lvlOff[i]; }
    d.pyrStride = (unsigned)L.pyrStride;
    d.invW = 1.f / (float)f.W; d.invH = 1.f / (float)f.H;
    d.planL = L.planL; d.planMW = L.planMW;
    d.plan = (const int*)((const char*)ws + L.planOff);
    d.smap = (const unsigned*)((const char*)ws + L.smapOff); d.smapStride = L.smapStride;
    d.nwin = (const int*)((const char*)ws + L.nwinOff);
    d.nsamp = (const int*)((const char*)ws + L.nsampOff);
    d.side = (float*)((char*)ws + L.sideOff); d.side_hdr = (int*)((char*)ws + L.sideHdrOff);
    d.cams = (const float*)((const char*)ws + L.camsOff);
    d.pyr = (const float*)((const char*)ws + L.pyrOff);
    d.pyr16 = (const void*)((const char*)ws + L.pyr16Off);
    d.src_images = f.d_src_images; d.feat_volume = f.d_feat_volume;
    d.depth_range = f.d_depth_range; d.vol_range = f.d_vol_range;
    return d;
}

#ifdef __HIPCC__
// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ const float* tar_cam(const DevFrame& f, int bi) {
    return f.cams + (size_t)bi * (TAR_STRIDE + f.V * SRC_STRIDE);
}
__device__ __forceinline__ const float* src_cam(const DevFrame& f, int bi, int v) {
    return tar_cam(f, bi) + TAR_STRIDE + v * SRC_STRIDE;
}

// Unnormalised ray direction of the pixel centre (x, y): [x, y, 1] * M^T.  bundle_sampler.py:70
__device__ __forceinline__ void ray_dir(const float* __restrict__ M, float x, float y, float d[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = M[3 * i] * x + M[3 * i + 1] * y + M[3 * i + 2];
}

// torch grid_sample coordinate, align_corners=False, padding_mode='border'.
__device__ __forceinline__ float gs_coord(float g, int size) {
    float x = ((g + 1.f) * (float)size - 1.f) / 2.f;
    return fminf(fmaxf(x, 0.f), (float)(size - 1));
}

// Per-bundle sample count.  bundle_sampler.py:152,179
__device__ __forceinline__ int sample_count(float nearv, float farv, float min_iv, int S_max, int adaptive) {
    if (!adaptive) return S_max;
    float c = ceilf(fabsf(farv - nearv) / min_iv);
    c = fminf(fmaxf(c, 1.f), (float)S_max);  // NaN -> 1 like torch.clamp? (clamp keeps NaN; counts of NaN are undefined upstream)
    return (int)c;
}

// Sphere radius per unit distance.  bundle_sampler.py:262
__device__ __forceinline__ float ball_unit(float disk, float cosv) {
    float t = sqrtf(fmaxf(1.f / (cosv * cosv) - 1.f, 1e-12f)) - disk;
    return disk * cosv / sqrtf(t * t + 1.f);
}

__device__ __forceinline__ float ball_unit_fast(float disk, float cosv) {
    float ic = __builtin_amdgcn_rcpf(cosv);
    float t = __builtin_amdgcn_sqrtf(fmaxf(ic * ic - 1.f, 1e-12f)) - disk;
    return disk * cosv * __builtin_amdgcn_rsqf(t * t + 1.f);
}

// Footprint -> mip level.  bundle_sampler.py:343-348
__device__ __forceinline__ float mip_level(float cx, float cy, float cz, float ball, float src_pixr) {
    float dist = sqrtf(cx * cx + cy * cy + cz * cz);
    float q = dist / cz;
    float sec2 = q * q;
    float r = dist / ball;
    float a = sqrtf(fmaxf(r * r - 1.f, 1e-12f));
    float c = sqrtf(fmaxf(sec2 - 1.f, 1e-12f));
    return log2f((sec2 / (a + c)) / src_pixr);
}

// nvdiffrast level selection with mip_level_bias only: clamp to [0, L]; NaN -> 0.
__device__ __forceinline__ void mip_select(float level, int L, int& l0, int& l1, float& frac) {
    float lv = fminf(fmaxf(level, 0.f), (float)L);  // fmaxf(NaN, 0) = 0
    l0 = (int)floorf(lv);
    l1 = min(l0 + 1, L);
    frac = lv - (float)l0;
}

// Texel-space coordinate of one mip level with clamp-to-edge addressing.
__device__ __forceinline__ void tex_coord(float u, int size, int& i0, int& i1, float& f) {
    float x = fminf(fmaxf(u * (float)size - 0.5f, 0.f), (float)(size - 1));
    float xf = floorf(x);
    i0 = (int)xf;
    i1 = min(i0 + 1, size - 1);
    f = x - xf;
}

__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float t) {
    return make_float4(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w));
}

__device__ __forceinline__ float softplus_t20(float x) {  // nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}

// Geometry of one bundle, shared by the mirror below and by the fused kernel.
template <int BB>  // BB = b*b
struct Bundle {
    float o[3];
    float d[BB][3];    // sub-ray directions, order by*b+bx            bundle_sampler.py:100
    float u, v;        // mean normalised pixel coordinate              :104
    float nearv, farv, vnear, vfar;
    float unit;        // sphere radius per unit distance               :262
    int count;
};

// FAST (fused kernel only): divisions become v_rcp multiplies.  The sample count keeps its IEEE division either way.
template <bool FAST> __device__ __forceinline__ float gdiv(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }

// Wave-uniform data (camera blocks, bias scalars) read through the constant address space: with a uniform
// address the load is an s_load into SGPRs (scalar cache, lgkmcnt) instead of a vector-memory round trip.
// Only for memory no kernel in flight writes (the camera block is written by k_prepare, an earlier launch).
typedef const float __attribute__((address_space(4))) kfloat;
__device__ __forceinline__ const kfloat* kptr(const float* p) { return (const kfloat*)p; }

// tc: the target camera block, either in global memory (tar_cam) or a register copy of it.
// near/far of the depth prior and of the cost volume at bundle (h, w): the only per-lane loads of a bundle
__device__ __forceinline__ void load_ranges(const DevFrame& f, int bi, int h, int w, float r[4]) {
    size_t hw = (size_t)f.H * f.W;
    unsigned p = (unsigned)(h * f.W + w);  // 32-bit offset inside one (H, W) plane: with a uniform bi the plane base stays scalar
    r[0] = (f.depth_range + ((size_t)bi * 2) * hw)[p]; r[1] = (f.depth_range + ((size_t)bi * 2 + 1) * hw)[p];
    r[2] = (f.vol_range + ((size_t)bi * 2) * hw)[p];   r[3] = (f.vol_range + ((size_t)bi * 2 + 1) * hw)[p];
}

// pre: ranges already loaded by the caller (load_ranges), or nullptr
// COUNT: derive the sample count from the ranges (bundle_sampler.py:179); false leaves q.count to the caller (the dense schedule
// reads it from the plan's sample list)
// (contraction off in the three geometry functions below: the mirrors' translation units have it off anyway; in the fused kernels'
// (-ffp-contract=fast) it keeps every instantiation on the same bits - gdb_fused.hip, tex_coord_f - and the fused multiply-adds that
// path wants are written out under FAST)
template <int BB, bool FAST = false, bool COUNT = true>
__device__ __forceinline__ void load_bundle(const DevFrame& f, const float* __restrict__ tc, int bi, int h, int w, Bundle<BB>& q,
                                            const float* pre = nullptr) {
#pragma clang fp contract(off)
    constexpr int b = BB == 1 ? 1 : (BB == 4 ? 2 : 4);
    float sum[3] = {0.f, 0.f, 0.f};
    float su = 0.f, sv = 0.f;
#pragma unroll
    for (int by = 0; by < b; ++by)
#pragma unroll
        for (int bx = 0; bx < b; ++bx) {
            float x = (float)(w * b + bx) + 0.5f, y = (float)(h * b + by) + 0.5f;
            if constexpr (FAST) {
#pragma unroll
                for (int i = 0; i < 3; ++i) q.d[by * b + bx][i] = fmaf(tc[T_M + 3 * i], x, fmaf(tc[T_M + 3 * i + 1], y, tc[T_M + 3 * i + 2]));
            } else ray_dir(tc + T_M, x, y, q.d[by * b + bx]);
#pragma unroll
            for (int i = 0; i < 3; ++i) sum[i] += q.d[by * b + bx][i];
            su += gdiv<FAST>(2.f * x, (float)f.Wo) - 1.f;
            sv += gdiv<FAST>(2.f * y, (float)f.Ho) - 1.f;
        }
    float md[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { md[i] = sum[i] / (float)BB; q.o[i] = tc[T_O + i]; }
    q.u = su / (float)BB; q.v = sv / (float)BB;
    float nrm = FAST ? sqrtf(fmaf(md[2], md[2], fmaf(md[1], md[1], md[0] * md[0]))) : sqrtf(md[0] * md[0] + md[1] * md[1] + md[2] * md[2]);
    float cosv = FAST ? gdiv<FAST>(fmaf(md[2], tc[T_Z + 2], fmaf(md[1], tc[T_Z + 1], md[0] * tc[T_Z])), nrm)
                      : gdiv<FAST>(md[0] * tc[T_Z] + md[1] * tc[T_Z + 1] + md[2] * tc[T_Z + 2], nrm);
    q.unit = FAST ? ball_unit_fast(tc[T_DISK], cosv) : ball_unit(tc[T_DISK], cosv);
    float n0 = pre ? pre[0] : 0.f, f0 = pre ? pre[1] : 0.f, vn = pre ? pre[2] : 0.f, vf = pre ? pre[3] : 0.f;
    if (!pre) {
        float r[4];
        load_ranges(f, bi, h, w, r);
        n0 = r[0]; f0 = r[1]; vn = r[2]; vf = r[3];
    }
    if (f.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; vn = 1.f / vn; vf = 1.f / vf; }  // :224-226 (IEEE: feeds the sample count)
    q.nearv = n0; q.farv = f0; q.vnear = vn; q.vfar = vf;
    q.count = COUNT ? sample_count(n0, f0, tc[T_MINIV], f.S_max, f.adaptive) : 1;
}

template <int BB, bool FAST = false>
__device__ __forceinline__ void load_bundle(const DevFrame& f, int bi, int h, int w, Bundle<BB>& q) {
    load_bundle<BB, FAST>(f, tar_cam(f, bi), bi, h, w, q);
}

// The bundle's CENTRE ray alone, for any bundle size b = f.b (the fused kernels at bundle_size 1 / 4, round 6): the mean of the b^2
// sub-ray directions is the direction through the mean pixel (the rays are linear in the pixel: bundle_sampler.py:67-71, :99), the
// mean of their normalised coordinates likewise (:104), and the mean of the sub-ray points o + d z is o + mean(d) z (:254-256) - the
// same quantities the b^2-ray form sums up, to fp32 rounding.  q.d[0] = that mean direction; bundle_sample<1> then gives the sample's
// centre point, its distance and ball radius.  The sub-ray colours are gathered elsewhere (k_bundle_colours).
template <bool COUNT = true>
__device__ __forceinline__ void load_bundle_center(const DevFrame& f, const float* __restrict__ tc, int bi, int h, int w, Bundle<1>& q,
                                                   const float* pre = nullptr) {
#pragma clang fp contract(off)
    const float hb = 0.5f * (float)f.b;
    const float x = (float)(w * f.b) + hb, y = (float)(h * f.b) + hb;   // mean of (w b + bx + 0.5), bx = 0 .. b - 1: exact in fp32
#pragma unroll
    for (int i = 0; i < 3; ++i) q.d[0][i] = fmaf(tc[T_M + 3 * i], x, fmaf(tc[T_M + 3 * i + 1], y, tc[T_M + 3 * i + 2]));
#pragma unroll
    for (int i = 0; i < 3; ++i) q.o[i] = tc[T_O + i];
    q.u = gdiv<true>(2.f * x, (float)f.Wo) - 1.f;
    q.v = gdiv<true>(2.f * y, (float)f.Ho) - 1.f;
    const float* md = q.d[0];
    const float nrm = sqrtf(fmaf(md[2], md[2], fmaf(md[1], md[1], md[0] * md[0])));
    const float cosv = gdiv<true>(fmaf(md[2], tc[T_Z + 2], fmaf(md[1], tc[T_Z + 1], md[0] * tc[T_Z])), nrm);
    q.unit = ball_unit_fast(tc[T_DISK], cosv);
    float n0 = pre ? pre[0] : 0.f, f0 = pre ? pre[1] : 0.f, vn = pre ? pre[2] : 0.f, vf = pre ? pre[3] : 0.f;
    if (!pre) {
        float r[4];
        load_ranges(f, bi, h, w, r);
        n0 = r[0]; f0 = r[1]; vn = r[2]; vf = r[3];
    }
    if (f.inv_depth) { n0 = 1.f / n0; f0 = 1.f / f0; vn = 1.f / vn; vf = 1.f / vf; }
    q.nearv = n0; q.farv = f0; q.vnear = vn; q.vfar = vf;
    q.count = COUNT ? sample_count(n0, f0, tc[T_MINIV], f.S_max, f.adaptive) : 1;
}

// Mid depth of sample k of a bundle (in disparity when inv_depth): the one value the dense schedule's composite derives a second
// time from the depth prior instead of keeping it in a register across gather + MLP.  bundle_sampler.py:183, :246
template <bool FAST>
__device__ __forceinline__ float sample_mid(float nearv, float farv, int count, int k) {
#pragma clang fp contract(off)
    float step = gdiv<FAST>(farv - nearv, (float)count);
    float t0 = FAST ? fmaf(step, (float)k, nearv) : nearv + step * (float)k, t1 = FAST ? fmaf(step, (float)(k + 1), nearv) : nearv + step * (float)(k + 1);
    return 0.5f * (t0 + t1);
}

// One sample of a bundle: mid depth, normalised volume depth, sub-ray points, sphere radius.
template <int BB, bool FAST = false>
__device__ __forceinline__ void bundle_sample(const DevFrame& f, const Bundle<BB>& q, int k, float& z, float& dnorm,
                                              float xyz[BB][3], float ctr[3], float& ball) {
#pragma clang fp contract(off)
    z = sample_mid<FAST>(q.nearv, q.farv, q.count, k);                           // :183, :246
    dnorm = gdiv<FAST>(2.f * (z - q.vnear), q.vfar - q.vnear) - 1.f;             // :247
    if (f.inv_depth) z = gdiv<FAST>(1.f, z);                                     // :250-251
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < BB; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) { xyz[r][i] = FAST ? fmaf(q.d[r][i], z, q.o[i]) : q.o[i] + q.d[r][i] * z; s[i] += xyz[r][i]; }  // :255
    float dd = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { ctr[i] = s[i] / (float)BB; float e = ctr[i] - q.o[i]; dd = FAST ? fmaf(e, e, dd) : dd + e * e; }  // :256,:259
    ball = sqrtf(dd) * q.unit;                                                   // :263
}


// Bilinear RGB at a projected point: 4-D grid_sample, border, align_corners=False.  :336
__device__ __forceinline__ void rgb_fetch(const float* __restrict__ img, int Ho, int Wo, float gx, float gy, float rgb[3]) {
    float x = gs_coord(gx, Wo), y = gs_coord(gy, Ho);
    float xf = floorf(x), yf = floorf(y);
    float wx = x - xf, wy = y - yf, ex = 1.f - wx, ey = 1.f - wy;
    int x0 = (int)xf, y0 = (int)yf;
    bool inx = x0 + 1 <= Wo - 1, iny = y0 + 1 <= Ho - 1;
    int x1 = min(x0 + 1, Wo - 1), y1 = min(y0 + 1, Ho - 1);
    float w00 = ex * ey, w10 = inx ? wx * ey : 0.f, w01 = iny ? ex * wy : 0.f, w11 = (inx && iny) ? wx * wy : 0.f;
    size_t plane = (size_t)Ho * Wo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = img + c * plane;
        rgb[c] = p[(size_t)y0 * Wo + x0] * w00 + p[(size_t)y0 * Wo + x1] * w10 + p[(size_t)y1 * Wo + x0] * w01 +
                 p[(size_t)y1 * Wo + x1] * w11;
    }
}

__device__ __forceinline__ void normalize3(const float a[3], float o[3]) {  // F.normalize, eps 1e-12
    float n = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-12f);
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}

#endif
