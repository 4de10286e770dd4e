/*
 * gdb_nerf_hip.h — C ABI of libgdbnerf_hip.so: the GDB-NeRF depth-guided bundle-sampling hot
 * path as HIP kernels for gfx950 (MI355X).
 *
 * The reference (KLMAV-CUC/GDB-NeRF) is pure Python; it has no FFI of its own.  Each entry
 * point below replaces one Python operator of the reference (file:line relative to the
 * reference root) or one of the two third-party CUDA ops that operator calls.  The
 * reference-side binding a maintainer would add is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.
 *  - Every pointer named d_* is DEVICE memory owned by the caller; the library never
 *    allocates, frees or retains device memory.  h_* pointers are host memory.
 *  - All tensors are contiguous float32 in the reference's own layouts unless stated.
 *  - Every function only ENQUEUES work on `stream` (a hipStream_t passed as void*); nothing
 *    synchronises with the host.  Data-dependent sample counts stay on the device.
 *  - Return value: GDB_OK (0) or a negative GdbStatus; gdb_last_error() returns a
 *    thread-local message for the last failure on the calling thread.  Shape / config
 *    violations are rejected before any launch.  The operator mirrors are NaN-transparent
 *    like the reference; the fused kernel at GDB_PREC_F16 converts activations to f16
 *    (|x| > 65504 becomes inf); at GDB_PREC_F32 it computes the MLP in fp32 throughout.
 *  - No process-global mutable state: every function is reentrant; the only per-thread state
 *    is the gdb_last_error() message.
 */
#ifndef GDB_NERF_HIP_H
#define GDB_NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDB_ABI_VERSION 7

typedef enum GdbStatus {
    GDB_OK = 0,
    GDB_E_BADARG = -1,   /* null pointer, non-positive size, unsupported option */
    GDB_E_SHAPE = -2,    /* sizes inconsistent with each other or with the config */
    GDB_E_HIP = -3,      /* a HIP runtime call failed (message carries hipGetErrorString) */
    GDB_E_WORKSPACE = -4 /* workspace too small */
} GdbStatus;

/* Knobs of the path; names follow the reference's YAML keys (configs/dtu_pretrain.yaml:17-42,
 * read at networks/gdb_nerf/network.py:29-47). */
typedef struct GdbConfig {
    int32_t bundle_size;      /* nerf.bundle_size: b, power of two, 1..4 */
    int32_t max_num_samples;  /* nerf.max_num_samples: S_max, 1..GDB_MAX_SAMPLES */
    int32_t is_adaptive;      /* nerf.is_adaptive */
    int32_t inv_depth;        /* mvs.inv_depth[-1] */
    int32_t global_num_depth; /* nerf.global_num_depth */
    int32_t max_mipmap_level; /* nerf.max_mipmap_level, 0..GDB_MAX_MIP */
    int32_t feat_dim;         /* fpn.feat_dims[feat_level]: C_f (img_feat carries C_f+3 channels) */
    int32_t voxel_dim;        /* mvs.voxel_dim: C_v */
    int32_t hid_dim;          /* nerf.nerf_hidden_dims */
    int32_t viewdir_agg;      /* nerf.viewdir_agg */
} GdbConfig;

#define GDB_MAX_SAMPLES 16
#define GDB_MAX_MIP 3
#define GDB_MAX_VIEWS 8

/* One batch of hot-path inputs (Network.forward, network.py:114-166). */
typedef struct GdbFrame {
    int32_t B, V;             /* batch, source views */
    int32_t Ho, Wo;           /* target / source image size */
    int32_t H, W;             /* bundle map size: Ho/b, Wo/b */
    int32_t D;                /* depth planes of feat_volume */
    const float* d_src_images;  /* (B,V,3,Ho,Wo) */
    const float* d_img_feat;    /* (B,V,C_f+3,H,W): FPN level ⊕ downsampled RGB, network.py:159-164 */
    const float* d_feat_volume; /* (B,C_v,D,H,W) */
    const float* d_depth_range; /* (B,2,H,W) near/far of the depth prior */
    const float* d_vol_range;   /* (B,2,H,W) near/far of the cost volume */
    const float* d_src_exts;    /* (B,V,4,4) world-to-camera */
    const float* d_src_ints;    /* (B,V,3,3) */
    const float* d_tar_exts;    /* (B,4,4) world-to-camera */
    const float* d_tar_ints;    /* (B,3,3) */
    const float* d_near_far;    /* (B,2) scene near/far */
} GdbFrame;

/* ---- library ------------------------------------------------------------------------ */
int gdb_abi_version(void);
const char* gdb_last_error(void);

/* Bytes of device workspace gdb_prepare() needs for a frame of this shape (camera block,
 * channel-last feature pyramid, per-bundle counts/offsets). */
int gdb_workspace_bytes(const GdbConfig* cfg, const GdbFrame* shape, size_t* out_bytes);

/* Where gdb_prepare() puts the feature pyramid inside the workspace, for callers that want to read it
 * back (it is what nvdiffrast builds inside texture(): bundle_sampler.py:355-359).  Per (batch, view) the
 * pyramid is [level][chunk 0..4][y][x] of 16-byte texel chunks (channels 4*chunk .. 4*chunk+3 of the
 * C_f+3 = 19 channels, channel 19 zero).  out[0] = byte offset of the first pyramid, out[1] = floats per
 * (batch, view) pyramid, out[2] = number of mip levels built beyond level 0 (stops at the first level with
 * an odd extent), out[3 + l] = float offset of level l inside one pyramid, l = 0..3. */
int gdb_pyramid_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[7]);

/* Where the plan of the dense schedule (GDB_SCHED_DENSE) lives, for callers / tests that want to read it: per bundle-map row
 * (B*H rows) out[1] int32 values [n_windows, first sample offset of window 0 .. n_windows-1, the row's sample total].  A window
 * is a run of WHOLE consecutive bundles holding at most 32 samples (offsets = exclusive prefix of the per-bundle sample counts
 * along the row, bundle_sampler.py:179-189) - one wave of the dense kernel, lane = sample.  Rows of 1024 .. 4095 sample offsets
 * (W * S_max) are cut greedily (a window ends only where the next bundle would not fit); shorter and longer rows at fixed offsets: window w =
 * the bundles whose first sample offset falls into [out[2] * w, out[2] * (w + 1)), out[2] = 33 - S_max.  out[0] = byte offset of
 * the first row record.  Built by gdb_prepare FROM THE CONTENTS OF d_depth_range AT THAT TIME when the frame carries it and the
 * config is adaptive; a dense render that is not told the plan is current (GDB_SCHED_PLAN_READY) rebuilds it first. */
int gdb_dense_plan_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[3]);
/* ... and the compacted sample list beside it (bundle_sampler.py:182-189: bundle-major, sample-minor): per bundle-map row out[1]
 * uint32 entries, entry s = the sample at offset s of the row as [bundle x | slot k << 16 | count of the bundle << 24],
 * 0xFFFFFFFF past the row's last sample.  out[0] = byte offset of the first row.  Window w of the plan is the wave that reads
 * the entries [start_w, start_{w+1}) of its row. */
int gdb_dense_map_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[2]);

/* ---- MLP weights -------------------------------------------------------------------- */
/* Number of floats in the packed weight buffer for cfg (fp32 section + MFMA-fragment
 * section). */
int gdb_packed_weight_floats(const GdbConfig* cfg, size_t* out_floats);
/* Pack the 18 NeRF tensors (state-dict order of networks/gdb_nerf/nerf.py:20-56:
 * view_fc.0, global_fc.0, agg_w_fc.0, fc.0, lr0.0, sigma.0, weight.0, weight.2,
 * feat_head.0; each .weight (out,in) row-major then .bias) into h_out.  Host only; the
 * caller uploads h_out to the device once per checkpoint.  If viewdir_agg is 0 the first
 * two pointers may be NULL. */
int gdb_pack_weights(const GdbConfig* cfg, const float* const h_tensors[18], float* h_out);

/* ---- per-frame preparation ---------------------------------------------------------- */
/* Camera block (matrix inverses, ray matrix, pixel radii: bundle_sampler.py:67-74,304-313)
 * and the channel-last mip pyramid of img_feat (what nvdiffrast.texture builds on every
 * call, bundle_sampler.py:355-359).  d_tar_exts, d_tar_ints, d_near_far are required; the source
 * side (d_src_exts + d_src_ints, d_img_feat) may be NULL when only gdb_build_rays / gdb_sample follow.
 * The workspace also reserves room for the per-bundle
 * sample counts and their exclusive scan, which gdb_sample fills
 * (bundle_sampler.py:179,182-189).
 * What gdb_prepare READS: the camera matrices, d_near_far, d_img_feat (and d_src_images in the _fpn form) - and, when the frame
 * carries d_depth_range and the config is adaptive, THE CONTENTS OF d_depth_range AS THEY ARE AT THIS CALL, from which it builds
 * the dense schedule's per-row plan and compacted sample list (gdb_dense_plan_layout / gdb_dense_map_layout).  A render call
 * uses that plan only when told it is current (GDB_SCHED_PLAN_READY); otherwise it rebuilds it from the frame it is given. */
int gdb_prepare(const GdbConfig* cfg, const GdbFrame* frame, void* d_workspace, size_t workspace_bytes,
                void* stream);

/* The same from the FPN output (next row N3): d_fpn_feat (B,V,C_f,H,W) is the feature level alone; the three colour
 * channels the reference concatenates (network.py:159-164: F.interpolate(src_images, size=(H,W), mode='bilinear',
 * align_corners=False)) are resampled from frame->d_src_images inside the same launch.  frame->d_img_feat is ignored. */
int gdb_prepare_fpn(const GdbConfig* cfg, const GdbFrame* frame, const float* d_fpn_feat, void* d_workspace,
                    size_t workspace_bytes, void* stream);

/* ---- operator mirrors (one per reference method) ------------------------------------ */
/* gdb_prepare / gdb_prepare_fpn (d_fpn_feat may be NULL: the gdb_prepare form) with options.  ABI v5.
 *   GDB_PREP_PYR16  also write the HALF-PRECISION copy of the feature pyramid, in the same launch, from the same registers: what a
 *                   GDB_PREC_F16 render gathers its feature taps from (16 + 4 bytes per tap and lane instead of 16 + 16 + 8, two
 *                   load instructions instead of three; values = the fp32 pyramid's, rounded to nearest half once, every mip level
 *                   box-filtered in fp32 first).  Replaces nothing of the reference: nvdiffrast rebuilds its fp32 mip stack inside
 *                   every texture() call (bundle_sampler.py:355-359); this is a second, narrower copy for the opt-in fast path.
 *                   Pass GDB_SCHED_PYR16_READY to the render calls of this frame.
 *   GDB_PREP_PYR16_ONLY  (with GDB_PREP_PYR16) write ONLY the half-precision pyramid: the fp32 pyramid in the workspace is left as
 *                   it was (26 of the 58 MB k_prepare moves at 512x640, V = 3).  Only GDB_PREC_F16 fused renders may follow; gdb_encode, the
 *                   f32 / split-f16 renders and gdb_build_pyr16 read the fp32 pyramid and need a prepare without this flag first
 *                   (like GDB_SCHED_PLAN_READY this is the caller's promise: the library keeps no state to check it against).
 *   (GDB_PREP_PYR16, in full: besides the pyramid it writes a HALF-PRECISION RGBA COPY OF frame->d_src_images behind it - 8 bytes per
 *                   pixel, halves (r, g, b, 0) - from which the GDB_PREC_F16 kernels read their colour taps (one 16-byte load per x pair of a
 *                   tap row).  So with this flag the colours a later f16 render sees are those of d_src_images AT THIS CALL:
 *                   GDB_SCHED_PYR16_READY is the promise that neither d_img_feat / d_fpn_feat nor d_src_images have changed since.
 *                   A frame that carries a feature map but no d_src_images is refused with this flag (GDB_E_BADARG).  The copy is made
 *                   for bundle_size 2 (the fused kernels' size) only; gdb_pyramid16_layout does not describe it: it starts at the
 *                   first 256-byte boundary behind the B * V pyramid blocks + 16 bytes of padding.)
 *   GDB_PREP_SOURCES_READY  (ABI v7) the caller vouches that everything in the workspace that depends on the SOURCE views alone - the fp32
 *                   pyramid, and with GDB_PREP_PYR16 the half-precision pyramid and image copy - was built by an earlier gdb_prepare*
 *                   call on this workspace, with the same flags, from d_img_feat / d_fpn_feat / d_src_images whose contents have not
 *                   changed: this call rebuilds the camera block (target AND source records) and the list schedules' plan only - what
 *                   a sweep of target views over fixed source views needs per view (bundle_sampler.py:304-313 recomputes the camera
 *                   terms per call; nvdiffrast rebuilds the mip stack per call, :355-359 - this flag is where this library does not).
 *                   Like the two render-side promises the library keeps no state to check it against. */
#define GDB_PREP_PYR16 1
#define GDB_PREP_PYR16_ONLY 2
#define GDB_PREP_SOURCES_READY 4
/* gdb_prepare_rows only: whether a strip smaller than the frame builds just its reach of the source-only products (see there).  Neither
 * flag: by size - the bound costs a launch of its own ahead of the tiles (~8 us of stream time), which pays once the whole-frame tile
 * work moves >= 128 MB (1200x1600 with 5 views at f16: 95 -> 46 us per rank at 8 ranks; at 512x640 it would cost 3 us more than it saves:
 * profiles/r06/time_prepare_rows_world8.json). */
#define GDB_PREP_STRIP_REACH 8   /* always build the strip's reach only */
#define GDB_PREP_STRIP_WHOLE 16  /* always build the whole pyramid (a later render of other rows stays valid) */
#define GDB_PREP_ALL (GDB_PREP_PYR16 | GDB_PREP_PYR16_ONLY | GDB_PREP_SOURCES_READY | GDB_PREP_STRIP_REACH | GDB_PREP_STRIP_WHOLE)
int gdb_prepare_ex(const GdbConfig* cfg, const GdbFrame* frame, const float* d_fpn_feat, int32_t flags, void* d_workspace,
                   size_t workspace_bytes, void* stream);

/* gdb_prepare_ex for a rank that renders ONE row strip of the frame (multi-GPU row sharding; ABI v6): the camera block as
 * ever, the list schedules' plan for the bundle-map rows [row_begin, row_end) of every batch item only - the rows outside the strip are
 * another rank's - and, when the strip is smaller than the frame (round 6), only the PART of the source-only products the strip's samples
 * can reach: per (batch, view) a launch of its own ahead of the tiles (k_strip_bounds) takes the extremes of the strip's depth prior
 * (a sample's depth lies between its bundle's near and far, bundle_sampler.py:122-191), projects the 8 vertices of the convex body
 * { o + d(x, y) z : (x, y) in the strip's pixel rectangle, z in [z_min, z_max] } (bundle_sampler.py:67-71, :254-256) into the view and
 * builds only the 32 x 8-texel pyramid tiles (all mip levels) and the image rows (half-precision copy) inside that box + the mip /
 * bilinear margins; a view it cannot bound (a vertex behind or near the camera plane, a non-finite or non-positive depth) is built
 * whole.  It reads d_depth_range, so the frame must carry it.  GDB_SCHED_PLAN_READY holds for render calls whose strip lies inside
 * [row_begin, row_end); RENDERS OF ROWS OUTSIDE IT ARE INVALID on this workspace until a prepare of the whole frame (and so are
 * gdb_encode / GDB_PREP_SOURCES_READY, which read the whole pyramid). */
int gdb_prepare_rows(const GdbConfig* cfg, const GdbFrame* frame, const float* d_fpn_feat, int32_t flags, int32_t row_begin,
                     int32_t row_end, void* d_workspace, size_t workspace_bytes, void* stream);

/* Where the half-precision pyramid sits in the workspace (tests / callers that read it): out[0] = byte offset, out[1] = bytes per
 * (batch, view), out[2] = mip levels beyond 0, out[3 + l] = byte offset of level l inside a (batch, view) block.  A level of hw
 * texels is three planes: [0, 16 hw) 16 bytes per texel = halves of channels 0..3, 8..11; [16 hw, 32 hw) channels 4..7, 12..15;
 * [32 hw, 40 hw) 8 bytes per texel = channels 16..19 (19 is padding). */
int gdb_pyramid16_layout(const GdbConfig* cfg, const GdbFrame* shape, size_t out[7]);

/* BundleSampler.build_rays, bundle_sampler.py:30-74.  Needs gdb_prepare on the same
 * workspace first.  Outputs: d_rays_d (B,Ho,Wo,3), d_uv (Ho,Wo,2), d_rays_o (B,3),
 * d_z_axis (B,3), d_tar_pixel_radius (B). */
int gdb_build_rays(const GdbConfig* cfg, const GdbFrame* frame, const void* d_workspace, float* d_rays_d,
                   float* d_uv, float* d_rays_o, float* d_z_axis, float* d_tar_pixel_radius, void* stream);

/* BundleSampler.sample, bundle_sampler.py:193-265.  Needs gdb_prepare first.  Arrays are
 * sized for the maximum N_max = B*H*W*S_max; the first *d_total entries are valid, in the
 * reference's order (bundle-major, sample-minor).  d_rays_xyz (N_max,3,b*b), d_uvd (N_max,3),
 * d_z_vals, d_ball_radii (N_max), d_indices int64 (N_max), d_samples_per_batch int64 (B),
 * d_samples_per_bundle int32 (B*H*W), d_total int64 (1). */
int gdb_sample(const GdbConfig* cfg, const GdbFrame* frame, void* d_workspace, float* d_rays_xyz,
               float* d_uvd, float* d_z_vals, float* d_ball_radii, int64_t* d_indices,
               int64_t* d_samples_per_batch, int32_t* d_samples_per_bundle, int64_t* d_total, void* stream);

/* BundleSampler.encode, bundle_sampler.py:267-371 (including the nvdiffrast.torch.texture
 * call at :355-359 and the two F.grid_sample calls at :323,:336).  Needs gdb_prepare first.
 * n_alloc = rows allocated in the sample arrays / outputs; the valid count is read on the
 * device from d_total and d_samples_per_batch.  Outputs d_rgbs_feat_dir (V,n_alloc,3b²+C_f+3+4),
 * d_vox_feat (n_alloc,C_v). */
int gdb_encode(const GdbConfig* cfg, const GdbFrame* frame, const void* d_workspace, const float* d_rays_xyz,
               const float* d_uvd, const float* d_ball_radii, const int64_t* d_samples_per_batch,
               const int64_t* d_total, int64_t n_alloc, float* d_rgbs_feat_dir, float* d_vox_feat,
               void* stream);

/* NeRF.forward, networks/gdb_nerf/nerf.py:84-115, exact fp32.  d_rgbs_feat_dir (V,n_alloc,P)
 * with P = 3b²+C_f+3+4; rows [0,n) are processed (n from *d_total when d_total != NULL, else
 * n_alloc).  Outputs d_sigma (n_alloc), d_feat (n_alloc, P-4+C_v). */
int gdb_mlp(const GdbConfig* cfg, const float* d_packed_weights, int32_t V, const float* d_vox_feat,
            const float* d_rgbs_feat_dir, const int64_t* d_total, int64_t n_alloc, float* d_sigma,
            float* d_feat, void* stream);

/* render_weight_from_density + accumulate_value_along_rays, networks/gdb_nerf/utils.py:19-43,
 * 88-121 (nerfacc.volrend.render_weight_from_alpha / accumulate_along_rays at :35,:110), with
 * the inv_depth handling of Network.render_bundles, network.py:83-89.  d_indices sorted
 * ascending.  channels = columns of d_feat.  Outputs d_weights (n_alloc), d_bundle_feat
 * (n_bundles,channels), d_depth, d_opacity (n_bundles). */
int gdb_composite(const GdbConfig* cfg, const float* d_sigma, const float* d_feat, const float* d_z_vals,
                  const int64_t* d_indices, const int64_t* d_total, int64_t n_alloc, int64_t n_bundles,
                  int32_t channels, float* d_weights, float* d_bundle_feat, float* d_depth, float* d_opacity,
                  void* d_scratch_2xnbundles_i32, void* stream);

/* render_weight_from_density alone, networks/gdb_nerf/utils.py:19-43 (alpha = 1-exp(-sigma);
 * nerfacc.volrend.render_weight_from_alpha :35; per-bundle normalisation :38-41).
 * d_scratch: 2*n_bundles int32. */
int gdb_render_weights(const GdbConfig* cfg, const float* d_sigma, const int64_t* d_indices, const int64_t* d_total,
                       int64_t n_alloc, int64_t n_bundles, float* d_weights, void* d_scratch_2xnbundles_i32, void* stream);

/* accumulate_value_along_rays alone, networks/gdb_nerf/utils.py:88-121
 * (nerfacc.volrend.accumulate_along_rays :110): segmented sum of weights * [feat | z | 1].  z_vals are
 * used as given (the caller applies inv_depth as Network.render_bundles does). */
int gdb_accumulate(const GdbConfig* cfg, const float* d_weights, const float* d_feat, const float* d_z_vals,
                   const int64_t* d_indices, const int64_t* d_total, int64_t n_alloc, int64_t n_bundles, int32_t channels,
                   float* d_feat_map, float* d_depth_map, float* d_opacity_map, void* d_scratch_2xnbundles_i32,
                   void* stream);

/* ---- production entry ---------------------------------------------------------------- */
/* The whole hot-path section of Network.forward (network.py:145-169: build_rays → sample →
 * encode → render_bundles) in one pass with no intermediate in HBM.  Needs gdb_prepare on
 * the same workspace first.  row_begin/row_end select a strip of bundle-map rows
 * [row_begin,row_end) of every batch item (multi-GPU row-strip sharding); outputs are full
 * size and only the strip's rows are written.
 *
 * precision (arithmetic of the NeRF MLP, nerf.py:84-115; fetch, geometry and composite are fp32 either way):
 *   GDB_PREC_F16  f16 MFMA operands, fp32 accumulate (v_mfma_f32_32x32x16_f16) — narrower than the reference;
 *   GDB_PREC_F32  fp32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fmaf chain, one rounding per product) —
 *                 the reference's own precision.
 *   GDB_PREC_F32X split-f16: every MFMA operand as an f16 pair hi + lo (about 22 bits), a product as lo·hi + hi·lo + hi·hi on
 *                 v_mfma_f32_32x32x16_f16 with fp32 accumulate — fp32-grade (not bit-exact fp32) at close to the f16 rate.
 *                 Like GDB_PREC_F16 it assumes activations inside the f16 range: the high half saturates at 65504 (the low
 *                 half then carries the rest up to about twice that; beyond it the value is clipped).  Operand range of the
 *                 "22 bits": the low half of a value below 2^-3 is an f16 subnormal (resolution 2^-24), so a pair carries an
 *                 ABSOLUTE error floor of 2^-25 per operand: |x| >= 0.125 keeps ~22 bits, |x| = 1e-3 about 15, and below 6e-5
 *                 the high half is subnormal too (plain f16 accuracy).  Measured on the MLP with every weight scaled (profiles/
 *                 r03/split_f16_small_operands.txt): |f32x - f32| / output scale 5e-7 at x1, 1.5e-6 at x0.2, 5e-6 at x0.05,
 *                 3e-5 at x0.01, 2e-4 (= GDB_PREC_F16's) at x0.001.  Weights are not rescaled at pack time: use GDB_PREC_F32 for
 *                 checkpoints with layers of such small magnitude.
 * schedule (work decomposition; results agree to rounding): GDB_SCHED_AUTO picks by shape, GDB_SCHED_SLOT_WAVES =
 *   one wave per sample slot with the composite through LDS, GDB_SCHED_SEGMENT_WAVE = one wave walks all slots of
 *   its 32 bundles with the composite in registers, GDB_SCHED_DENSE = the reference's compacted sample list
 *   (bundle_sampler.py:182-189): one wave per window of WHOLE consecutive bundles of a bundle-map row holding <= 32 samples,
 *   composite across lanes; GDB_SCHED_FLAT = the same list read as one list over the rows and cut into windows of EXACTLY 32
 *   consecutive samples (4-8 % fewer waves; a bundle may straddle two windows - both waves then leave its samples' records in
 *   d_workspace and whichever of the two arrives last at the boundary's counter composites it, inside the same launch, with the in-wave
 *   composite's own arithmetic, bit for bit: the result depends neither on where the windows fall nor on the order of arrival, row strips
 *   equal the full render; bit-identical to GDB_SCHED_DENSE.  The counters live in d_workspace, are zeroed by gdb_prepare and by the
 *   plan rebuild of a render call, and are left at zero by every completed render: one render at a time per workspace).
 *   (GDB_SCHED_AUTO takes DENSE for adaptive counts, FLAT where it measured faster: fp32, S_max <= 4, frames of few tiles per
 *   wave slot.  Both need the per-row plan + sample list in d_workspace: by default the render
 *   call builds them itself from frame->d_depth_range, a small launch of its own on the same stream — the one place a render
 *   call writes the workspace.  gdb_prepare builds the same plan inside its own launch when the frame it is given carries
 *   d_depth_range and the config is adaptive; a caller that has NOT changed the contents of d_depth_range since that
 *   gdb_prepare may OR GDB_SCHED_PLAN_READY into `schedule` to skip the rebuild.  With the flag set on a stale or missing plan
 *   the result is undefined but memory-safe: the kernel clamps everything it reads from the plan to the frame.)
 * Both are per-call arguments: the library keeps no process-global state (two engines with different settings may
 * interleave calls on different streams or threads).
 * Outputs d_bundle_feat (B*H*W, 3b²+C_f+3+C_v), d_depth, d_opacity (B*H*W).
 * bundle_size 1 and 4 (network.py:31-34; configs/dtu_pretrain.yaml:33 "4 for 4*4"; since round 6): TWO launches - the dense list kernel
 * on the bundles' CENTRE rays (everything of a sample but its 3b² sub-ray colours, which the MLP never sees: nerf.py:98), which also
 * leaves per (sample, view) the weight those colours get in the bundle's output (normalised composite weight x softmax blend weight,
 * utils.py:35-41, nerf.py:108-110), then k_bundle_colours: one thread per (bundle, sub-ray) gathers the colours with the reference's own
 * per-tap arithmetic (bundle_sampler.py:327-337) and assembles the rows.  `schedule` beyond its flags is ignored there (the one
 * schedule is DENSE), GDB_PREC_F32X runs the fp32 kernel; rows strips, batches and both output layouts as for bundle_size 2. */
#define GDB_PREC_F16 0
#define GDB_PREC_F32 1
#define GDB_PREC_F32X 2
#define GDB_SCHED_AUTO 0
#define GDB_SCHED_SLOT_WAVES 1
#define GDB_SCHED_SEGMENT_WAVE 2
#define GDB_SCHED_DENSE 3
#define GDB_SCHED_FLAT 4 /* (needs W x S_max < 65536, B x H < 65536, H x W < 2^24; GDB_E_SHAPE otherwise) */
#define GDB_SCHED_PLAN_READY 0x100 /* flag: the dense plan in d_workspace was built by gdb_prepare from the current d_depth_range */
#define GDB_SCHED_PYR16_READY 0x200 /* flag (GDB_PREC_F16): the half-precision pyramid in d_workspace was built by gdb_prepare_ex(GDB_PREP_PYR16)
                                     * for this frame; without it a GDB_PREC_F16 render first converts the fp32 pyramid (a launch of its own) */
int gdb_render_bundles_fused(const GdbConfig* cfg, const GdbFrame* frame, const void* d_workspace,
                             const float* d_packed_weights, int32_t row_begin, int32_t row_end,
                             int32_t precision, int32_t schedule, float* d_bundle_feat, float* d_depth,
                             float* d_opacity, void* stream);

/* What the two fused entries would do for this (config, frame shape, precision, row strip), without launching anything (ABI v6): the
 * library's own answer, so that no caller restates its rules.  shape needs B, V, Ho, Wo, H, W, D (no pointers).
 *   out[0]  1 when gdb_render_bundles_fused / _packed accept the config and frame (>= 2 source views; bundle_size 1, 2 or 4 since round
 *           6); 0: only the operator mirrors above run it (gdb_sample -> gdb_encode -> gdb_mlp -> gdb_composite).
 *   out[1]  the schedule GDB_SCHED_AUTO resolves to for this call (GDB_SCHED_SLOT_WAVES .. GDB_SCHED_FLAT; 0 when out[0] is 0).
 *   out[2]  1 when gdb_prepare builds the list schedules' plan for this config if the frame it is given carries d_depth_range - i.e. when a
 *           render of that frame may be passed GDB_SCHED_PLAN_READY (adaptive counts: while the contents of d_depth_range are unchanged).
 *   out[3]  kernel launches the render enqueues (1; one per batch item for the flat schedule and for a dense row strip of a batch; one
 *           more - k_bundle_colours - at bundle_size 1 / 4). */
int gdb_render_info(const GdbConfig* cfg, const GdbFrame* shape, int32_t precision, int32_t row_begin, int32_t row_end, int32_t out[4]);

/* The same with ONE output buffer d_out (B*H*W, Q+2), row = [bundle_feat (Q) | depth | opacity]: what
 * Network.render_bundles returns (network.py:54-91) as a single tensor, so that a row strip is one contiguous
 * block and the multi-GPU exchange is a single all-gather. */
int gdb_render_bundles_packed(const GdbConfig* cfg, const GdbFrame* frame, const void* d_workspace,
                              const float* d_packed_weights, int32_t row_begin, int32_t row_end,
                              int32_t precision, int32_t schedule, float* d_out, void* stream);

/* ---- "next" rows: the step just before the hot path ------------------------ */
/* build_feature_volume, networks/gdb_nerf/depth_net.py:424-476: plane-sweep warp of the source feature maps
 * d_src_feat (B,V,C,Hs,Ws) onto the target frustum planes d_depth_values (B,D,Ht,Wt) (depth or, with
 * inv_depth, disparity) + biased variance over views -> d_out (B,C,D,Ht,Wt).  Intrinsics are the
 * stage-scaled ones the reference passes.  d_proj_ws: B*V*12 floats of scratch. */
int gdb_build_feature_volume(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                             const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                             int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                             int32_t inv_depth, float* d_proj_ws, float* d_out, void* stream);

/* The same with scratch for a channel-pair re-layout of the source maps (ABI v5): d_pair_ws = B*V*C*Hs*Ws floats, or NULL (then
 * exactly gdb_build_feature_volume).  With it, an even C and V <= 4 the maps are first copied to [c/2][y][x][2] and the sweep takes
 * one 16-byte load per (channel PAIR, row, view) instead of two 8-byte ones: the kernel is bound by load instructions (texture
 * addresser), 48 -> 37 us at the DTU 256x320 stage including the copy.  Bit-identical results. */
int gdb_build_feature_volume_ws(const float* d_src_feat, const float* d_src_exts, const float* d_src_ints,
                                const float* d_tar_exts, const float* d_tar_ints, const float* d_depth_values,
                                int32_t B, int32_t V, int32_t C, int32_t Hs, int32_t Ws, int32_t D, int32_t Ht, int32_t Wt,
                                int32_t inv_depth, float* d_proj_ws, float* d_pair_ws, float* d_out, void* stream);

/* depth_regression, depth_net.py:479-514: d_depth (B,1,H,W) soft-argmax of d_depth_values (B,D,H,W) under
 * d_depth_prob, d_ci (B,2,H,W) = mean -/+ ci_scale*std clipped to the hypothesis range (both returned as
 * depths when inv_depth). */
int gdb_depth_regression(const float* d_depth_values, const float* d_depth_prob, int32_t B, int32_t D, int32_t H, int32_t W,
                         float ci_scale, int32_t inv_depth, float* d_depth, float* d_ci, void* stream);

/* ---- merge around the decoder (next row N1) --------------------------------------------- */
/* network.py:170-182.  rgb_f = pixel_shuffle(bundle_feat[:, :3 b^2], b); img = rgb_c + rgb_f (rgb_c = the decoder's
 * (B,3,Ho,Wo) output, NULL = zeros); reweighting != 0: img = 0.5 (img + rgb_f).  d_out_depth / d_out_opacity
 * (B,Ho,Wo), either may be NULL: F.interpolate(scale_factor=b, mode='bilinear', align_corners=False) of the
 * (B,H,W) bundle maps.  shape needs B, H, W (Ho = H b, Wo = W b). */
int gdb_merge(const GdbConfig* cfg, const GdbFrame* shape, const float* d_bundle_feat, const float* d_rgb_c,
              const float* d_bundle_depth, const float* d_bundle_opacity, int32_t reweighting, float* d_img,
              float* d_out_depth, float* d_out_opacity, void* stream);
/* The same on the packed render of gdb_render_bundles_packed ((n_bundles, Q + 2) rows [bundle_feat | depth | opacity]), read in
 * place: the buffer a row-strip all-gather leaves on every rank (network.py:170-182 after the exchange of SURVEY.md 8(e)). */
int gdb_merge_packed(const GdbConfig* cfg, const GdbFrame* shape, const float* d_packed, const float* d_rgb_c, int32_t reweighting,
                     float* d_img, float* d_out_depth, float* d_out_opacity, void* stream);

/* ---- the decoder itself (next row N1) --------------------------------------------------- */
/* Decoder.forward, networks/gdb_nerf/decoder_rdn.py:44-81 (instantiated at network.py:51 as Decoder(C_f+3+C_v, 3, num_feats=64,
 * num_layers=nerf.dec_layers, upscale_factor=b); called at network.py:170-175): in_conv, num_layers ResidualDenseBlocks with
 * squeeze-excitation, up-conv + PixelShuffle, 1x1 out_conv — 3x3 convolutions as implicit GEMMs on fp32 MFMA, channel-last.
 * bundle_size 2 (upscale_factor 2: one up stage) and, since round 6, 4 (two up stages, decoder_rdn.py:59-62); bundle_size 1 (no up stage)
 * is refused; num_layers 1 .. 16 (two activation buffers alternate between the blocks).
 *
 * gdb_pack_decoder_weights: h_tensors in state-dict order — in_conv.weight (64,C_in,3,3), in_conv.bias, then per block
 * conv1.weight (32,64,3,3), conv2.weight (32,96,3,3), conv3.weight (64,128,3,3), se.fc.0.weight (4,64), se.fc.2.weight (64,4),
 * then up.0.weight (256,64,3,3), up.0.bias, [bundle_size 4: up.2.weight (256,64,3,3), up.2.bias,] out_conv.weight (3,64,1,1),
 * out_conv.bias: 2 + 5 num_layers + 4 (+ 2) host pointers.
 * The LAST up stage is folded with out_conv into one 64 -> 12 convolution on the host (no non-linearity sits between up-conv,
 * PixelShuffle and out_conv); at bundle_size 4 the first up stage runs as four 64 -> 64 convolutions, one per sub-pixel of its
 * PixelShuffle, into a (2H, 2W, 64) map in the workspace, on which the folded stage then runs: d_rgb_c is (B, 3, 4H, 4W). */
int gdb_decoder_packed_floats(const GdbConfig* cfg, int32_t num_layers, size_t* out_floats);
int gdb_pack_decoder_weights(const GdbConfig* cfg, int32_t num_layers, const float* const* h_tensors, float* h_out);
/* shape needs B, H, W (the bundle map). */
int gdb_decoder_workspace_bytes(const GdbConfig* cfg, const GdbFrame* shape, size_t* out_bytes);
/* d_bundle_feat: the hot path's output rows (B*H*W, ld_bundle_feat >= Q), read in place: the decoder's input is channels
 * 3b² .. Q-1 of every row (network.py:170-174: nerf_feat[:, 3b²:]).  d_rgb_c (B,3,H*b,W*b) = the reference's `rgb_c`; feed it
 * to gdb_merge.  d_workspace: gdb_decoder_workspace_bytes, caller-owned scratch.
 * precision: GDB_PREC_F32 (fp32 MFMA, the reference's arithmetic) or GDB_PREC_F32X (split-f16 operand pairs, fp32 accumulate:
 * fp32-grade, the convolutions' matrix time 5.3x shorter); the packed weights hold both forms. */
int gdb_decode(const GdbConfig* cfg, const GdbFrame* shape, const float* d_bundle_feat, int32_t ld_bundle_feat,
               const float* d_packed_decoder_weights, int32_t num_layers, int32_t precision, void* d_workspace,
               size_t workspace_bytes, float* d_rgb_c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDB_NERF_HIP_H */
