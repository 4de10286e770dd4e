// N1: the merge around the decoder, reference networks/gdb_nerf/network.py:170-182.
//   rgb_f = pixel_shuffle(bundle_feat[:, :3 b^2], b);  img = rgb_c + rgb_f;  reweighting: img = 0.5 (img + rgb_f)
//   nerf_depth / nerf_opacity = bilinear x b upsampling (align_corners False) of the (B, H, W) bundle maps
// HBM-bound: per bundle reads 4 (3 b^2 [+ 3 b^2 rgb_c] + ~2) and writes 4 (3 b^2 + 2 b^2) bytes.  Exact fp32, operation order of the reference's torch ops (-ffp-contract=off).
#include "gdb_internal.h"

int gdb_fail(int code, const char* fmt, ...);
int gdb_check_cfg(const GdbConfig* c);
int gdb_check_frame(const GdbConfig* c, const GdbFrame* f, bool need_ptrs);

struct MergeArgs {
    int B, H, W, b, Q, rew;
    int ld, ms;  // floats between consecutive bundle rows of bf; between consecutive entries of the depth / opacity maps
    const float* bf; const float* rgb_c; const float* dep; const float* opa;
    float* img; float* odep; float* oopa;
};

// F.interpolate(..., mode='bilinear', align_corners=False), one axis: src = (dst + 0.5) / b - 0.5 clamped at 0
__device__ __forceinline__ void up_taps(int dst, int n_in, float inv_b, int& i0, int& i1, float& l0, float& l1) {
    float src = fmaxf(((float)dst + 0.5f) * inv_b - 0.5f, 0.f);
    i0 = (int)floorf(src);
    i1 = min(i0 + 1, n_in - 1);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// Vectors of BS floats with 4-byte alignment (bundle_feat rows are 39 floats apart): one load / store per pixel row of a block.
template <int BS> struct __attribute__((packed, aligned(4))) RowV { float v[BS]; };

// One thread per (bundle, colour channel), channel fastest: the three threads of a bundle read its 3 b^2 colours as one
// contiguous run, and each writes its b x b block of one colour plane (a wave covers ~21 neighbouring blocks of a row).
// Threads of channel 0 / 1 also upsample the depth / opacity map for their bundle's block.
template <int BS>
__global__ void __launch_bounds__(256) k_merge(MergeArgs a) {
    const int n = a.B * a.H * a.W;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    const int ch = t % 3, bu = t / 3;
    const int x = bu % a.W, y = (bu / a.W) % a.H, bi = bu / (a.W * a.H);
    const int Ho = a.H * BS, Wo = a.W * BS;
    const float* row = a.bf + (size_t)bu * a.ld + ch * BS * BS;  // channel ch*b^2 + dy*b + dx   (pixel_shuffle)
#pragma unroll
    for (int dy = 0; dy < BS; ++dy) {
        const RowV<BS> f = *(const RowV<BS>*)(row + dy * BS);
        const size_t o = (((size_t)bi * 3 + ch) * Ho + (size_t)y * BS + dy) * Wo + (size_t)x * BS;
        RowV<BS> c, v;
        if (a.rgb_c) c = *(const RowV<BS>*)(a.rgb_c + o);
#pragma unroll
        for (int dx = 0; dx < BS; ++dx) {
            float w = (a.rgb_c ? c.v[dx] : 0.f) + f.v[dx];
            if (a.rew) w = 0.5f * (w + f.v[dx]);
            v.v[dx] = w;
        }
        *(RowV<BS>*)(a.img + o) = v;
    }
    const float* src = ch == 0 ? a.dep : (ch == 1 ? a.opa : nullptr);
    float* dst = ch == 0 ? a.odep : (ch == 1 ? a.oopa : nullptr);
    if (!dst) return;
    const float inv_b = 1.f / (float)BS;
    const size_t base = (size_t)bi * a.H * a.W;
#pragma unroll
    for (int dy = 0; dy < BS; ++dy) {
        int y0, y1; float ly0, ly1;
        up_taps(y * BS + dy, a.H, inv_b, y0, y1, ly0, ly1);
        RowV<BS> v;
#pragma unroll
        for (int dx = 0; dx < BS; ++dx) {
            int x0, x1; float lx0, lx1;
            up_taps(x * BS + dx, a.W, inv_b, x0, x1, lx0, lx1);
            const size_t i00 = base + (size_t)y0 * a.W + x0, i01 = base + (size_t)y0 * a.W + x1;
            const size_t i10 = base + (size_t)y1 * a.W + x0, i11 = base + (size_t)y1 * a.W + x1;
            v.v[dx] = ly0 * (lx0 * src[i00 * a.ms] + lx1 * src[i01 * a.ms]) + ly1 * (lx0 * src[i10 * a.ms] + lx1 * src[i11 * a.ms]);
        }
        *(RowV<BS>*)(dst + ((size_t)bi * Ho + (size_t)y * BS + dy) * Wo + (size_t)x * BS) = v;
    }
}

static int merge_entry(const GdbConfig* cfg, const GdbFrame* shape, const float* bf, int ld, const float* rgb_c, const float* dep, const float* opa,
                       int ms, int32_t reweighting, float* img, float* out_dep, float* out_opa, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    rc = gdb_check_frame(cfg, shape, false); if (rc) return rc;
    if (!bf || !img) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if ((out_dep && !dep) || (out_opa && !opa)) return gdb_fail(GDB_E_BADARG, "an upsampled map is requested without its bundle map");
    MergeArgs a;
    a.B = shape->B; a.H = shape->H; a.W = shape->W; a.b = cfg->bundle_size; a.rew = reweighting != 0;
    a.Q = 3 * a.b * a.b + cfg->feat_dim + 3 + cfg->voxel_dim;
    a.ld = ld > 0 ? ld : a.Q; a.ms = ms;
    a.bf = bf; a.rgb_c = rgb_c; a.dep = dep; a.opa = opa; a.img = img; a.odep = out_dep; a.oopa = out_opa;
    const int n = a.B * a.H * a.W;
    if (n == 0) return GDB_OK;
    hipStream_t st = (hipStream_t)stream_;
    const dim3 grid((3 * n + 255) / 256), block(256);
    if (a.b == 1) hipLaunchKernelGGL(k_merge<1>, grid, block, 0, st, a);
    else if (a.b == 2) hipLaunchKernelGGL(k_merge<2>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_merge<4>, grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "launch k_merge: %s", hipGetErrorString(e));
    return GDB_OK;
}

extern "C" int gdb_merge(const GdbConfig* cfg, const GdbFrame* shape, const float* bf, const float* rgb_c, const float* dep, const float* opa,
                         int32_t reweighting, float* img, float* out_dep, float* out_opa, void* stream_) {
    return merge_entry(cfg, shape, bf, 0, rgb_c, dep, opa, 1, reweighting, img, out_dep, out_opa, stream_);
}

// The same on the PACKED render (gdb_render_bundles_packed: rows [feat Q | depth | opacity]) read in place: what a row-strip
// all-gather leaves on every rank, and what Network.forward renders into.
extern "C" int gdb_merge_packed(const GdbConfig* cfg, const GdbFrame* shape, const float* packed, const float* rgb_c, int32_t reweighting,
                                float* img, float* out_dep, float* out_opa, void* stream_) {
    if (!cfg || !packed) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    const int Q = 3 * cfg->bundle_size * cfg->bundle_size + cfg->feat_dim + 3 + cfg->voxel_dim;
    return merge_entry(cfg, shape, packed, Q + 2, rgb_c, packed + Q, packed + Q + 1, Q + 2, reweighting, img, out_dep, out_opa, stream_);
}
