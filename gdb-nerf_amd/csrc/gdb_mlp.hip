// NeRF.forward (networks/gdb_nerf/nerf.py:58-115) in exact fp32 on the vector ALUs, one lane
// per sample with the weights broadcast from scalar registers, and the normalised alpha
// composite (networks/gdb_nerf/utils.py:19-43,88-121).  Operator mirrors; the fused kernel
// (gdb_fused.hip) is the production path.
#include "gdb_internal.h"
#include <cstring>

int gdb_fail(int code, const char* fmt, ...);
int gdb_check_cfg(const GdbConfig* c);
size_t gdb_mfma_section_floats();
void gdb_pack_mfma_section(const float* fp32_section, float* out);

#define LAUNCH_CHECK(name)                                                                    \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return gdb_fail(GDB_E_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ============================================================================================
// weight packing (host)
// ============================================================================================
extern "C" int gdb_packed_weight_floats(const GdbConfig* cfg, size_t* out_floats) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!out_floats) return gdb_fail(GDB_E_BADARG, "out_floats is NULL");
    *out_floats = (size_t)PW_FP32_FLOATS + gdb_mfma_section_floats();
    return GDB_OK;
}

extern "C" int gdb_pack_weights(const GdbConfig* cfg, const float* const t[18], float* out) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!t || !out) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    static const int offs[18] = {PW_VIEW_W, PW_VIEW_B, PW_GLOB_W, PW_GLOB_B, PW_AGG_W, PW_AGG_B, PW_FC_W, PW_FC_B, PW_LR0_W,
                                 PW_LR0_B, PW_SIG_W, PW_SIG_B, PW_W0_W, PW_W0_B, PW_W2_W, PW_W2_B, PW_FH_W, PW_FH_B};
    static const int sizes[18] = {GDB_CFR * 4, GDB_CFR, GDB_GF * 3 * GDB_CFR, GDB_GF, GDB_GF, 1, GDB_IM * GDB_GF, GDB_IM,
                                  GDB_HID * GDB_HD, GDB_HID, GDB_HID, 1, GDB_HID * GDB_W0IN, GDB_HID, GDB_HID, 1,
                                  GDB_CV * GDB_HID, GDB_CV};
    memset(out, 0, sizeof(float) * ((size_t)PW_FP32_FLOATS + gdb_mfma_section_floats()));
    for (int i = 0; i < 18; ++i) {
        // view_fc does not exist without viewdir_agg (nerf.py:19-23): its slots stay zero, which makes the
        // fused kernel's ReLU(W_view dir + b) term vanish whatever the caller handed in
        if (i < 2 && !cfg->viewdir_agg) continue;
        if (!t[i]) return gdb_fail(GDB_E_BADARG, "weight tensor %d is NULL", i);
        memcpy(out + offs[i], t[i], sizeof(float) * sizes[i]);
    }
    gdb_pack_mfma_section(out, out + PW_FP32_FLOATS);
    return GDB_OK;
}

// ============================================================================================
// A5  MLP, fp32
// ============================================================================================
__device__ __forceinline__ float relu(float x) { return x < 0.f ? 0.f : x; }  // NaN stays NaN, as torch's ReLU

// g_v = feat19 + ReLU(W_view dir + b)      nerf.py:69-71
__device__ __forceinline__ void view_feat(const float* __restrict__ pw, int viewdir, const float* __restrict__ tail,
                                          float g[GDB_CFR]) {
#pragma unroll
    for (int c = 0; c < GDB_CFR; ++c) g[c] = tail[c];
    if (viewdir) {
        float d0 = tail[GDB_CFR], d1 = tail[GDB_CFR + 1], d2 = tail[GDB_CFR + 2], d3 = tail[GDB_CFR + 3];
#pragma unroll
        for (int c = 0; c < GDB_CFR; ++c) {
            float a = pw[PW_VIEW_B + c];
            a = fmaf(pw[PW_VIEW_W + 4 * c + 0], d0, a);
            a = fmaf(pw[PW_VIEW_W + 4 * c + 1], d1, a);
            a = fmaf(pw[PW_VIEW_W + 4 * c + 2], d2, a);
            a = fmaf(pw[PW_VIEW_W + 4 * c + 3], d3, a);
            g[c] += relu(a);
        }
    }
}

__device__ __forceinline__ void softmax_views(float s[GDB_MAX_VIEWS], int V) {
    float m = s[0];
#pragma unroll
    for (int v = 1; v < GDB_MAX_VIEWS; ++v) if (v < V) m = fmaxf(m, s[v]);
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) if (v < V) { s[v] = expf(s[v] - m); sum += s[v]; }
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) if (v < V) s[v] = s[v] / sum;
}

__global__ void __launch_bounds__(128)
k_mlp(const float* __restrict__ pw, int V, int viewdir, int P, const float* __restrict__ vox, const float* __restrict__ xin,
      const int64_t* __restrict__ total, int64_t n_alloc, float* __restrict__ sigma_out, float* __restrict__ feat_out) {
    int64_t n = total ? *total : n_alloc;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int T = P - GDB_FV;  // first channel of [feat | rgb | dir]
    const float Vf = (float)V;

    // ---- view aggregation: mean / unbiased variance over views      nerf.py:73
    float mean[GDB_CFR], var[GDB_CFR];
#pragma unroll
    for (int c = 0; c < GDB_CFR; ++c) { mean[c] = 0.f; var[c] = 0.f; }
    for (int v = 0; v < V; ++v) {
        float g[GDB_CFR];
        view_feat(pw, viewdir, xin + ((size_t)v * n_alloc + i) * P + T, g);
#pragma unroll
        for (int c = 0; c < GDB_CFR; ++c) mean[c] += g[c];
    }
#pragma unroll
    for (int c = 0; c < GDB_CFR; ++c) mean[c] = mean[c] / Vf;
    for (int v = 0; v < V; ++v) {
        float g[GDB_CFR];
        view_feat(pw, viewdir, xin + ((size_t)v * n_alloc + i) * P + T, g);
#pragma unroll
        for (int c = 0; c < GDB_CFR; ++c) { float d = g[c] - mean[c]; var[c] += d * d; }
    }
#pragma unroll
    for (int c = 0; c < GDB_CFR; ++c) var[c] = var[c] / (Vf - 1.f);

    // global_fc on [g_v | var | mean]: the var/mean part is shared by all views      :77-78
    float base[GDB_GF];
#pragma unroll
    for (int j = 0; j < GDB_GF; ++j) {
        float a = pw[PW_GLOB_B + j];
#pragma unroll
        for (int c = 0; c < GDB_CFR; ++c) a = fmaf(pw[PW_GLOB_W + j * 3 * GDB_CFR + GDB_CFR + c], var[c], a);
#pragma unroll
        for (int c = 0; c < GDB_CFR; ++c) a = fmaf(pw[PW_GLOB_W + j * 3 * GDB_CFR + 2 * GDB_CFR + c], mean[c], a);
        base[j] = a;
    }
    float s[GDB_MAX_VIEWS];
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) s[v] = 0.f;
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) {
        if (v < V) {
            float g[GDB_CFR];
            view_feat(pw, viewdir, xin + ((size_t)v * n_alloc + i) * P + T, g);
            float sv = pw[PW_AGG_B];
#pragma unroll
            for (int j = 0; j < GDB_GF; ++j) {
                float a = base[j];
#pragma unroll
                for (int c = 0; c < GDB_CFR; ++c) a = fmaf(pw[PW_GLOB_W + j * 3 * GDB_CFR + c], g[c], a);
                sv = fmaf(pw[PW_AGG_W + j], relu(a), sv);
            }
            s[v] = relu(sv);  // :79
        }
    }
    softmax_views(s, V);
    float agg[GDB_GF];
#pragma unroll
    for (int j = 0; j < GDB_GF; ++j) agg[j] = 0.f;
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) {
        if (v < V) {
            float g[GDB_CFR];
            view_feat(pw, viewdir, xin + ((size_t)v * n_alloc + i) * P + T, g);
#pragma unroll
            for (int j = 0; j < GDB_GF; ++j) {
                float a = base[j];
#pragma unroll
                for (int c = 0; c < GDB_CFR; ++c) a = fmaf(pw[PW_GLOB_W + j * 3 * GDB_CFR + c], g[c], a);
                agg[j] += relu(a) * s[v];  // :80
            }
        }
    }

    // ---- geometry branch      :82,:100-102
    float h[GDB_HD];
#pragma unroll
    for (int c = 0; c < GDB_CV; ++c) h[c] = vox[(size_t)i * GDB_CV + c];
#pragma unroll
    for (int j = 0; j < GDB_IM; ++j) {
        float a = pw[PW_FC_B + j];
#pragma unroll
        for (int c = 0; c < GDB_GF; ++c) a = fmaf(pw[PW_FC_W + j * GDB_GF + c], agg[c], a);
        h[GDB_CV + j] = relu(a);
    }
    float x[GDB_HID];
    float sg = pw[PW_SIG_B];
#pragma unroll
    for (int j = 0; j < GDB_HID; ++j) {
        float a = pw[PW_LR0_B + j];
#pragma unroll
        for (int c = 0; c < GDB_HD; ++c) a = fmaf(pw[PW_LR0_W + j * GDB_HD + c], h[c], a);
        x[j] = relu(a);
        sg = fmaf(pw[PW_SIG_W + j], x[j], sg);
    }
    sigma_out[i] = softplus_t20(sg);

    // ---- colour branch: blend weights per view      :106-110
    float shared[GDB_HID];
#pragma unroll
    for (int j = 0; j < GDB_HID; ++j) {
        float a = pw[PW_W0_B + j];
#pragma unroll
        for (int c = 0; c < GDB_HID; ++c) a = fmaf(pw[PW_W0_W + j * GDB_W0IN + c], x[c], a);
#pragma unroll
        for (int c = 0; c < GDB_HD; ++c) a = fmaf(pw[PW_W0_W + j * GDB_W0IN + GDB_HID + c], h[c], a);
        shared[j] = a;
    }
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) s[v] = 0.f;
#pragma unroll
    for (int v = 0; v < GDB_MAX_VIEWS; ++v) {
        if (v < V) {
            const float* tail = xin + ((size_t)v * n_alloc + i) * P + T;
            float fv[GDB_FV];
#pragma unroll
            for (int c = 0; c < GDB_FV; ++c) fv[c] = tail[c];
            float u = pw[PW_W2_B];
#pragma unroll
            for (int j = 0; j < GDB_HID; ++j) {
                float a = shared[j];
#pragma unroll
                for (int c = 0; c < GDB_FV; ++c) a = fmaf(pw[PW_W0_W + j * GDB_W0IN + GDB_HID + GDB_HD + c], fv[c], a);
                u = fmaf(pw[PW_W2_W + j], relu(a), u);
            }
            s[v] = relu(u);
        }
    }
    softmax_views(s, V);
    const int Q = P - 4;  // blended channels [rgbs | feat | rgb]
    float* fo = feat_out + (size_t)i * (Q + GDB_CV);
    for (int c = 0; c < Q; ++c) {
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < GDB_MAX_VIEWS; ++v)
            if (v < V) a += xin[((size_t)v * n_alloc + i) * P + c] * s[v];
        fo[c] = a;
    }
#pragma unroll
    for (int j = 0; j < GDB_CV; ++j) {  // feat_head      :111-113
        float a = pw[PW_FH_B + j];
#pragma unroll
        for (int c = 0; c < GDB_HID; ++c) a = fmaf(pw[PW_FH_W + j * GDB_HID + c], x[c], a);
        fo[Q + j] = relu(a);
    }
}

extern "C" int gdb_mlp(const GdbConfig* cfg, const float* pw, int32_t V, const float* vox, const float* xin,
                       const int64_t* total, int64_t n_alloc, float* sigma, float* feat, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!pw || !vox || !xin || !sigma || !feat) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (V < 1 || V > GDB_MAX_VIEWS) return gdb_fail(GDB_E_SHAPE, "V=%d outside 1..%d", V, GDB_MAX_VIEWS);
    if (n_alloc < 1) return gdb_fail(GDB_E_SHAPE, "n_alloc must be positive");
    int P = 3 * cfg->bundle_size * cfg->bundle_size + GDB_FV;
    hipLaunchKernelGGL(k_mlp, dim3((unsigned)((n_alloc + 127) / 128)), dim3(128), 0, (hipStream_t)stream_, pw, V,
                       cfg->viewdir_agg, P, vox, xin, total, n_alloc, sigma, feat);
    LAUNCH_CHECK("k_mlp");
    return GDB_OK;
}

// ============================================================================================
// A6+A7  composite
// ============================================================================================
__global__ void k_seg_bounds(const int64_t* __restrict__ idx, const int64_t* __restrict__ total, int64_t n_alloc,
                             int64_t n_bundles, int32_t* __restrict__ seg) {
    int64_t n = total ? *total : n_alloc;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t b = idx[i];
    if (b < 0 || b >= n_bundles) return;
    if (i == 0 || idx[i - 1] != b) seg[2 * b] = (int32_t)i;
    if (i == n - 1 || idx[i + 1] != b) seg[2 * b + 1] = (int32_t)(i + 1);
}

// One lane per bundle: alpha = 1 - exp(-sigma) (utils.py:34), w_i = alpha_i * prod_{j<i}(1-alpha_j)
// (nerfacc render_weight_from_alpha, :35), normalised by max(sum, 1e-6) (:38-41).
__global__ void k_comp_weights(int64_t n_bundles, const int32_t* __restrict__ seg, const float* __restrict__ sigma,
                               float* __restrict__ weights) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bundles) return;
    int s = seg[2 * b], e = seg[2 * b + 1];
    float T = 1.f, sum = 0.f;
    for (int i = s; i < e; ++i) {
        float alpha = 1.f - expf(-sigma[i]);
        float w = alpha * T;
        T = T * (1.f - alpha);
        weights[i] = w;
        sum += w;
    }
    float den = fmaxf(sum, 1e-6f);
    for (int i = s; i < e; ++i) weights[i] = weights[i] / den;
}

// One lane per (bundle, channel), channel fastest: sum_i w_i * [feat | z | 1]      utils.py:109-119
__global__ void k_comp_accum(int64_t n_bundles, int C, int inv_depth, const int32_t* __restrict__ seg,
                             const float* __restrict__ weights, const float* __restrict__ feat, const float* __restrict__ z,
                             float* __restrict__ bf, float* __restrict__ depth, float* __restrict__ opac) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t b = t / (C + 2); int c = (int)(t % (C + 2));
    if (b >= n_bundles) return;
    int s = seg[2 * b], e = seg[2 * b + 1];
    float acc = 0.f;
    for (int i = s; i < e; ++i) {
        float val = c < C ? feat[(size_t)i * C + c] : (c == C ? (inv_depth ? 1.f / z[i] : z[i]) : 1.f);  // network.py:83-84
        acc += val * weights[i];
    }
    if (c < C) bf[b * C + c] = acc;
    else if (c == C) depth[b] = inv_depth ? 1.f / acc : acc;  // network.py:88-89
    else opac[b] = acc;
}

extern "C" int gdb_composite(const GdbConfig* cfg, const float* sigma, const float* feat, const float* z, const int64_t* idx,
                             const int64_t* total, int64_t n_alloc, int64_t n_bundles, int32_t channels, float* weights,
                             float* bf, float* depth, float* opac, void* scratch, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!sigma || !feat || !z || !idx || !weights || !bf || !depth || !opac || !scratch) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (n_alloc < 1 || n_bundles < 1 || channels < 1) return gdb_fail(GDB_E_SHAPE, "non-positive size");
    if (n_alloc >= ((int64_t)1 << 31)) return gdb_fail(GDB_E_SHAPE, "more than 2^31 samples");
    hipStream_t st = (hipStream_t)stream_;
    int32_t* seg = (int32_t*)scratch;
    hipError_t e = hipMemsetAsync(seg, 0, sizeof(int32_t) * 2 * (size_t)n_bundles, st);
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_seg_bounds, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, st, idx, total, n_alloc, n_bundles, seg);
    LAUNCH_CHECK("k_seg_bounds");
    hipLaunchKernelGGL(k_comp_weights, dim3((unsigned)((n_bundles + 255) / 256)), dim3(256), 0, st, n_bundles, seg, sigma, weights);
    LAUNCH_CHECK("k_comp_weights");
    int64_t nt = n_bundles * (channels + 2);
    hipLaunchKernelGGL(k_comp_accum, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, n_bundles, channels, cfg->inv_depth,
                       seg, weights, feat, z, bf, depth, opac);
    LAUNCH_CHECK("k_comp_accum");
    return GDB_OK;
}

static int seg_bounds(const int64_t* idx, const int64_t* total, int64_t n_alloc, int64_t n_bundles, int32_t* seg, hipStream_t st) {
    hipError_t e = hipMemsetAsync(seg, 0, sizeof(int32_t) * 2 * (size_t)n_bundles, st);
    if (e != hipSuccess) return gdb_fail(GDB_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_seg_bounds, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, st, idx, total, n_alloc, n_bundles, seg);
    LAUNCH_CHECK("k_seg_bounds");
    return GDB_OK;
}

extern "C" int gdb_render_weights(const GdbConfig* cfg, const float* sigma, const int64_t* idx, const int64_t* total,
                                  int64_t n_alloc, int64_t n_bundles, float* weights, void* scratch, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!sigma || !idx || !weights || !scratch) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (n_alloc < 1 || n_bundles < 1 || n_alloc >= ((int64_t)1 << 31)) return gdb_fail(GDB_E_SHAPE, "bad sizes");
    hipStream_t st = (hipStream_t)stream_;
    rc = seg_bounds(idx, total, n_alloc, n_bundles, (int32_t*)scratch, st); if (rc) return rc;
    hipLaunchKernelGGL(k_comp_weights, dim3((unsigned)((n_bundles + 255) / 256)), dim3(256), 0, st, n_bundles, (const int32_t*)scratch, sigma, weights);
    LAUNCH_CHECK("k_comp_weights");
    return GDB_OK;
}

extern "C" int gdb_accumulate(const GdbConfig* cfg, const float* weights, const float* feat, const float* z, const int64_t* idx,
                              const int64_t* total, int64_t n_alloc, int64_t n_bundles, int32_t channels, float* fm, float* dm,
                              float* om, void* scratch, void* stream_) {
    int rc = gdb_check_cfg(cfg); if (rc) return rc;
    if (!weights || !feat || !z || !idx || !fm || !dm || !om || !scratch) return gdb_fail(GDB_E_BADARG, "NULL pointer");
    if (n_alloc < 1 || n_bundles < 1 || channels < 1 || n_alloc >= ((int64_t)1 << 31)) return gdb_fail(GDB_E_SHAPE, "bad sizes");
    hipStream_t st = (hipStream_t)stream_;
    rc = seg_bounds(idx, total, n_alloc, n_bundles, (int32_t*)scratch, st); if (rc) return rc;
    int64_t nt = n_bundles * (channels + 2);
    hipLaunchKernelGGL(k_comp_accum, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, n_bundles, channels, 0,
                       (const int32_t*)scratch, weights, feat, z, fm, dm, om);
    LAUNCH_CHECK("k_comp_accum");
    return GDB_OK;
}
