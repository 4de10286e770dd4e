// Attainable-peak micro-benchmarks that bench.py runs on the node beside the hot path ("each bench run
// must also print measured attainable HBM bandwidth and MFMA throughput"): an HBM stream triad and bare MFMA loops.
// Measurement infrastructure, not product: built as libgdbpeaks.so by gdb-nerf_amd/build.py, loaded by bench.py only.
// gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// a[i] = b[i] + s * c[i] on float4: 2 reads + 1 write of 16 B per element; grid-stride over n4 elements
__global__ void __launch_bounds__(256) k_triad(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                               float s, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

// Four independent accumulators per wave, operands in registers: the issue-rate ceiling of the matrix pipe.
__global__ void __launch_bounds__(256) k_mfma_f32(float* sink, int iters, float seed) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = seed + (float)(threadIdx.x & 63) * 1e-3f, b = seed * 0.5f + (float)(threadIdx.x & 31) * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f) sink[blockIdx.x] = s;  // keeps the loop alive, (practically) never stores
}

__global__ void __launch_bounds__(256) k_mfma_f16(float* sink, int iters, float seed) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + (float)((threadIdx.x + i) & 63) * 1e-3f); b[i] = (_Float16)(seed * 0.5f + (float)((threadIdx.x + 3 * i) & 31) * 2e-3f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f) sink[blockIdx.x] = s;
}

extern "C" {
// Enqueue one triad pass over n4 float4 elements per array; bytes moved = 48 * n4.
int gdb_peak_triad(void* a, const void* b, const void* c, size_t n4, void* stream) {
    hipLaunchKernelGGL(k_triad, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (float4*)a, (const float4*)b, (const float4*)c, 1.5f, n4);
    return (int)hipGetLastError();
}
// Enqueue `blocks` workgroups of 4 waves, each wave issuing 4 * iters MFMAs.  FLOP = blocks * 4 * 4 * iters * flop_per_mfma
// (v_mfma_f32_32x32x2_f32: 4096; v_mfma_f32_32x32x16_f16: 32768).
int gdb_peak_mfma(int f16, float* sink, int blocks, int iters, void* stream) {
    if (f16) hipLaunchKernelGGL(k_mfma_f16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, 0.25f);
    else hipLaunchKernelGGL(k_mfma_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, 0.25f);
    return (int)hipGetLastError();
}
}

// Do MFMAs and fp32 VALU FMAs overlap on a SIMD?  One workgroup per CU (launch 256 of them): with 256 threads each SIMD hosts one
// wave, with 512 threads two (waves 0..3 and 4..7 of a workgroup each cover the four SIMDs).  mode 0: every wave issues MFMAs;
// 1: every wave issues VALU FMAs; 2 (512 threads): waves 0..3 issue MFMAs, waves 4..7 VALU FMAs, so each SIMD hosts one of each.
// f16 != 0 uses v_mfma_f32_32x32x16_f16.
__global__ void __launch_bounds__(512) k_mix(float* sink, int iters, int mode, int f16) {
    // modes: 0 all waves MFMA; 1 all waves scalar v_fma_f32; 2 waves 0..3 MFMA + waves 4..7 scalar FMA; 3 all waves v_pk_fma_f32;
    // 4 waves 0..3 MFMA + waves 4..7 packed FMA.  (This file is built with -fno-slp-vectorize: the scalar loop stays scalar.)
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 4) && threadIdx.x < 256);
    const bool packed = mode >= 3;
    float s = 0.f;
    if (do_mfma) {
        f32x16 acc[2];
        for (int t = 0; t < 2; ++t)
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        if (f16) {
            half8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.25f + (float)((threadIdx.x + i) & 63) * 1e-3f); b[i] = (_Float16)(0.125f + (float)((threadIdx.x + 3 * i) & 31) * 2e-3f); }
            for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            }
        } else {
            float a = 0.25f + (float)(threadIdx.x & 63) * 1e-3f, b = 0.125f + (float)(threadIdx.x & 31) * 2e-3f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        for (int t = 0; t < 2; ++t)
            for (int i = 0; i < 16; ++i) s += acc[t][i];
    } else if (!packed) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
        const float m = 0.999f, c = 1e-4f;
        for (int it = 0; it < iters; ++it) {   // 8 independent chains x 4 = 32 v_fma_f32 per iteration
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], m, c);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    } else {
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v x[4];
        for (int i = 0; i < 4; ++i) x[i] = f2v{(float)(threadIdx.x + i) * 1e-3f, (float)(threadIdx.x + i + 4) * 1e-3f};
        const f2v m = {0.999f, 0.999f}, c = {1e-4f, 1e-4f};
        for (int it = 0; it < iters; ++it) {   // the same 32 FMAs as 16 v_pk_fma_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
        }
        for (int i = 0; i < 4; ++i) s += x[i].x + x[i].y;
    }
    if (s == 12345.678f) sink[blockIdx.x] = s;
}
extern "C" int gdb_peak_mix(float* sink, int blocks, int threads, int iters, int mode, int f16, void* stream) {
    hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, sink, iters, mode, f16);
    return (int)hipGetLastError();
}
