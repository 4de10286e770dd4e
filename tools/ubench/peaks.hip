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
