// Does v_pk_fma_f32 compute correctly when several waves share a SIMD?  Each lane runs a chain of packed FMAs whose exact
// result is known (x <- x * 1 + 1, N times, from lane-dependent start values), interleaved with plain FMAs and an MFMA
// per iteration; results are checked on the host.  Build: hipcc --offload-arch=gfx950 -O3 -o pk_verify pk_verify.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <bool MFMA>
__global__ void __launch_bounds__(256) k(float* out, const float* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i].x = in[(t * 16 + 2 * i) & 1023]; a[i].y = in[(t * 16 + 2 * i + 1) & 1023]; s[i] = a[i].x; }
    const f2 one = {1.f, 1.f};
    f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h8 ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)1.f; hb[i] = (_Float16)(1.f / 16.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], one, one);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = fmaf(s[i], 1.f, 1.f);
        if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);  // adds 1 to every element
        if ((it & 63) == 63) { float v = in[(t + it) & 1023]; a[0].x += v - v; }            // a load in the loop now and then
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += (a[i].x - s[i]) + (a[i].y - in[(t * 16 + 2 * i + 1) & 1023] - (float)iters);
    if (MFMA) for (int i = 0; i < 16; ++i) r += acc[i] - (float)iters;
    out[t] = r;  // 0 when every packed result equals its scalar twin / its closed form
}

int main() {
    const int iters = 2000;
    std::vector<float> hin(1024); for (int i = 0; i < 1024; ++i) hin[i] = (float)(i % 97);  // small integers: exact in f32
    float *din, *dout; CK(hipMalloc(&din, 4096)); CK(hipMalloc(&dout, sizeof(float) * 256 * 256 * 8));
    CK(hipMemcpy(din, hin.data(), 4096, hipMemcpyHostToDevice));
    for (int mf = 0; mf < 2; ++mf)
        for (int wps = 1; wps <= 4; ++wps) {
            const int blocks = 256 * wps, n = blocks * 256;
            long bad = 0;
            for (int rep = 0; rep < 20; ++rep) {
                if (mf) k<true><<<blocks, 256>>>(dout, din, iters); else k<false><<<blocks, 256>>>(dout, din, iters);
                CK(hipDeviceSynchronize());
                std::vector<float> h(n); CK(hipMemcpy(h.data(), dout, sizeof(float) * n, hipMemcpyDeviceToHost));
                for (int i = 0; i < n; ++i) bad += h[i] != 0.f;
            }
            printf("mfma %d, waves/SIMD %d: %ld wrong lanes in 20 launches of %d lanes\n", mf, wps, bad, n);
        }
    return 0;
}
