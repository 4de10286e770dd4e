// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs, and does v_cvt_pkrtz_f16_f32 produce them?  (The split-f16
// precision GDB_PREC_F32X stores the low parts of weights ~0.1 as f16 subnormals.)  Prints one line per case.  gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__global__ void k(float a_val, float b_val, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // one product: A[row][k=0] = a_val for every row, B[k=0][col] = b_val for every column (element 0 of half 0)
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // cvt_pkrtz of a value in the f16 subnormal range
    auto p = __builtin_amdgcn_cvt_pkrtz(a_val, b_val);
    half2v q = __builtin_bit_cast(half2v, p);
    if (threadIdx.x == 0) { out[1] = (float)q.x; out[2] = (float)q.y; }
    // residual of a split
    float v = a_val * 1000.f + b_val;
    auto hp = __builtin_amdgcn_cvt_pkrtz(v, v);
    half2v hq = __builtin_bit_cast(half2v, hp);
    float r = v - (float)hq.x;
    if (threadIdx.x == 0) { out[3] = v; out[4] = (float)hq.x; out[5] = r; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float cases[][2] = {{3.0e-5f, 1.0f}, {1.0f, 3.0e-5f}, {6.0e-8f, 1.0f}, {3.0e-5f, 3.0e-5f}, {1.0e-6f, 2048.f}, {0.1f, 0.7f}};
    for (auto& c : cases) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h[6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("a=%.9g b=%.9g  mfma=%.9g (exact f16 product %.9g)  pkrtz=(%.9g, %.9g)  split: v=%.9g hi=%.9g lo=%.9g\n", c[0], c[1], h[0],
               (double)(float)(_Float16)c[0] * (double)(float)(_Float16)c[1], h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}
