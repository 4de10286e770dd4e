// Issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950, at 1..3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define REP8(x) x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
    float m = 0.999f, c = 0.001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 16 independent v_fma_f32 (same flops as 8 packed)
            asm volatile(REP8(
                "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "v"(m), "v"(c));
        } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7};
            f2 mm = {m, m}, cc = {c, c};
            asm volatile(REP8(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
            b0 = p4.x; b1 = p4.y; b2 = p5.x; b3 = p5.y; b4 = p6.x; b5 = p6.y; b6 = p7.x; b7 = p7.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int MODE>
double run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < 8; ++r) {  // minimum of 8: the first launches run while the clock is still ramping
        CK(hipEventRecord(e0));
        k<MODE><<<blocks, 256>>>(out, iters, 1.f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 256 * 16));
    const int iters = 4000;
    run<0>(out, 1024, iters); run<1>(out, 1024, iters);  // warm the clocks
    for (int wps = 1; wps <= 8; ++wps) {       // waves per SIMD: blocks of 4 waves, wps blocks per CU
        int blocks = 256 * wps;
        double t0 = run<0>(out, blocks, iters), t1 = run<1>(out, blocks, iters);
        double fma_pairs = (double)iters * 8 * 8;  // per wave: 64 "two-fma units" per iteration
        // cycles per instruction per SIMD at 2.4 GHz: time * clk / (instr per wave * waves per SIMD)
        printf("waves/SIMD %d: v_fma_f32 x2  %.3f ms (%.2f cyc per 2 fma)   v_pk_fma_f32 %.3f ms (%.2f cyc per pk)   ratio %.2f\n", wps,
               t0, t0 * 1e-3 * 2.4e9 / (fma_pairs * wps), t1, t1 * 1e-3 * 2.4e9 / (fma_pairs * wps), t0 / t1);
    }
    return 0;
}
