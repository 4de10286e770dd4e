// What does a vector load cost the texture-addresser / L1 path of a gfx950 CU, by width, alignment and lane stride?
// Every wave hammers a small L1-resident buffer (16 KiB per workgroup region) with one load shape; 8 waves per SIMD so that the
// path, not latency, is the limit.  Reports cycles of CU time per wave instruction (at the clock measured by s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate tools/ubench/ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F4 __attribute__((ext_vector_type(4)));
typedef F2 F2u __attribute__((aligned(4)));
typedef F4 F4u __attribute__((aligned(4)));

template <int W, bool UNALIGNED>
__global__ void __launch_bounds__(256) k(const float* buf, int stride_b, int off_b, int iters, float* sink, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    const char* base = (const char*)buf + (blockIdx.x & 255) * 16384;  // 4 MiB buffer: L2-resident, 16 KiB regions L1-resident
    unsigned o = (unsigned)(lane * stride_b + off_b);
    float s = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned a = (o + (unsigned)r * 512u) & 8191u;   // stays inside the region, keeps the lane pattern (512 | all strides used)
            if (W == 1) s += *(const float*)(base + a);
            else if (W == 2) { F2 v = UNALIGNED ? (F2)*(const F2u*)(base + a) : *(const F2*)(base + a); s += v[0] + v[1]; }
            else { F4 v = UNALIGNED ? (F4)*(const F4u*)(base + a) : *(const F4*)(base + a); s += v[0] + v[1] + v[2] + v[3]; }
        }
        o += 64u;  // walk (keeps alignment class for the strides below: 64 is a multiple of 16)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}

struct Case { const char* name; int w; bool un; int stride, off; };

int main() {
    float* buf; float* sink; unsigned long long* clk;
    hipMalloc(&buf, 4 << 20 | 65536); hipMemset(buf, 0, 4 << 20 | 65536); hipMalloc(&sink, 64); hipMalloc(&clk, 16);
    const Case cases[] = {
        {"dword    stride 4  (contiguous)", 1, false, 4, 0},
        {"dword    stride 64 (one lane per 64-B segment)", 1, false, 64, 0},
        {"dwordx2  stride 8  aligned", 2, false, 8, 0},
        {"dwordx2  stride 8  at +4 B (misaligned)", 2, true, 8, 4},
        {"dwordx2  stride 4  (overlapping x pairs: lane i reads floats i, i+1)", 2, true, 4, 0},
        {"dwordx2  stride 8  at +4 B but lanes 2 px apart: stride 8 +4", 2, true, 8, 4},
        {"dwordx4  stride 16 aligned (contiguous)", 4, false, 16, 0},
        {"dwordx4  stride 16 at +4 B (misaligned)", 4, true, 16, 4},
        {"dwordx4  stride 64 aligned", 4, false, 64, 0},
        {"dwordx4  stride 80 aligned (20-float texels)", 4, false, 80, 0},
        {"dwordx4  stride 8  (overlapping 4-float runs, lanes 2 floats apart)", 4, true, 8, 0},
    };
    const int blocks = 256 * 8, iters = 2000;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    printf("%-72s %10s %12s\n", "load shape (64 lanes)", "ns/instr/CU", "cycles/instr");
    for (const Case& c : cases) {
        float best = 1e9; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (c.w == 1) hipLaunchKernelGGL((k<1, false>), dim3(blocks), dim3(256), 0, 0, buf, c.stride, c.off, iters, sink, clk);
            else if (c.w == 2 && !c.un) hipLaunchKernelGGL((k<2, false>), dim3(blocks), dim3(256), 0, 0, buf, c.stride, c.off, iters, sink, clk);
            else if (c.w == 2) hipLaunchKernelGGL((k<2, true>), dim3(blocks), dim3(256), 0, 0, buf, c.stride, c.off, iters, sink, clk);
            else if (!c.un) hipLaunchKernelGGL((k<4, false>), dim3(blocks), dim3(256), 0, 0, buf, c.stride, c.off, iters, sink, clk);
            else hipLaunchKernelGGL((k<4, true>), dim3(blocks), dim3(256), 0, 0, buf, c.stride, c.off, iters, sink, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
        }
        const double instr_per_cu = (double)blocks / 256 * 4 * iters * 8;   // wave instructions issued on one CU
        const double ghz = h[0] ? (double)h[1] / ((double)h[0] * 10.0) : 0;  // s_memrealtime ticks at 100 MHz
        printf("%-72s %10.2f %12.1f   (clock %.2f GHz)\n", c.name, best * 1e6 / instr_per_cu, best * 1e6 / instr_per_cu * ghz, ghz);
    }
    return 0;
}
