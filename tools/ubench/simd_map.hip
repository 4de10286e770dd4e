// Where does the dispatcher put the waves of small workgroups?  Launch WGs of `waves` waves with `lds` bytes of
// LDS each, keep them resident for a while, record HW_ID per wave, and histogram waves per (CU, SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o simd_map simd_map.hip ; run: ./simd_map <waves per WG> <LDS bytes> <WGs>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <array>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    float acc = 0.f;
    while (wall_clock64() - t0 < spin) acc += lds[threadIdx.x & 15];  // 100 MHz counter: spin/100 microseconds
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
    if (acc == 123.456f) out[0] = 0;
}

int main(int argc, char** argv) {
    int waves = argc > 1 ? atoi(argv[1]) : 3, lds = argc > 2 ? atoi(argv[2]) : 40320, wgs = argc > 3 ? atoi(argv[3]) : 1024;
    unsigned* d; CK(hipMalloc(&d, sizeof(unsigned) * 2 * wgs * waves));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<wgs, waves * 64, lds>>>(d, 20000); CK(hipDeviceSynchronize());  // 200 us: every WG that fits is resident at once
    CK(hipEventRecord(e0));
    k<<<wgs, waves * 64, lds>>>(d, 20000);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel time %.0f us for a 200 us spin: %.1f rounds -> at most %d workgroups resident per CU\n", ms * 1e3, ms * 1e3 / 200.0,
           (int)((wgs + 255) / 256 / (int)(ms * 1e3 / 200.0 + 0.5)));
    std::vector<unsigned> h(2 * wgs * waves);
    CK(hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
    std::map<unsigned, std::array<int, 4>> per_cu;
    for (int w = 0; w < wgs * waves; ++w) {
        unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu][simd]++;
    }
    std::map<std::array<int, 4>, int> patterns;
    for (auto& kv : per_cu) patterns[kv.second]++;
    printf("%d WGs of %d waves, %d B LDS: %zu CUs seen; waves per SIMD [s0 s1 s2 s3] -> number of CUs\n", wgs, waves, lds, per_cu.size());
    for (auto& kv : patterns) printf("  [%d %d %d %d] x %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    return 0;
}
