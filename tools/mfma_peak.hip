// Ceiling probe: pure v_mfma_f32_32x32x2_f32 issue rate with W waves per SIMD (no memory traffic), plus core clock
// from s_memtime / s_memrealtime.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC> void run(int blocks, const char* name) {
    float* out; unsigned long long* clk; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 100, clk); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(out, iters, clk); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = 2.0 * 32 * 32 * 2 * (double)NACC * iters * 4 * blocks;
    printf("%-28s blocks %4d  %.3f ms  %.1f TF   shader clock %.0f MHz (memtime/realtime@100MHz)\n", name, blocks, ms, fl / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0));
}
int main() {
    run<1>(256, "1 wave/SIMD, 1 acc (chain)");
    run<2>(256, "1 wave/SIMD, 2 acc");
    run<4>(256, "1 wave/SIMD, 4 acc");
    run<16>(256, "1 wave/SIMD, 16 acc");
    run<4>(512, "2 waves/SIMD, 4 acc");
    run<4>(1024, "4 waves/SIMD, 4 acc");
    return 0;
}
