// Probe: issue rate of v_fma_f32 vs v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (16 independent chains per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[16]; v2f p[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{v[i], v[i] + 1.f}; }
    const v2f c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], c1, c2);
                if (KIND == 2) p[i] = p[i] + c2;
                if (KIND == 3) p[i] = p[i] * c1;
            }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 64.0 * iters * 4 * blocks;             // wave-instructions
    printf("%-14s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_pk_mul_f32", w); }
    return 0;
}
