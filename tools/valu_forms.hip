// Probe (GPU box): issue cost of the VALU forms the plane sweep uses, 4 waves per SIMD, 16 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_forms.hip -o /tmp/vf && /tmp/vf
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, int iters, float sa, float sb) {
    float v[16], w[16], u[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 1e-3f + i; w[i] = v[i] * 0.37f + 0.1f; u[i] = 0.25f * i + threadIdx.x * 1e-4f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define OP(i) \
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(u[i])); \
            if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(u[i])); \
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(sa), "v"(u[i])); \
            if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])); \
            if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])); \
            if (KIND == 5) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(v[i]) : "v"(w[i])); \
            if (KIND == 6) asm volatile("v_fract_f32 %0, %1" : "=v"(v[i]) : "v"(w[i])); \
            if (KIND == 7) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(v[i]) : "v"(w[i])); \
            if (KIND == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "s"(sa), "v"(u[i])); \
            if (KIND == 9) asm volatile("v_mul_u32_u24 %0, %0, 48" : "+v"(v[i])); \
            if (KIND == 10) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[i]) : "v"(w[i])); \
            if (KIND == 11) asm volatile("v_add_f32 %0, |%0|, |%1|" : "+v"(v[i]) : "v"(w[i])); \
            if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[i & ~1]) : "v"(*(double*)&w[i & ~1]), "v"(*(double*)&u[i & ~1])); \
            if (KIND == 13) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(v[i]) : "v"(u[i])); \
            if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(w[i])); \
            if (KIND == 15) asm volatile("v_floor_f32 %0, %1" : "=v"(v[i]) : "v"(w[i]));
            REP16(OP)
#undef OP
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += v[i] + w[i] + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 10, 1.5f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, iters, 1.5f, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 64.0 * iters * 4 * blocks;             // wave-instructions
    printf("%-34s %d waves/SIMD: %.2f ns-cycles(2.4GHz) per wave-instruction per SIMD\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
    hipFree(out);
}
int main() {
    for (int w : {4, 1}) {
        run<0>("v_fma_f32 v,v,v (3 VGPR)", w); run<1>("v_fmac_f32 (3 VGPR)", w); run<2>("v_fma_f32 v,s,v", w); run<13>("v_fma_f32 v,1.0,v", w);
        run<3>("v_add_f32", w); run<4>("v_mul_f32", w); run<11>("v_add_f32 |a|,|b| (VOP3)", w); run<5>("v_cvt_u32_f32", w); run<6>("v_fract_f32", w); run<15>("v_floor_f32", w);
        run<7>("v_med3_f32", w); run<8>("v_mad_u32_u24 v,s,v", w); run<9>("v_mul_u32_u24", w); run<10>("v_rcp_f32", w); run<12>("v_pk_fma_f32 (3 VGPR pairs)", w); run<14>("v_mov_b32", w);
    }
    return 0;
}
