// Probe: how many independent VALU / LDS / VMEM instructions can one wave issue per v_mfma_f32_32x32x2_f32 (64-cycle
// matrix-pipe occupancy) before the matrix pipe starts to idle?  One wave per SIMD, like the Winograd kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NV, int KIND, int THREADS = 256>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: ds_read_b128, 3: global_load_dwordx4 (L2-resident)
__global__ __launch_bounds__(THREADS, 1) void k(float* out, const float4* src, int iters) {
    __shared__ float4 lds[1024];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    lds[threadIdx.x & 1023] = make_float4(1, 2, 3, 4); __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float v[8]; v2f p[8]; float4 l[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; p[i] = v2f{a, b}; l[i] = make_float4(0, 0, 0, 0); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int j = (m * NV + n) & 7;
                if (KIND == 0) v[j] = fmaf(v[j], 1.0001f, 0.5f);
                if (KIND == 1) p[j] = __builtin_elementwise_fma(p[j], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f});
                if (KIND == 2) l[j] = lds[(threadIdx.x + 16 * n + it) & 1023];
                if (KIND == 3) l[j] = src[(size_t)blockIdx.x * 4096 + ((threadIdx.x + 64 * (m * NV + n) + it * 7) & 4095)];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + l[i].x + l[i].w;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int NV, int KIND, int THREADS = 256> void run(const char* name, const float4* src) {
    float* out; hipMalloc(&out, 256 * THREADS * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, KIND, THREADS><<<256, THREADS>>>(out, src, 50); hipDeviceSynchronize();
    hipEventRecord(e0); k<NV, KIND, THREADS><<<256, THREADS>>>(out, src, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 2 * 16.0 * iters * (THREADS / 64) * 256;
    printf("%-14s %2d per MFMA, %d waves/SIMD: %.1f TF\n", name, NV, THREADS / 256, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    float4* src; hipMalloc(&src, 256 * 4096 * 16); hipMemset(src, 0, 256 * 4096 * 16);
    run<0, 0>("baseline", src);
    run<2, 0>("v_fma_f32", src); run<4, 0>("v_fma_f32", src); run<8, 0>("v_fma_f32", src); run<12, 0>("v_fma_f32", src); run<16, 0>("v_fma_f32", src);
    run<2, 1>("v_pk_fma_f32", src); run<4, 1>("v_pk_fma_f32", src); run<8, 1>("v_pk_fma_f32", src);
    run<1, 2>("ds_read_b128", src); run<2, 2>("ds_read_b128", src); run<4, 2>("ds_read_b128", src);
    run<0, 0, 512>("baseline", src); run<2, 0, 512>("v_fma_f32", src); run<4, 0, 512>("v_fma_f32", src); run<8, 0, 512>("v_fma_f32", src); run<4, 1, 512>("v_pk_fma_f32", src);
    run<1, 3>("global_load_x4", src); run<2, 3>("global_load_x4", src);
    return 0;
}
