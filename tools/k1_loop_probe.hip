// Probe (GPU box): the plane sweep's sample loop in isolation - no staging, no footprints, no output stream.  What does a
// SIMD sustain per sample, at 4 / 8 waves per SIMD, and which part of the loop is it waiting for?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DPROBE_MODE=n] tools/k1_loop_probe.hip -o /tmp/k1lp && /tmp/k1lp
// PROBE_MODE 0: the loop as shipped (coordinates, six ds_read_b64, blend)   1: no LDS reads (texels = registers)
//            2: LDS reads + blend, coordinates computed once (same address)   3: coordinates + reads, no blend (sum of two words)
//            4: coordinates only
#include "../cnmnet_amd/csrc/planesweep.hip"
#include <cstdio>
#include <cstdlib>
#ifndef PROBE_MODE
#define PROBE_MODE 0
#endif

__global__ __launch_bounds__(SWEEP_NT) __attribute__((amdgpu_waves_per_eu(SWEEP_MINW, SWEEP_MINW))) void probe_kernel(float* out, int iters, float step) {
    char* const box = SWEEP_LDS_BOX;
    float* const zsh = SWEEP_LDS_Z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SWEEP_CAP * 6; i += SWEEP_NT) reinterpret_cast<float*>(box)[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    if (tid < CNM_MAX_PLANES) zsh[tid] = 1.0f / (0.1f + tid * step);
    __syncthreads();
    // a 100 x 19 box, lanes on consecutive texels, 1.3 texels of parallax per plane: u' = ug + pa r, r = 1 / (a2 z + k2)
    float a2 = 1.0f + 1e-4f * lane, k2v = 0.01f, pa = 28.0f, pb = 0.3f, ug = 2.0f + lane * 0.97f, vg = 1.0f + wave * 0.9f;
    float umax = 98.f, vmax = 18.f; unsigned rwv = 100u;
    float nr = -0.1f, ng = 0.2f, nb = 0.05f;
    asm volatile("" : "+v"(a2), "+v"(k2v), "+v"(pa), "+v"(pb), "+v"(ug), "+v"(vg), "+v"(umax), "+v"(vmax), "+v"(rwv));
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int d0 = 0; d0 < 64; d0 += 4) {
            float cost[4];
#if PROBE_MODE == 0
            sweep_quad<false>(box, zsh + d0, cost, ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, nr, ng, nb);
#else
            const float4 zq = *reinterpret_cast<const float4*>(zsh + d0);
            const float zz[4] = {zq.x, zq.y, zq.z, zq.w};
            SweepCoord c0 = sweep_coords_parallax<false>(ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, zz[0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                SweepCoord c = c0;
                if (PROBE_MODE != 2) c = sweep_coords_parallax<false>(ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, zz[j]);
                if (PROBE_MODE == 4) { cost[j] = c.wu + c.wv + __uint_as_float(c.off); continue; }
                SweepTexels t;
                if (PROBE_MODE == 1) { t.lp = sw_f32x2{c.wu, c.wv}; t.ld = sw_f32x2{c.wv, nr}; t.lb = sw_f32x2{ng, c.wu}; t.rp = t.ld; t.rd = t.lb; t.rb = t.lp; asm volatile("" : "+v"(c.off)); }
                else t = sweep_texels_lds(box, c.off);
                if (PROBE_MODE == 3) cost[j] = (t.lp.x + t.ld.y) + (t.lb.x + t.rp.y) + (t.rd.x + t.rb.y) + c.wu * c.wv;
                else cost[j] = sweep_blend(t, c.wu, c.wv, nr, ng, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            acc += (cost[0] + cost[1]) + (cost[2] + cost[3]);
        }
        vg += 1e-3f;
    }
    out[blockIdx.x * SWEEP_NT + tid] = acc;
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * SWEEP_WG_PER_CU;
    float* out; hipMalloc(&out, (size_t)blocks * SWEEP_NT * 4);
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe_kernel, SWEEP_NT, 0);
    const int iters = 40;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_kernel<<<blocks, SWEEP_NT>>>(out, 2, 0.046f); hipDeviceSynchronize();
    hipEventRecord(e0); probe_kernel<<<blocks, SWEEP_NT>>>(out, iters, 0.046f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_samples_per_simd = (double)iters * 64 * (SWEEP_NT / 64) * SWEEP_WG_PER_CU / 4.0;
    printf("mode %d, %d workgroups of %d waves per CU (occupancy API %d): %.1f ns per wave-sample per SIMD = %.0f cycles at 2.4 GHz  (a 64 x 16 tile of 64 planes: %.1f us per CU)\n",
           PROBE_MODE, SWEEP_WG_PER_CU, SWEEP_NT / 64, nb, ms * 1e6 / wave_samples_per_simd, ms * 1e6 / wave_samples_per_simd * 2.4, ms * 1e3 / wave_samples_per_simd * 256);
    return 0;
}
