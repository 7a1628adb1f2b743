// Standalone micro-benchmark of the plane-sweep kernel (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSWEEP_...=..] tools/k1_bench.hip -o /tmp/k1 && /tmp/k1
// Synthetic geometry as cnmnet_amd/synthetic.py: K = [[1.125W,0,W/2],[0,1.5H,H/2]], small rotation, +-0.1 m baseline.
#include "../cnmnet_amd/csrc/planesweep.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ __launch_bounds__(256) void ref_empty_kernel(float* out) {
    __shared__ float4 t[2048];
    t[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    if (out == nullptr) out[0] = t[threadIdx.x ^ 1].x;
}
__global__ __launch_bounds__(256) void ref_store_kernel(float* out, int G, int HW, int W) {   // pure output stream, same grid
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), p = blockIdx.z;
    for (int g = 0; g < G; ++g)
        *reinterpret_cast<float4*>(out + c4_offset(p, G, g, HW, y * W + x)) = make_float4(g, x, y, p);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, S = 2, H = argc > 2 ? atoi(argv[2]) : 192, W = argc > 3 ? atoi(argv[3]) : 256, D = argc > 4 ? atoi(argv[4]) : 64;
    const int P = B * S; const size_t HW = (size_t)H * W;
    std::vector<float> ref(B * 3 * HW), src(P * 3 * HW), hmkt(P * 12);
    srand(1);
    for (auto& v : ref) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto& v : src) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    const double fx = 1.125 * W, fy = 1.5 * H, cx = W / 2.0, cy = H / 2.0;
    for (int p = 0; p < P; ++p) {
        const double ry = ((p * 37) % 7 - 3) * 0.01, rz = ((p * 11) % 5 - 2) * 0.008, tx = (p & 1) ? -0.1 : 0.1;
        const double R[9] = {cos(ry) * cos(rz), -sin(rz), sin(ry), sin(rz), cos(rz), 0, -sin(ry), 0, cos(ry)};
        const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
        double RKi[9], Hm[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * Ki[k * 3 + j]; RKi[i * 3 + j] = s; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * RKi[k * 3 + j]; Hm[i * 3 + j] = s; }
        for (int i = 0; i < 9; ++i) hmkt[p * 12 + i] = (float)Hm[i];
        const double T[3] = {tx, 0.01, -0.01};
        for (int i = 0; i < 3; ++i) hmkt[p * 12 + 9 + i] = (float)(K[i * 3] * T[0] + K[i * 3 + 1] * T[1] + K[i * 3 + 2] * T[2]);
    }
    float *dref, *dsrc, *dh, *dout, *dws; const size_t wsn = cnm_planesweep_workspace_floats(B, S, H, W); hipMalloc(&dws, wsn * 4);
    const size_t outn = (size_t)P * (D / 4 + 1) * HW * 4;
    hipMalloc(&dref, ref.size() * 4); hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dh, hmkt.size() * 4); hipMalloc(&dout, outn * 4);
    hipMemcpy(dref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, hmkt.data(), hmkt.size() * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 5; ++i) cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, dws, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 50;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, dws, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    std::vector<float> out(outn); hipMemcpy(out.data(), dout, outn * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < outn; i += 7) cs += out[i];
    {
        dim3 grid(W / 64, H / 4, P);   // reference kernels use 64x4 tiles of 256 threads
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) ref_empty_kernel<<<grid, 256>>>(dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); printf("   (empty kernel, same grid, 32 KB LDS: %.1f us)\n", t / iters * 1e3);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) ref_store_kernel<<<grid, 256>>>(dout, D / 4 + 1, H * W, W);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t, e0, e1); printf("   (pure c4 store stream, same grid: %.1f us = %.0f GB/s)\n", t / iters * 1e3, (double)P * (D + 4) * HW * 4 / (t / iters) / 1e6);
    }
    {   // texture pre-pass alone
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) sweep_texture_kernel<<<8192, 256>>>(dsrc, reinterpret_cast<float4*>(dws), P, H, W);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); printf("   (texture pre-pass alone: %.1f us)\n", t / iters * 1e3);
    }
#ifdef SWEEP_TRACE
    {
        static long long tr[4096][40]; hipMemcpyFromSymbol(tr, HIP_SYMBOL(sweep_trace), sizeof(tr));
        const int ng = (D + SWEEP_PG - 1) / SWEEP_PG; const int nb = 3072 < 4096 ? 3072 : 4096;
        double pro = 0, box0 = 0, wait = 0, issue = 0, comp = 0, tot = 0, tail = 0;
        for (int b = 0; b < nb; ++b) {
            pro += tr[b][1] - tr[b][0]; box0 += tr[b][2] - tr[b][1]; tot += tr[b][39] - tr[b][0];
            for (int g = 0; g < ng; ++g) { wait += tr[b][4 + 4 * g] - tr[b][3 + 4 * g]; issue += tr[b][5 + 4 * g] - tr[b][4 + 4 * g]; comp += tr[b][6 + 4 * g] - tr[b][5 + 4 * g]; }
            tail += tr[b][39] - tr[b][6 + 4 * (ng - 1)];
        }
        printf("   trace (s_memtime ticks, avg per workgroup of wave 0): total %.0f | prologue %.0f box0+stage0 %.0f | per group: barrier-wait %.0f  stores+dma-issue %.0f  compute %.0f | tail %.0f\n",
               tot / nb, pro / nb, box0 / nb, wait / nb / ng, issue / nb / ng, comp / nb / ng, tail / nb);
        long long t0 = tr[0][0], t1 = tr[0][39]; for (int b = 0; b < nb; ++b) { if (tr[b][0] < t0) t0 = tr[b][0]; if (tr[b][39] > t1) t1 = tr[b][39]; }
        printf("   kernel span %lld ticks\n", t1 - t0);
    }
#endif
#ifdef SWEEP_STATS
    { unsigned int st[2]; hipMemcpyFromSymbol(st, HIP_SYMBOL(sweep_stats), 8); printf("   groups staged %u, fallback %u (%.2f%%)\n", st[0], st[1], 100.0 * st[1] / (st[0] + st[1] + 1e-9)); }
#endif
    const double bytes = (double)B * 3 * HW * 4 + (double)P * 3 * HW * 4 + (double)P * (D + 3) * HW * 4;
    printf("%-40s %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)  checksum %.6e\n", argc > 5 ? argv[5] : "k1", ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, cs);
    return 0;
}
