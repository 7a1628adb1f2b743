// Standalone check + micro-benchmark of the plane-sweep kernel (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DSWEEP_...=..] tools/k1_bench.hip -o /tmp/k1 && /tmp/k1
// Synthetic geometry as cnmnet_amd/synthetic.py: K = [[1.125W,0,W/2],[0,1.5H,H/2]], small rotation, +-0.1 m baseline.
// Every run compares the c4 output of pair 0 .. NCHECK-1 with a float64 closed-form evaluation on the host
// (explicit per-corner zero padding), and the NCHW layout with the c4 layout bit for bit.
#include "../cnmnet_amd/csrc/planesweep.hip"
#ifdef K1_NO_QUEUE
#define K1_WS nullptr   // fixed tile stride instead of the ticket queue
#else
#define K1_WS dws
#endif
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void ref_store_kernel(float* out, int G, int HW, int W) {   // pure output stream, same grid
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), p = blockIdx.z;
    for (int g = 0; g < G; ++g)
        *reinterpret_cast<float4*>(out + c4_offset(p, G, g, HW, y * W + x)) = make_float4(g, x, y, p);
}

static double host_sample(const float* img, int H, int W, double ix, double iy) {   // grid_sample bilinear, zeros padding
    if (!(std::fabs(ix) < 1e7) || !(std::fabs(iy) < 1e7)) return 0.0;
    const double fx = std::floor(ix), fy = std::floor(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const double wx = ix - fx, wy = iy - fy;
    auto at = [&](int yy, int xx) -> double { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? img[(size_t)yy * W + xx] : 0.0; };
    return (1 - wx) * (1 - wy) * at(y0, x0) + wx * (1 - wy) * at(y0, x0 + 1) + (1 - wx) * wy * at(y0 + 1, x0) + wx * wy * at(y0 + 1, x0 + 1);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, S = 2, H = argc > 2 ? atoi(argv[2]) : 192, W = argc > 3 ? atoi(argv[3]) : 256, D = argc > 4 ? atoi(argv[4]) : 64;
    const char* tag = argc > 5 ? argv[5] : "k1";
    const double geom = argc > 6 ? atof(argv[6]) : 1.0;          // scales rotation and baseline (stress: 4 .. 20)
    const int ncheck = argc > 7 ? atoi(argv[7]) : 2;
    const int P = B * S; const size_t HW = (size_t)H * W;
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::vector<float> ref(B * 3 * HW), src(P * 3 * HW), hmkt(P * 12);
    srand(1);
    for (auto& v : ref) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto& v : src) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    const double fx = 1.125 * W, fy = 1.5 * H, cx = W / 2.0, cy = H / 2.0;
    for (int p = 0; p < P; ++p) {
        const double ry = ((p * 37) % 7 - 3) * 0.01 * geom, rz = ((p * 11) % 5 - 2) * 0.008 * geom, tx = ((p & 1) ? -0.1 : 0.1) * geom;
        const double R[9] = {cos(ry) * cos(rz), -sin(rz), sin(ry), sin(rz), cos(rz), 0, -sin(ry), 0, cos(ry)};
        const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
        double RKi[9], Hm[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * Ki[k * 3 + j]; RKi[i * 3 + j] = s; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * RKi[k * 3 + j]; Hm[i * 3 + j] = s; }
        for (int i = 0; i < 9; ++i) hmkt[p * 12 + i] = (float)Hm[i];
        const double T[3] = {tx, 0.01 * geom, -0.01 * geom};
        for (int i = 0; i < 3; ++i) hmkt[p * 12 + 9 + i] = (float)(K[i * 3] * T[0] + K[i * 3 + 1] * T[1] + K[i * 3 + 2] * T[2]);
    }
    float *dref, *dsrc, *dh, *dout, *dvol, *dws; const size_t wsn = cnm_planesweep_workspace_floats(B, S, H, W); hipMalloc(&dws, wsn * 4 + 16); hipMemset(dws, 0, wsn * 4 + 16);
    const size_t outn = (size_t)P * (D / 4 + 1) * HW * 4, voln = (size_t)P * D * HW;
    hipMalloc(&dref, ref.size() * 4); hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dh, hmkt.size() * 4); hipMalloc(&dout, outn * 4); hipMalloc(&dvol, voln * 4);
    hipMemcpy(dref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, hmkt.data(), hmkt.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, outn * 4); hipMemset(dvol, 0xff, voln * 4);
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, planesweep_kernel<1, SWEEP_STORE_AUX>, SWEEP_NT, 0);
    for (int i = 0; i < 5; ++i) {
        const int rc = cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
        if (rc != 0) { printf("launch failed: %d\n", rc); return 1; }
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    printf("   c4 launches ok\n");
    if (cnm_planesweep_volume_nchw_f32(dref, dsrc, dh, dvol, dws, wsn, B, S, H, W, D, 0.1, 3.0, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) { printf("nchw fault\n"); return 1; }
    printf("   nchw launch ok\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 50;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    float iso_med = 0.f;
    {   // launches ONE AT A TIME (events around each, the device drained in between): what a launch costs between other kernels of a step
        std::vector<float> t;
        for (int i = 0; i < 25; ++i) {
            hipDeviceSynchronize();
            hipEventRecord(e0); cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
            float v; hipEventElapsedTime(&v, e0, e1); t.push_back(v);
        }
        std::sort(t.begin(), t.end()); iso_med = t[t.size() / 2];
    }
    cnm_planesweep_volume_nchw_f32(dref, dsrc, dh, dvol, dws, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
    std::vector<float> out(outn), vol(voln);
    hipMemcpy(out.data(), dout, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(vol.data(), dvol, voln * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < outn; i += 7) cs += out[i];
    // ---- layouts agree bit for bit; ref group present
    size_t nmis = 0;
    for (int p = 0; p < P; ++p) for (int d = 0; d < D; ++d) for (size_t q = 0; q < HW; ++q) {
        const float a = out[c4_offset(p, D / 4 + 1, d >> 2, (int)HW, (int)q) + (d & 3)], b = vol[((size_t)p * D + d) * HW + q];
        if (!(a == b)) ++nmis;
    }
    for (int p = 0; p < P; ++p) for (size_t q = 0; q < HW; q += 5) for (int c = 0; c < 3; ++c)
        if (out[c4_offset(p, D / 4 + 1, D / 4, (int)HW, (int)q) + c] != ref[((size_t)(p / S) * 3 + c) * HW + q]) ++nmis;
    // ---- float64 closed form on the host
    double emax = 0, esum = 0; size_t ecnt = 0, nbig = 0;
    const double step = (3.0 - 0.1) / (D - 1.0);
    for (int p = 0; p < ncheck && p < P; ++p) {
        const float* hk = &hmkt[p * 12];
        for (int d = 0; d < D; ++d) {
            const double z = (double)(float)(1.0 / (0.1 + d * step));
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
                const double t0 = ((double)hk[0] * x + (double)hk[1] * y + hk[2]) * z + hk[9];
                const double t1 = ((double)hk[3] * x + (double)hk[4] * y + hk[5]) * z + hk[10];
                const double t2 = ((double)hk[6] * x + (double)hk[7] * y + hk[8]) * z + hk[11] + 1e-6;
                const double ix = t0 / t2 - 0.5, iy = t1 / t2 - 0.5;
                double c = 0;
                for (int ch = 0; ch < 3; ++ch)
                    c += std::fabs(host_sample(&src[((size_t)p * 3 + ch) * HW], H, W, ix, iy) - ref[((size_t)(p / S) * 3 + ch) * HW + (size_t)y * W + x]);
                const double e = std::fabs(c - out[c4_offset(p, D / 4 + 1, d >> 2, (int)HW, y * W + x) + (d & 3)]);
                if (!(e <= emax)) emax = e;
                if (e > 1e-3) ++nbig;
                esum += e; ++ecnt;
            }
        }
    }
    printf("   check: layouts mismatching %zu | vs float64 closed form (%d pairs): max %.3e  mean %.3e  >1e-3: %zu of %zu\n", nmis, ncheck, emax, esum / (ecnt + 1e-9), nbig, ecnt);
#ifdef SWEEP_TIMELINE
    { unsigned long long tl[2][64]; hipMemcpyFromSymbol(tl, HIP_SYMBOL(sweep_tl), sizeof(tl));
      for (int w = 0; w < 2; ++w) for (int t = 0; t < 3; ++t) { const unsigned long long* q = tl[w] + 8 * t;
        printf("   timeline wave %2d tile %d (cycles of the 100 MHz counter x 24 ~ shader cycles): footprints %5lld  barrier %5lld  to staging %5lld  stage first box %5lld  barrier %5lld  rest of the tile (sweeps, further boxes) %6lld  | tile %6lld\n",
               w ? SWEEP_TH - 1 : 0, t, (long long)(q[1] - q[0]), (long long)(q[2] - q[1]), (long long)(q[3] - q[2]), (long long)(q[4] - q[3]), (long long)(q[5] - q[4]), (long long)(q[6] - q[5]), (long long)(q[6] - q[0])); } }
#endif
#ifdef SWEEP_SPAN
    { static unsigned long long sp[1024][4];
      cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr); hipDeviceSynchronize();   // spans of ONE c4 launch
      hipMemcpyFromSymbol(sp, HIP_SYMBOL(sweep_span), sizeof(sp));
      // s_memrealtime: 100 MHz, one base for the device.  Times in us relative to the earliest kernel entry of any workgroup.
      unsigned long long t0 = ~0ull, t1 = 0; int nwg = 0, hist[16] = {0};
      for (int i = 0; i < 1024; ++i) if (sp[i][3]) { ++nwg; t0 = std::min(t0, sp[i][0]); t1 = std::max(t1, sp[i][2]); }
      std::vector<double> ent, st, en, du;
      for (int i = 0; i < 1024; ++i) if (sp[i][3]) { ent.push_back((sp[i][0] - t0) / 100.0); st.push_back((sp[i][1] - sp[i][0]) / 100.0); en.push_back((t1 - sp[i][2]) / 100.0); du.push_back((sp[i][2] - sp[i][1]) / 100.0); hist[std::min<unsigned long long>(sp[i][3], 15)]++; }
      std::sort(ent.begin(), ent.end()); std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
      auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
      printf("   span (us): %d workgroups, earliest entry -> last tile end %.1f | entry after the earliest: median %.1f p90 %.1f max %.1f | entry -> first tile: median %.1f max %.1f | tiles (busy): min %.1f median %.1f p90 %.1f max %.1f | idle after the last tile until the last workgroup ends: median %.1f p90 %.1f max %.1f | units per workgroup:",
             nwg, (t1 - t0) / 100.0, q(ent, .5), q(ent, .9), q(ent, 1.), q(st, .5), q(st, 1.), q(du, 0.), q(du, .5), q(du, .9), q(du, 1.), q(en, .5), q(en, .9), q(en, 1.));
      for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %dx%d", hist[i], i);
      printf("\n");
      static unsigned int ut[4096]; hipMemcpyFromSymbol(ut, HIP_SYMBOL(sweep_unit_ticks), sizeof(ut));
      const int ntx = (W + 63) / 64, nty = (H + SWEEP_TH - 1) / SWEEP_TH, tpp = ntx * nty;
      if (P * tpp <= 4096) {
          printf("   tile us by pair (mean min max):");
          for (int p = 0; p < P; ++p) { double m = 0, lo = 1e9, hi = 0; for (int t = 0; t < tpp; ++t) { const double v = ut[p * tpp + t] / 100.0; m += v; lo = std::min(lo, v); hi = std::max(hi, v); } printf(" %d: %.1f %.1f %.1f |", p, m / tpp, lo, hi); }
          printf("\n   tile us by tile row (mean over pairs and columns):");
          for (int ty = 0; ty < nty; ++ty) { double m = 0; for (int p = 0; p < P; ++p) for (int tx = 0; tx < ntx; ++tx) m += ut[p * tpp + ty * ntx + tx] / 100.0; printf(" %.1f", m / (P * ntx)); }
          printf("\n   tile us by tile column:");
          for (int tx = 0; tx < ntx; ++tx) { double m = 0; for (int p = 0; p < P; ++p) for (int ty = 0; ty < nty; ++ty) m += ut[p * tpp + ty * ntx + tx] / 100.0; printf(" %.1f", m / (P * nty)); }
          printf("\n");
      } }
#endif
#ifdef SWEEP_EMU
    {   // what a footprint pre-pass and a cost-ordered queue would buy (emulated; see planesweep.hip): the launches above recorded the groups
        const int ntx = (W + 63) / 64, nty = (H + SWEEP_TH - 1) / SWEEP_TH, ntl = P * ntx * nty;
        if (ntl <= 4096) {
            static unsigned int ut[4096]; std::vector<int> order(4096);
            auto run = [&](int mode, const char* what) {
                hipMemcpyToSymbol(HIP_SYMBOL(sweep_emu_mode), &mode, sizeof(int));
                for (int i = 0; i < 5; ++i) cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
                hipEventRecord(e0);
                for (int i = 0; i < iters; ++i) cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                std::vector<float> o2(outn); hipMemcpy(o2.data(), dout, outn * 4, hipMemcpyDeviceToHost);
                size_t bad = 0; for (size_t i = 0; i < outn; ++i) if (!(o2[i] == out[i])) ++bad;
                printf("   emu %-58s %6.1f us   (output differs from the plain launch in %zu values)\n", what, t / iters * 1e3, bad);
            };
            run(0, "plain (footprints worked out per tile, queue in tile order)");
            hipMemcpyFromSymbol(ut, HIP_SYMBOL(sweep_unit_ticks), sizeof(ut));
            for (int i = 0; i < 4096; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.begin() + ntl, [&](int a, int b) { return ut[a] > ut[b]; });
            hipMemcpyToSymbol(HIP_SYMBOL(sweep_emu_order), order.data(), 4096 * sizeof(int));
            run(1, "groups read from a table (a pre-pass would have written it)");
            run(2, "footprints per tile, tiles drawn longest first (recorded durations)");
            run(3, "table + longest first");
            run(4, "box staging skipped (wrong output): a launch whose staging is free");
            run(5, "table + staging skipped (wrong output)");
            run(0, "plain again");
        }
    }
#endif
#ifdef SWEEP_TRACE
    {   // one traced launch -> gpurun_out/k1_trace.txt: workgroup unit hw_id xcc t0 t_foot(ticks) t_end   (100 MHz ticks)
        unsigned int zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(sweep_trace_n), &zero, 4);
        cnm_planesweep_cat_c4_f32(dref, dsrc, dh, dout, K1_WS, wsn, B, S, H, W, D, 0.1, 3.0, nullptr); hipDeviceSynchronize();
        static unsigned long long tr[8192][4]; unsigned int n = 0;
        hipMemcpyFromSymbol(tr, HIP_SYMBOL(sweep_trace), sizeof(tr)); hipMemcpyFromSymbol(&n, HIP_SYMBOL(sweep_trace_n), 4);
        FILE* f = fopen("gpurun_out/k1_trace.txt", "w");
        if (f) { for (unsigned i = 0; i < n && i < 8192; ++i) fprintf(f, "%u %u %u %u %llu %llu %llu\n", (unsigned)(tr[i][0] >> 32), (unsigned)tr[i][0], (unsigned)(tr[i][1] >> 32), (unsigned)tr[i][1], tr[i][2], tr[i][3] & 0xFFFFF, tr[i][3] >> 20); fclose(f); }
        printf("   trace: %u units -> gpurun_out/k1_trace.txt\n", n);
        static unsigned long long tf[1024][6]; hipMemcpyFromSymbol(tf, HIP_SYMBOL(sweep_trace_first), sizeof(tf));
        unsigned long long e0 = ~0ull; for (int i = 0; i < 512; ++i) if (tf[i][0]) e0 = std::min(e0, tf[i][0]);
        for (int st = 1; st < 6; ++st) { std::vector<double> v; for (int i = 0; i < 512; ++i) if (tf[i][0]) v.push_back((double)(long long)(tf[i][st] - tf[i][st - 1]) / 100.0); std::sort(v.begin(), v.end());
            static const char* nm[] = {"", "entry -> loop top", "-> footprints done (wave 0)", "-> barrier passed", "-> first box staged", "-> first octet done"};
            if (!v.empty()) printf("   first unit, %-30s min %.2f median %.2f max %.2f us\n", nm[st], v.front(), v[v.size() / 2], v.back()); }
    }
#endif
#ifdef SWEEP_BOXTIME
    { unsigned long long bt[4]; hipMemcpyFromSymbol(bt, HIP_SYMBOL(sweep_boxtime), 32); const double nl = 5 + iters + 25 + 1, wg = 512;   // launches so far; ticks of 10 ns, thread 0 of each workgroup
      printf("   per workgroup and launch (us, thread 0): footprints + barrier %.1f, box staging %.1f, sweeps %.1f | units per launch %.0f\n",
             bt[0] / nl / wg / 100.0, bt[1] / nl / wg / 100.0, bt[2] / nl / wg / 100.0, bt[3] / nl); }
#endif
#ifdef SWEEP_STATS
    { unsigned int st[4]; hipMemcpyFromSymbol(st, HIP_SYMBOL(sweep_stats), 16); const double nl = 5 + iters + 1;   // launches so far
      printf("   per launch: workgroups %.0f, boxes staged %.0f, octets gathered from global %.0f, texels per box %.0f\n", st[0] / nl, st[1] / nl, st[2] / nl, (double)st[3] / (st[1] + 1e-9)); }
#endif
    if (argc <= 8) {
        dim3 grid(W / 64, H / 4, P);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) ref_store_kernel<<<grid, 256>>>(dout, D / 4 + 1, H * W, W);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); printf("   (pure c4 store stream, same output: %.1f us = %.0f GB/s)\n", t / iters * 1e3, (double)P * (D + 4) * HW * 4 / (t / iters) / 1e6);
    }
    const double bytes = (double)B * 3 * HW * 4 + (double)P * 3 * HW * 4 + (double)P * (D + 3) * HW * 4;
    printf("%-44s %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)  | one at a time: %.1f us (%.1f%%)  blocks/CU %d  checksum %.6e\n", tag, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0,
           iso_med * 1e3, bytes / iso_med / 1e6 / 80.0, nb, cs);
    return 0;
}
