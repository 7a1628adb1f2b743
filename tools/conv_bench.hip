// Standalone micro-benchmark of the MFMA conv kernel (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] tools/conv_bench.hip -o /tmp/cb && /tmp/cb
#include "../cnmnet_amd/csrc/conv_mfma.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Shape { const char* name; int N, H, W, Cin, Cout, k, s; };

int main(int argc, char** argv) {
    const Shape shapes[] = {
        {"conv1.0 67->128 k7 s1 192x256 x16", 16, 192, 256, 67, 128, 7, 1},
        {"conv1.3 128->128 k7 s2 192x256 x16", 16, 192, 256, 128, 128, 7, 2},
        {"conv2.0 128->256 k5 s1 96x128 x16", 16, 96, 128, 128, 256, 5, 1},
        {"conv3.0 256->512 k3 s1 48x64 x16", 16, 48, 64, 256, 512, 3, 1},
        {"iconv4.0 1024->512 k3 s1 24x32 x16", 16, 24, 32, 1024, 512, 3, 1},
        {"upconv1.1 128->64 k3 s1 192x256 x16", 16, 192, 256, 128, 64, 3, 1},
        {"iconv5.0 1024->512 k3 s1 12x16 x16", 16, 12, 16, 1024, 512, 3, 1},
        {"conv5.3 512->512 k3 s2 12x16 x16", 16, 12, 16, 512, 512, 3, 2},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    double tot_f = 0, tot_ms = 0;
    for (int si = 0; si < (int)(sizeof(shapes) / sizeof(shapes[0])); ++si) {
        if (only >= 0 && si != only) continue;
        const Shape& S = shapes[si];
        const int Gin = (S.Cin + 3) / 4, Ho = S.H / S.s, Wo = S.W / S.s;
        const size_t nin = (size_t)S.N * Gin * S.H * S.W * 4, nout = (size_t)S.N * (S.Cout / 4) * Ho * Wo * 4;
        const size_t nw = cnm_packed_conv_floats(S.Cout, S.Cin, S.k);
        std::vector<float> hin(nin), hw(nw), hb(S.Cout, 0.1f);
        for (auto& v : hin) v = (rand() / (float)RAND_MAX - 0.5f);
        for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
        float *din, *dout, *dw, *db;
        hipMalloc(&din, nin * 4); hipMalloc(&dout, nout * 4); hipMalloc(&dw, nw * 4); hipMalloc(&db, S.Cout * 4);
        hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), S.Cout * 4, hipMemcpyHostToDevice);
        auto run = [&]() { return cnm_conv2d_c4_f32(din, Gin, 0, Gin, dout, S.Cout / 4, 0, S.Cout, dw, db, S.N, S.H, S.W, S.k, S.s, 1, nullptr); };
        if (run() != 0) { printf("launch failed\n"); return 1; }
        run();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 5;
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
        std::vector<float> hout(nout); hipMemcpy(hout.data(), dout, nout * 4, hipMemcpyDeviceToHost);
        {   // determinism: a second run must be bitwise identical
            hipMemset(dout, 0xff, nout * 4); run();
            std::vector<float> h2(nout); hipMemcpy(h2.data(), dout, nout * 4, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0; for (size_t i = 0; i < nout; ++i) if (memcmp(&h2[i], &hout[i], 4)) { if (!bad) first = i; ++bad; }
            if (bad) printf("   NONDETERMINISTIC: %zu of %zu outputs differ (first at %zu: %g vs %g)\n", bad, nout, first, hout[first], h2[first]);
        }
        double cs = 0; for (size_t i = 0; i < nout; i += 97) cs += hout[i];
        const double gf = 2.0 * S.Cout * S.Cin * S.k * S.k * Ho * Wo * S.N / 1e9;
        printf("%-40s %8.3f ms %7.1f TFLOP/s  checksum %.6e\n", S.name, ms, gf / ms, cs);
        tot_f += gf; tot_ms += ms;
        hipFree(din); hipFree(dout); hipFree(dw); hipFree(db);
    }
    printf("%-40s %8.3f ms %7.1f TFLOP/s\n", argc > 2 ? argv[2] : "total", tot_ms, tot_f / tot_ms);
    return 0;
}
