// Host twins ("_cpu") of the C ABI -- SURVEY 8(b): "each with a _cpu twin taking host pointers"; BASELINE configs[0]:
// "DepthNet eval, 1 ref + 1 src, 256x192, 32 depth planes, batch=1 on CPU (plumbing, no GPU)".
//
// Plain C++17, no HIP: the same operators on HOST memory, same argument meaning, same c4 activation layout
// ([N][G][H][W][4] floats), written from the reference's formulas (file:line under the reference checkout next to each).
// They exist so that the product runs the plumbing configuration and CPU tensors without a GPU and WITHOUT importing
// oracle/ (which stays test infrastructure); they are not the measured path and make no performance claim beyond "a frame
// in seconds": direct convolutions, rows vectorised by the compiler (AVX2 + FMA clones where the CPU has them), one
// std::thread per core.  The whole-network executors are nets.hip's launch sequences run on the EngHost policy.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/cnm_engine.h"
#include "host_ops.h"

#define CNMH_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define CNMH_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define CNMH_CLONES
#endif

namespace {

inline size_t c4off(int n, int Gt, int g, int HW, int pix) { return (((size_t)n * Gt + g) * (size_t)HW + pix) * 4; }

// threads = CNM_CPU_THREADS, else min(hardware threads, cgroup v2 CPU quota): a container that shows 256 CPUs under a quota of
// 16 must not get 256 runnable threads
int host_threads() {
    static int n = [] {
        const char* e = std::getenv("CNM_CPU_THREADS");
        if (e) return std::max(1, std::min(std::atoi(e), 256));
        int v = (int)std::thread::hardware_concurrency();
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0}; long long period = 0;
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
                v = std::min<long long>(v, std::max<long long>(1, std::atoll(q) / period));
            std::fclose(f);
        }
        return std::max(1, std::min(v, 256));
    }();
    return n;
}

// fn(i) for i in [0, n): work items are handed out one at a time (they are rows or row blocks: coarse enough).  Nothing may leave the
// library as a C++ exception (include/cnm_engine.h: "never throws"): a worker that throws (std::bad_alloc in a row buffer) or a thread
// that cannot be created is recorded, the remaining items run on the threads that exist, and the entry point returns CNM_ERR_LAUNCH.
thread_local bool g_pf_failed = false;
int pf_result() { const bool f = g_pf_failed; g_pf_failed = false; return f ? CNM_ERR_LAUNCH : CNM_OK; }
void parallel_for(long long n, const std::function<void(long long)>& fn) {
    const int nt = (int)std::min<long long>(host_threads(), n);
    std::atomic<long long> next{0};
    std::atomic<bool> failed{false};
    auto work = [&] {
        for (long long i; (i = next.fetch_add(1)) < n;) {
            try { fn(i); } catch (...) { failed.store(true); }
        }
    };
    std::vector<std::thread> th;
    try {
        th.reserve(nt > 1 ? nt - 1 : 0);
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
    } catch (...) { failed.store(true); }
    work();                                                              // the calling thread is one of the workers
    for (auto& t : th) t.join();
    if (failed.load()) g_pf_failed = true;
}

bool inv_nxn(double* A, double* Ai, int n) {                             // Gauss-Jordan, partial pivoting (as K0 on the device)
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ai[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; ++c) {
        int piv = c; double best = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[r * n + c]) > best) { best = std::fabs(A[r * n + c]); piv = r; }
        if (piv != c) for (int j = 0; j < n; ++j) { std::swap(A[c * n + j], A[piv * n + j]); std::swap(Ai[c * n + j], Ai[piv * n + j]); }
        const double d = 1.0 / A[c * n + c];
        for (int j = 0; j < n; ++j) { A[c * n + j] *= d; Ai[c * n + j] *= d; }
        for (int r = 0; r < n; ++r) if (r != c) {
            const double f = A[r * n + c];
            for (int j = 0; j < n; ++j) { A[r * n + j] -= f * A[c * n + j]; Ai[r * n + j] -= f * Ai[c * n + j]; }
        }
    }
    return true;
}

// acc[x] += w * t[x * stride + off], x in [0, n): the one hot loop of the convolution
CNMH_CLONES void axpy_row(float* __restrict__ acc, const float* __restrict__ t, float w, int n, int stride) {
    if (stride == 1) for (int x = 0; x < n; ++x) acc[x] += w * t[x];
    else for (int x = 0; x < n; ++x) acc[x] += w * t[2 * x];
}

}  // namespace

namespace cnmh {

// depth_util.py:33-52 (process_camera_parameters): Hm = K_r R K_l^-1, KT = K_r T with [R|T] = E_r E_l^-1, in double
int homography(const float* ref_cam, const float* src_cam, float* hmkt, int B, int S) {
    CNMH_REQUIRE(ref_cam && src_cam && hmkt && B > 0 && S > 0, CNM_ERR_BAD_ARG);
    for (int p = 0; p < B * S; ++p) {
        const float* lc = ref_cam + (size_t)(p / S) * 32;
        const float* rc = src_cam + (size_t)p * 32;
        double El[16], Eli[16], Kl[9], Kli[9], rel[16], RKi[9];
        for (int i = 0; i < 16; ++i) El[i] = lc[i];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Kl[i * 3 + j] = lc[16 + i * 4 + j];
        inv_nxn(El, Eli, 4); inv_nxn(Kl, Kli, 3);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += (double)rc[i * 4 + k] * Eli[k * 4 + j]; rel[i * 4 + j] = s; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += rel[i * 4 + k] * Kli[k * 3 + j]; RKi[i * 3 + j] = s; }
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += (double)rc[16 + i * 4 + k] * RKi[k * 3 + j]; hmkt[(size_t)p * 12 + i * 3 + j] = (float)s; }
            double s = 0; for (int k = 0; k < 3; ++k) s += (double)rc[16 + i * 4 + k] * rel[k * 4 + 3];
            hmkt[(size_t)p * 12 + 9 + i] = (float)s;
        }
    }
    return pf_result();
}

// depthNet_model.py:185-224 (getVolume) + :233 (cat): cost[p,d,y,x] = sum_c | bilinear_zero(src[p,c], u'-0.5, v'-0.5) - ref[b,c,y,x] |,
// (u',v') = (t0,t1)/(t2+1e-6), t = Hm (x,y,1) z_d + KT in fp32 as the reference computes it; z_d from python doubles (:193-194,209)
int sweep(const float* ref, const float* src, const float* hmkt, float* out, int B, int S, int H, int W, int D, double idmin, double idmax, int nchw) {
    CNMH_REQUIRE(ref && src && hmkt && out && B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= 128 && (nchw || D % 4 == 0), CNM_ERR_BAD_ARG);   // as sweep_launch (planesweep.hip): 2 .. CNM_MAX_PLANES planes, the c4 layout in groups of four
    const int HW = H * W, G = D / 4 + 1;
    std::vector<float> zd(D);
    const double step = (idmax - idmin) / (D - 1.0);
    for (int d = 0; d < D; ++d) zd[d] = (float)(1.0 / (idmin + (double)d * step));
    parallel_for((long long)B * S * H, [&](long long item) {
        const int y = (int)(item % H), p = (int)(item / H), b = p / S;
        const float* hm = hmkt + (size_t)p * 12;
        const float* r = ref + (size_t)b * 3 * HW;
        const float* s = src + (size_t)p * 3 * HW;
        for (int x = 0; x < W; ++x) {
            const float a0 = hm[0] * x + hm[1] * y + hm[2], a1 = hm[3] * x + hm[4] * y + hm[5], a2 = hm[6] * x + hm[7] * y + hm[8];
            const float rr = r[y * W + x], rg = r[HW + y * W + x], rb = r[2 * HW + y * W + x];
            for (int d = 0; d < D; ++d) {
                const float z = zd[d];
                const float t2 = a2 * z + hm[11] + 1e-6f;
                const float ix = (a0 * z + hm[9]) / t2 - 0.5f, iy = (a1 * z + hm[10]) / t2 - 0.5f;
                float v[3] = {0.f, 0.f, 0.f};
                if (std::fabs(ix) < 1e7f && std::fabs(iy) < 1e7f) {
                    const float fx = std::floor(ix), fy = std::floor(iy);
                    const int xi = (int)fx, yi = (int)fy;
                    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                    const bool x0 = (unsigned)xi < (unsigned)W, x1 = (unsigned)(xi + 1) < (unsigned)W, y0 = (unsigned)yi < (unsigned)H, y1 = (unsigned)(yi + 1) < (unsigned)H;
                    for (int c = 0; c < 3; ++c) {
                        const float* q = s + (size_t)c * HW + (ptrdiff_t)yi * W + xi;
                        float acc = 0.f;
                        if (y0 && x0) acc += wx0 * wy0 * q[0];
                        if (y0 && x1) acc += wx1 * wy0 * q[1];
                        if (y1 && x0) acc += wx0 * wy1 * q[W];
                        if (y1 && x1) acc += wx1 * wy1 * q[W + 1];
                        v[c] = acc;
                    }
                } else if (ix != ix || iy != iy || std::isinf(ix) || std::isinf(iy)) {
                    v[0] = v[1] = v[2] = NAN;                             // grid_sample on non-finite coordinates
                }
                const float cost = std::fabs(v[0] - rr) + std::fabs(v[1] - rg) + std::fabs(v[2] - rb);
                if (nchw) out[((size_t)p * D + d) * HW + y * W + x] = cost;
                else out[c4off(p, G, d >> 2, HW, y * W + x) + (d & 3)] = cost;
            }
            if (!nchw) { float* o = out + c4off(p, G, D / 4, HW, y * W + x); o[0] = rr; o[1] = rg; o[2] = rb; o[3] = 0.f; }
        }
    });
    return pf_result();
}

// Conv2d(k, stride, padding = k/2, BatchNorm folded) + ReLU on c4 views (depthNet_model.py:19-79).  w [Cout][k*k][4*(Ga+Gb)].
int conv(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
         float* out, int Gout_total, int gout0, int Cout, const float* w, const float* bias, int N, int H, int W, int ksize, int stride, int relu) {
    CNMH_REQUIRE(in_a && out && w && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0 && (Gb == 0 || in_b) && Cout > 0 && Cout % 4 == 0, CNM_ERR_BAD_ARG);
    CNMH_REQUIRE((ksize == 3 || ksize == 5 || ksize == 7) && (stride == 1 || stride == 2), CNM_ERR_BAD_ARG);
    const int pad = ksize / 2, Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, G = Ga + Gb, K4 = 4 * G;
    const int HW = H * W, HoWo = Ho * Wo, CB = 16, ncb = (Cout + CB - 1) / CB, WP = W + 2 * pad + 2;
    parallel_for((long long)N * Ho * ncb, [&](long long item) {
        const int cb = (int)(item % ncb), oy = (int)((item / ncb) % Ho), n = (int)(item / ((long long)ncb * Ho));
        const int c0 = cb * CB, nc = std::min(CB, Cout - c0);
        std::vector<float> acc((size_t)CB * Wo), t((size_t)4 * WP);
        for (int c = 0; c < nc; ++c) std::fill(acc.begin() + (size_t)c * Wo, acc.begin() + (size_t)(c + 1) * Wo, bias ? bias[c0 + c] : 0.f);
        for (int ky = 0; ky < ksize; ++ky) {
            const int iy = oy * stride + ky - pad;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int g = 0; g < G; ++g) {
                const float* row = g < Ga ? in_a + c4off(n, Ga_total, ga0 + g, HW, iy * W) : in_b + c4off(n, Gb_total, gb0 + g - Ga, HW, iy * W);
                std::fill(t.begin(), t.end(), 0.f);                       // planar copy of the row with its zero padding
                for (int x = 0; x < W; ++x) for (int j = 0; j < 4; ++j) t[(size_t)j * WP + pad + x] = row[4 * x + j];
                for (int c = 0; c < nc; ++c) {
                    const float* wk = w + ((size_t)(c0 + c) * ksize * ksize + (size_t)ky * ksize) * K4 + 4 * g;
                    float* a = acc.data() + (size_t)c * Wo;
                    for (int kx = 0; kx < ksize; ++kx)
                        for (int j = 0; j < 4; ++j) {
                            const float wv = wk[(size_t)kx * K4 + j];
                            if (wv != 0.f) axpy_row(a, t.data() + (size_t)j * WP + kx, wv, Wo, stride);
                        }
                }
            }
        }
        for (int c = 0; c < nc; ++c) {
            const int co = c0 + c;
            float* o = out + c4off(n, Gout_total, gout0 + (co >> 2), HoWo, oy * Wo) + (co & 3);
            const float* a = acc.data() + (size_t)c * Wo;
            for (int x = 0; x < Wo; ++x) o[4 * x] = relu ? std::max(a[x], 0.f) : a[x];
        }
    });
    return pf_result();
}

// nn.Upsample(scale_factor=2, mode='bilinear'), align_corners=False (depthNet_model.py:94,105)
int upsample2x(const float* in, int Gin_total, int gin0, float* out, int Gout_total, int gout0, int N, int G, int H, int W) {
    CNMH_REQUIRE(in && out && N > 0 && G > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const int HW = H * W, Ho = 2 * H, Wo = 2 * W;
    parallel_for((long long)N * G * Ho, [&](long long item) {
        const int oy = (int)(item % Ho), g = (int)((item / Ho) % G), n = (int)(item / ((long long)Ho * G));
        const float sy = std::max((oy + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, y1 = std::min(y0 + 1, H - 1);
        const float ly = sy - y0;
        const float* r0 = in + c4off(n, Gin_total, gin0 + g, HW, y0 * W);
        const float* r1 = in + c4off(n, Gin_total, gin0 + g, HW, y1 * W);
        float* o = out + c4off(n, Gout_total, gout0 + g, 4 * HW, oy * Wo);
        for (int ox = 0; ox < Wo; ++ox) {
            const float sx = std::max((ox + 0.5f) * 0.5f - 0.5f, 0.f);
            const int x0 = (int)sx, x1 = std::min(x0 + 1, W - 1);
            const float lx = sx - x0;
            for (int j = 0; j < 4; ++j) {
                const float top = r0[4 * x0 + j] + lx * (r0[4 * x1 + j] - r0[4 * x0 + j]);
                const float bot = r1[4 * x0 + j] + lx * (r1[4 * x1 + j] - r1[4 * x0 + j]);
                o[4 * ox + j] = top + ly * (bot - top);
            }
        }
    });
    return pf_result();
}

// depth_layer + scale + F.upsample(nearest) (depthNet_model.py:82-84,246-261,351,365)
int head(const float* in, int Gin_total, int gin0, int C, const float* w_head, const float* bias, float scale,
         float* disp, float* up_out, int up_Gtotal, int up_g, int N, int H, int W) {
    CNMH_REQUIRE(in && w_head && bias && disp && N > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const int HW = H * W, G = C / 4;
    parallel_for((long long)N * H, [&](long long item) {
        const int y = (int)(item % H), n = (int)(item / H);
        std::vector<double> acc(W, (double)bias[0]);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int g = 0; g < G; ++g) {
                const float* row = in + c4off(n, Gin_total, gin0 + g, HW, iy * W);
                for (int kx = 0; kx < 3; ++kx) {
                    const float* wt = w_head + (size_t)(ky * 3 + kx) * C + 4 * g;
                    for (int x = 0; x < W; ++x) {
                        const int ix = x + kx - 1;
                        if ((unsigned)ix >= (unsigned)W) continue;
                        const float* q = row + 4 * ix;
                        acc[x] += (double)(wt[0] * q[0]) + (double)(wt[1] * q[1]) + (double)(wt[2] * q[2]) + (double)(wt[3] * q[3]);
                    }
                }
            }
        }
        for (int x = 0; x < W; ++x) {
            const float v = scale / (1.f + std::exp(-(float)acc[x]));
            disp[(size_t)n * HW + y * W + x] = v;
            if (up_out)
                for (int a = 0; a < 2; ++a) for (int b2 = 0; b2 < 2; ++b2) {
                    float* o = up_out + c4off(n, up_Gtotal, up_g, 4 * HW, (2 * y + a) * 2 * W + 2 * x + b2);
                    o[0] = v; o[1] = o[2] = o[3] = 0.f;
                }
        }
    });
    return pf_result();
}

// depthNet_model.py:332-333 (channels rotated: 64 features first): x = [f1 + f2, (id1, id2, |id1 - id2|, 0)]
int assemble(const float* id1, const float* id2, long long ids, const float* f1, int G1t, int g1, const float* f2, int G2t, int g2,
             float* x, int N, int C, int H, int W) {
    CNMH_REQUIRE(id1 && id2 && f1 && f2 && x && N > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0 && ids >= (long long)H * W, CNM_ERR_BAD_ARG);
    const int HW = H * W, G = C / 4;
    parallel_for((long long)N * (G + 1), [&](long long item) {
        const int g = (int)(item % (G + 1)), n = (int)(item / (G + 1));
        float* o = x + c4off(n, G + 1, g, HW, 0);
        if (g < G) {
            const float* a = f1 + c4off(n, G1t, g1 + g, HW, 0); const float* b = f2 + c4off(n, G2t, g2 + g, HW, 0);
            for (int i = 0; i < 4 * HW; ++i) o[i] = a[i] + b[i];
        } else {
            for (int p = 0; p < HW; ++p) { const float a = id1[(size_t)n * ids + p], b = id2[(size_t)n * ids + p]; o[4 * p] = a; o[4 * p + 1] = b; o[4 * p + 2] = std::fabs(a - b); o[4 * p + 3] = 0.f; }
        }
    });
    return pf_result();
}

// eval.py:656-663 (S = 4), :917-929 (S = 6): even sources -> side 1, odd -> side 2, averaged, then assembled
int assemble_multi(const float* idp, const float* f, float* x, int B, int S, int C, int H, int W) {
    CNMH_REQUIRE(idp && f && x && B > 0 && S >= 2 && S % 2 == 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const int HW = H * W, G = C / 4, h = S / 2;
    parallel_for((long long)B * (G + 1), [&](long long item) {
        const int g = (int)(item % (G + 1)), b = (int)(item / (G + 1));
        float* o = x + c4off(b, G + 1, g, HW, 0);
        auto avg = [&](float s) { return h == 1 ? s : h == 2 ? s * 0.5f : s / (float)h; };
        if (g < G) {
            for (int i = 0; i < 4 * HW; ++i) {
                float s1 = 0.f, s2 = 0.f;
                for (int k = 0; k < h; ++k) { s1 += f[c4off(b * S + 2 * k, G, g, HW, 0) + i]; s2 += f[c4off(b * S + 2 * k + 1, G, g, HW, 0) + i]; }
                o[i] = avg(s1) + avg(s2);
            }
        } else {
            for (int p = 0; p < HW; ++p) {
                float s1 = 0.f, s2 = 0.f;
                for (int k = 0; k < h; ++k) { s1 += idp[(size_t)(b * S + 2 * k) * HW + p]; s2 += idp[(size_t)(b * S + 2 * k + 1) * HW + p]; }
                const float a = avg(s1), c = avg(s2);
                o[4 * p] = a; o[4 * p + 1] = c; o[4 * p + 2] = std::fabs(a - c); o[4 * p + 3] = 0.f;
            }
        }
    });
    return pf_result();
}

}  // namespace cnmh

// ------------------------------------------------------------------ C ABI
extern "C" {

int cnm_homography_terms_cpu(const float* ref_cam, const float* src_cam, float* hmkt, int B, int S) { return cnmh::homography(ref_cam, src_cam, hmkt, B, S); }

int cnm_planesweep_volume_nchw_cpu(const float* ref, const float* src, const float* hmkt, float* volume, int B, int S, int H, int W, int D,
                                   double idepth_min, double idepth_max) {
    return cnmh::sweep(ref, src, hmkt, volume, B, S, H, W, D, idepth_min, idepth_max, 1);
}

int cnm_planesweep_cat_c4_cpu(const float* ref, const float* src, const float* hmkt, float* x, int B, int S, int H, int W, int D,
                              double idepth_min, double idepth_max) {
    return cnmh::sweep(ref, src, hmkt, x, B, S, H, W, D, idepth_min, idepth_max, 0);
}

size_t cnm_packed_conv_floats_cpu(int Cout, int Cin, int ksize) { return (size_t)Cout * ksize * ksize * 4 * ((Cin + 3) / 4); }

// Eval-mode BatchNorm folded into the filter (fp64 scale, as cnm_pack_conv_bn_f32): w_packed [Cout][k*k][4*ceil(Cin/4)], channel
// position (ci + Cin - rot) % Cin (the first layers' inputs carry their 3 image / map channels LAST); b_packed [Cout].
int cnm_pack_conv_bn_cpu(const float* w_oihw, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                         const float* bias, float eps, int Cout, int Cin, int ksize, int rot, float* w_packed, float* b_packed) {
    CNMH_REQUIRE(w_oihw && w_packed && b_packed && Cout > 0 && Cin > 0 && ksize > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    const int K4 = 4 * ((Cin + 3) / 4), kk = ksize * ksize;
    std::fill(w_packed, w_packed + (size_t)Cout * kk * K4, 0.f);
    for (int co = 0; co < Cout; ++co) {
        const double sc = bn_gamma ? (double)bn_gamma[co] / std::sqrt((double)bn_var[co] + (double)eps) : 1.0;
        for (int ci = 0; ci < Cin; ++ci) {
            const int cp = (ci + Cin - rot) % Cin;
            for (int t = 0; t < kk; ++t) w_packed[((size_t)co * kk + t) * K4 + cp] = (float)((double)w_oihw[((size_t)co * Cin + ci) * kk + t] * sc);
        }
        b_packed[co] = bn_gamma ? (float)((double)bn_beta[co] - (double)bn_mean[co] * sc) : (bias ? bias[co] : 0.f);
    }
    return pf_result();
}

int cnm_pack_head_cpu(const float* w_oihw, int C, float* w_head) {
    CNMH_REQUIRE(w_oihw && w_head && C > 0 && C % 4 == 0, CNM_ERR_BAD_ARG);
    for (int t = 0; t < 9; ++t) for (int c = 0; c < C; ++c) w_head[t * C + c] = w_oihw[c * 9 + t];
    return pf_result();
}

int cnm_conv2d_cat2_c4_cpu(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
                           float* out, int Gout_total, int gout0, int Cout, const float* w_packed, const float* b_packed,
                           int N, int H, int W, int ksize, int stride, int relu) {
    return cnmh::conv(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, w_packed, b_packed, N, H, W, ksize, stride, relu);
}

int cnm_upsample2x_c4_cpu(const float* in, int Gin_total, int gin0, float* out, int Gout_total, int gout0, int N, int G, int H, int W) {
    return cnmh::upsample2x(in, Gin_total, gin0, out, Gout_total, gout0, N, G, H, W);
}

int cnm_head_sigmoid_c4_cpu(const float* in, int Gin_total, int gin0, int C, const float* w_head, const float* bias, float scale,
                            float* disp, float* up_out, int up_Gtotal, int up_g, int N, int H, int W) {
    return cnmh::head(in, Gin_total, gin0, C, w_head, bias, scale, disp, up_out, up_Gtotal, up_g, N, H, W);
}

int cnm_nchw_to_c4_cpu(const float* nchw, float* c4, int G_total, int g0, int N, int C, int H, int W) {
    CNMH_REQUIRE(nchw && c4 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 3) / 4 <= G_total, CNM_ERR_BAD_ARG);
    const int HW = H * W, G = (C + 3) / 4;
    parallel_for((long long)N * G, [&](long long item) {
        const int g = (int)(item % G), n = (int)(item / G);
        float* o = c4 + c4off(n, G_total, g0 + g, HW, 0);
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * g + j;
            const float* s = c < C ? nchw + ((size_t)n * C + c) * HW : nullptr;
            for (int p = 0; p < HW; ++p) o[4 * p + j] = s ? s[p] : 0.f;
        }
    });
    return pf_result();
}

int cnm_c4_to_nchw_cpu(const float* c4, int G_total, int g0, float* nchw, int N, int C, int H, int W) {
    CNMH_REQUIRE(nchw && c4 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 3) / 4 <= G_total, CNM_ERR_BAD_ARG);
    const int HW = H * W;
    parallel_for((long long)N * C, [&](long long item) {
        const int c = (int)(item % C), n = (int)(item / C);
        const float* s = c4 + c4off(n, G_total, g0 + (c >> 2), HW, 0) + (c & 3);
        float* o = nchw + ((size_t)n * C + c) * HW;
        for (int p = 0; p < HW; ++p) o[p] = s[4 * p];
    });
    return pf_result();
}

int cnm_intrinsics_inverse_cpu(const float* cam, long long cam_stride, float* K_inv, int B) {
    CNMH_REQUIRE(cam && K_inv && B > 0 && cam_stride >= 32, CNM_ERR_BAD_ARG);
    for (int b = 0; b < B; ++b) {
        double K[9], Ki[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K[i * 3 + j] = cam[(size_t)b * cam_stride + 16 + i * 4 + j];
        inv_nxn(K, Ki, 3);
        for (int i = 0; i < 9; ++i) K_inv[(size_t)b * 9 + i] = (float)Ki[i];
    }
    return pf_result();
}

// Depth2normal.forward without the plane branch (depth_util.py:149-203): window sums and solve in double
int cnm_depth2normal_cpu(const float* depth, const float* K_inv, float* normal, float* points, int B, int H, int W, int ksize, int input_is_idepth) {
    CNMH_REQUIRE(depth && K_inv && normal && points && B > 0 && H > 0 && W > 0 && ksize >= 1 && (ksize & 1) && ksize <= 15, CNM_ERR_BAD_ARG);
    const int HW = H * W, r = ksize / 2;
    std::vector<float> pts((size_t)B * HW * 4);
    parallel_for((long long)B * H, [&](long long item) {
        const int y = (int)(item % H), b = (int)(item / H);
        const float* ki = K_inv + (size_t)b * 9;
        for (int x = 0; x < W; ++x) {
            float z = depth[(size_t)b * HW + y * W + x];
            if (input_is_idepth) z = 1.0f / z;
            const float px = (ki[0] * x + ki[1] * y + ki[2]) * z, py = (ki[3] * x + ki[4] * y + ki[5]) * z, pz = (ki[6] * x + ki[7] * y + ki[8]) * z;
            const size_t o = (size_t)b * 3 * HW + y * W + x;
            points[o] = px; points[o + HW] = py; points[o + 2 * (size_t)HW] = pz;
            float* q = pts.data() + ((size_t)b * HW + y * W + x) * 4;
            const bool ok = z > 0.f && z < 10.0f;
            q[0] = ok ? px : 0.f; q[1] = ok ? py : 0.f; q[2] = ok ? pz : 0.f; q[3] = ok ? 1.f : 0.f;
        }
    });
    parallel_for((long long)B * H, [&](long long item) {
        const int y = (int)(item % H), b = (int)(item / H);
        for (int x = 0; x < W; ++x) {
            double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0, sx = 0, sy = 0, sz = 0;
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = -r; dx <= r; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    const float* q = pts.data() + ((size_t)b * HW + yy * W + xx) * 4;
                    const double px = q[0], py = q[1], pz = q[2];
                    sxx += px * px; sxy += px * py; sxz += px * pz; syy += py * py; syz += py * pz; szz += pz * pz; sx += px; sy += py; sz += pz;
                }
            }
            const double c00 = syy * szz - syz * syz, c01 = sxz * syz - sxy * szz, c02 = sxy * syz - sxz * syy;
            const double c11 = sxx * szz - sxz * sxz, c12 = sxy * sxz - sxx * syz, c22 = sxx * syy - sxy * sxy;
            const double det = sxx * c00 + sxy * c01 + sxz * c02;
            double gx, gy, gz;
            if (!(det >= 1e-5)) { gx = sx; gy = sy; gz = sz; }             // depth_util.py:185-198: S := I
            else { const double id = 1.0 / det; gx = (c00 * sx + c01 * sy + c02 * sz) * id; gy = (c01 * sx + c11 * sy + c12 * sz) * id; gz = (c02 * sx + c12 * sy + c22 * sz) * id; }
            const double inv = 1.0 / (std::sqrt(gx * gx + gy * gy + gz * gz) + 1e-5);   // :201
            const size_t o = (size_t)b * 3 * HW + y * W + x;
            normal[o] = (float)(gx * inv); normal[o + HW] = (float)(gy * inv); normal[o + 2 * (size_t)HW] = (float)(gz * inv);
        }
    });
    return pf_result();
}

// inverse_warp / pixel2cam / cam2pixel, padding_mode = 'zeros' (inverse_warp.py:27-118)
int cnm_inverse_warp_cpu(const float* feat, const float* depth, const float* pose, const float* K, const float* K_inv, float* out, int B, int C, int H, int W) {
    CNMH_REQUIRE(feat && depth && pose && K && K_inv && out && B > 0 && C > 0 && H > 1 && W > 1, CNM_ERR_BAD_ARG);
    const int HW = H * W;
    parallel_for((long long)B * H, [&](long long item) {
        const int y = (int)(item % H), b = (int)(item / H);
        const float* ki = K_inv + (size_t)b * 9; const float* kk = K + (size_t)b * 9; const float* ps = pose + (size_t)b * 12;
        float P[12];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) P[i * 4 + j] = kk[i * 3] * ps[j] + kk[i * 3 + 1] * ps[4 + j] + kk[i * 3 + 2] * ps[8 + j];   // :110
        for (int x = 0; x < W; ++x) {
            const float z = depth[(size_t)b * HW + y * W + x];
            const float cx = (ki[0] * x + ki[1] * y + ki[2]) * z, cy = (ki[3] * x + ki[4] * y + ki[5]) * z, cz = (ki[6] * x + ki[7] * y + ki[8]) * z;
            const float X = P[0] * cx + P[1] * cy + P[2] * cz + P[3], Y = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
            const float Z = std::max(P[8] * cx + P[9] * cy + P[10] * cz + P[11], 1e-3f);        // :67
            float xn = 2.f * (X / Z) / (float)(W - 1) - 1.f, yn = 2.f * (Y / Z) / (float)(H - 1) - 1.f;   // :69-70
            if (xn > 1.f || xn < -1.f) xn = 2.f;                                                   // :71-75
            if (yn > 1.f || yn < -1.f) yn = 2.f;
            const float ix = ((xn + 1.f) * W - 1.f) * 0.5f, iy = ((yn + 1.f) * H - 1.f) * 0.5f;    // grid_sample, align_corners=False
            float w00 = 0, w01 = 0, w10 = 0, w11 = 0; int xi = 0, yi = 0; bool x0 = false, x1 = false, y0 = false, y1 = false;
            if (std::fabs(ix) < 1e7f && std::fabs(iy) < 1e7f) {
                const float fx = std::floor(ix), fy = std::floor(iy);
                const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                xi = (int)fx; yi = (int)fy;
                x0 = (unsigned)xi < (unsigned)W; x1 = (unsigned)(xi + 1) < (unsigned)W; y0 = (unsigned)yi < (unsigned)H; y1 = (unsigned)(yi + 1) < (unsigned)H;
                w00 = wx0 * wy0; w01 = wx1 * wy0; w10 = wx0 * wy1; w11 = wx1 * wy1;
            }
            for (int c = 0; c < C; ++c) {
                const float* s = feat + ((size_t)b * C + c) * HW + (ptrdiff_t)yi * W + xi;
                float v = 0.f;
                if (y0 && x0) v = w00 * s[0];
                if (y0 && x1) v += w01 * s[1];
                if (y1 && x0) v += w10 * s[W];
                if (y1 && x1) v += w11 * s[W + 1];
                out[((size_t)b * C + c) * HW + y * W + x] = v;
            }
        }
    });
    return pf_result();
}

}  // extern "C"
