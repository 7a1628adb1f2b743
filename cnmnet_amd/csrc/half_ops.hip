// fp16 path (BASELINE config 5): the memory-bound operators on "c8" activations -- [N][G][H][W][8 halfs],
// channel c = 8g + j, 16 bytes per (pixel, group) exactly like the fp32 c4 layout, so every offset helper is shared.
// Arithmetic inside each kernel is fp32; only storage is fp16.
#include "cnm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 ld8(const float* base, size_t chunk_off) {      // chunk_off in floats (16 B = 4 floats)
    return *reinterpret_cast<const f16x8*>(base + chunk_off);
}
__device__ __forceinline__ void st8(float* base, size_t chunk_off, f16x8 v) {
    *reinterpret_cast<f16x8*>(base + chunk_off) = v;
}

// ------------------------------------------------------------------ bilinear x2
// One thread per LOW-resolution (pixel, channel group): its 2 x 2 output pixels come from the 3 x 3 neighbourhood (9 loads
// and 72 conversions per 4 outputs instead of 16 and 128; one 32-bit index decomposition per 4 outputs -- the
// one-thread-per-output version with 64-bit div / mod ran at a third of the HBM rate).  Same expression per output as
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) / 2 - 0.5 clamped at 0, i.e. weights
// 0.75 / 0.25, and weight 0 on the clamped neighbour at the first row / column.
__global__ __launch_bounds__(256) void upsample2x_c8h_kernel(const float* __restrict__ in, int Gin_tot, int gin0,
                                                             float* __restrict__ out, int Gout_tot, int gout0,
                                                             int N, int G, int H, int W) {
    const int Wo = 2 * W, Ho = 2 * H;
    const unsigned total = (unsigned)N * G * H * W;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int x = (int)(idx % (unsigned)W);
        unsigned r = idx / (unsigned)W;
        const int y = (int)(r % (unsigned)H); r /= (unsigned)H;
        const int g = (int)(r % (unsigned)G), n = (int)(r / (unsigned)G);
        const int ym = max(y - 1, 0), yp = min(y + 1, H - 1), xm = max(x - 1, 0), xp = min(x + 1, W - 1);
        const size_t b = c4_offset(n, Gin_tot, gin0 + g, H * W, 0);
        float p[3][3][8];
        const int ys[3] = {ym, y, yp}, xs[3] = {xm, x, xp};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f16x8 v = ld8(in, b + (size_t)(ys[i] * W + xs[j]) * 4);
#pragma unroll
                for (int c = 0; c < 8; ++c) p[i][j][c] = (float)v[c];
            }
        // output row 2y + a reads rows (a, a + 1) of the neighbourhood with ly = 0.75 / 0.25; at y = 0 the reference reads
        // (row 0, row 1) with ly = 0 -- here (row 0, row 0) with ly = 0: the same value for finite data.  Columns alike.
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float ly = a ? 0.25f : (y > 0 ? 0.75f : 0.f), lx = bb ? 0.25f : (x > 0 ? 0.75f : 0.f), hy = 1.f - ly, hx = 1.f - lx;
                f16x8 v;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[c] = (_Float16)(hy * (hx * p[a][bb][c] + lx * p[a][bb + 1][c]) + ly * (hx * p[a + 1][bb][c] + lx * p[a + 1][bb + 1][c]));
                st8(out, c4_offset(n, Gout_tot, gout0 + g, Ho * Wo, (2 * y + a) * Wo + 2 * x + bb), v);
            }
    }
}

extern "C" int cnm_upsample2x_c8_f16(const void* in, int Gin_total, int gin0, void* out, int Gout_total, int gout0,
                                     int N, int G, int H, int W, void* stream) {
    CNM_REQUIRE(in && out && N > 0 && G > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(gin0 >= 0 && gin0 + G <= Gin_total && gout0 >= 0 && gout0 + G <= Gout_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * G * H * W;
    CNM_REQUIRE(total < (1ll << 31), CNM_ERR_BAD_ARG);
    upsample2x_c8h_kernel<<<(int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384), 256, 0, cnm_stream(stream)>>>(
        static_cast<const float*>(in), Gin_total, gin0, static_cast<float*>(out), Gout_total, gout0, N, G, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ disparity head (fp32 weights, fp32 output)
// Workgroup = 64 consecutive pixels x NS channel slices (one wave per slice), partial sums meet in LDS in a fixed order
// (same scheme as head_sigmoid_c4_kernel; one lane per pixel over all channels was a 4608-tap serial loop on the 512-channel head).
template <int NS>
__global__ __launch_bounds__(64 * NS) void head_sigmoid_c8h_kernel(const float* __restrict__ in, int Gin_tot, int gin0, int G,
                                                                   const float* __restrict__ wh, const float* __restrict__ bias,
                                                                   float scale, float* __restrict__ disp,
                                                                   float* __restrict__ up_out, int up_Gtot, int up_g,
                                                                   int N, int H, int W) {
    __shared__ float part[NS][64];
    const int HW = H * W;
    const long long total = (long long)N * HW;
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long idx = (long long)blockIdx.x * 64 + lane;
    const bool live = idx < total;
    const long long ii = live ? idx : 0;
    const int n = (int)(ii / HW), pix = (int)(ii - (long long)n * HW);
    const int y = pix / W, x = pix - y * W;
    const int gper = (G + NS - 1) / NS, gbeg = slice * gper, gend = min(G, gbeg + gper);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = gbeg; g < gend; ++g) {
        const size_t b = c4_offset(n, Gin_tot, gin0 + g, HW, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + kx - 1;
                if (!live || (unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const f16x8 v = ld8(in, b + (size_t)(iy * W + ix) * 4);
                const float* w = wh + (size_t)(ky * 3 + kx) * (G * 8) + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 3] = fmaf((float)v[j], w[j], acc[j & 3]);
            }
        }
    }
    part[slice][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (slice != 0 || !live) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NS; q += 4) s += (part[q][lane] + part[q + 1][lane]) + (part[q + 2][lane] + part[q + 3][lane]);
    s += bias[0];
    const float d = scale / (1.f + expf(-s));
    disp[idx] = d;
    if (up_out) {
        const int Wo = 2 * W;
        f16x8 v = {(_Float16)d, 0, 0, 0, 0, 0, 0, 0};
        const size_t o = c4_offset(n, up_Gtot, up_g, 4 * HW, (2 * y) * Wo + 2 * x);
        st8(up_out, o, v); st8(up_out, o + 4, v); st8(up_out, o + (size_t)Wo * 4, v); st8(up_out, o + (size_t)Wo * 4 + 4, v);
    }
}

// The same head on the high-resolution levels [r5]: a workgroup walks HR output rows of 62 columns (lane = column, lanes 0 / 63 are halo), every
// (pixel, group) chunk is loaded ONCE -- 10 row loads per 8 output rows, the left / right neighbours come from the adjacent lanes by DPP on the
// packed halfs -- where the kernel above loads its nine taps separately (41.7 us per launch at the 192 x 256 levels against a 6-13 us HBM floor).
// Four waves split the channel groups; partial sums meet in LDS in a fixed order.  Same scheme as head_sigmoid_c4_rows_kernel (pointwise.hip).
typedef unsigned int hd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hd_shr1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }   // wave_shr:1 -> lane l gets lane l - 1
__device__ __forceinline__ unsigned hd_shl1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }   // wave_shl:1 -> lane l gets lane l + 1
template <int HR>
__global__ __launch_bounds__(256) void head_sigmoid_c8h_rows_kernel(const float* __restrict__ in, int Gin_tot, int gin0, int G,
                                                                    const float* __restrict__ wh, const float* __restrict__ bias,
                                                                    float scale, float* __restrict__ disp,
                                                                    float* __restrict__ up_out, int up_Gtot, int up_g,
                                                                    int N, int H, int W, int tilesX, int tilesY) {
    __shared__ float part[4][HR][64];
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY, n = b / tilesY;
    const int x = tx * 62 - 1 + lane, y0 = ty * HR;
    const bool xin = (unsigned)x < (unsigned)W;
    const int gper = (G + 3) / 4, gbeg = slice * gper, gend = min(G, gbeg + gper);
    float acc[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) acc[r] = 0.f;
    for (int g = gbeg; g < gend; ++g) {
        const hd_u32x4* base = reinterpret_cast<const hd_u32x4*>(in + c4_offset(n, Gin_tot, gin0 + g, HW, 0));
        hd_u32x4 row[HR + 2];
#pragma unroll
        for (int i = 0; i < HR + 2; ++i) {                               // input rows y0-1 .. y0+HR: all loads first
            const int iy = y0 + i - 1;
            row[i] = (xin && (unsigned)iy < (unsigned)H) ? base[iy * W + x] : hd_u32x4{0u, 0u, 0u, 0u};
        }
        float wk[9][8];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) wk[k][j] = wh[(size_t)k * (G * 8) + g * 8 + j];   // wave-uniform
#pragma unroll
        for (int i = 0; i < HR + 2; ++i) {
            union U { hd_u32x4 u; f16x8 h; } c, l, r;
            c.u = row[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) { l.u[q] = hd_shr1(c.u[q]); r.u[q] = hd_shl1(c.u[q]); }   // columns x-1 / x+1 (lanes 0 / 63 get garbage: halo lanes, no output)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int o = i - ky;                                    // output row of the tile this input row feeds through kernel row ky
                if (o < 0 || o >= HR) continue;
                float a = acc[o];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a = fmaf((float)l.h[j], wk[ky * 3 + 0][j], a);
                    a = fmaf((float)c.h[j], wk[ky * 3 + 1][j], a);
                    a = fmaf((float)r.h[j], wk[ky * 3 + 2][j], a);
                }
                acc[o] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) part[slice][r][lane] = acc[r];
    __syncthreads();
    for (int r = slice; r < HR; r += 4) {
        const int y = y0 + r;
        if (lane == 0 || lane == 63 || !xin || y >= H) continue;
        const float s = ((part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane])) + bias[0];
        const float d = scale / (1.f + expf(-s));
        disp[(size_t)n * HW + y * W + x] = d;
        if (up_out) {
            const int Wo = 2 * W;
            f16x8 v = {(_Float16)d, 0, 0, 0, 0, 0, 0, 0};
            const size_t o = c4_offset(n, up_Gtot, up_g, 4 * HW, (2 * y) * Wo + 2 * x);
            st8(up_out, o, v); st8(up_out, o + 4, v); st8(up_out, o + (size_t)Wo * 4, v); st8(up_out, o + (size_t)Wo * 4 + 4, v);
        }
    }
}

extern "C" int cnm_head_sigmoid_c8_f16(const void* in, int Gin_total, int gin0, int C,
                                       const float* w_head, const float* bias, float scale,
                                       float* disp, void* up_out, int up_Gtotal, int up_g,
                                       int N, int H, int W, void* stream) {
    CNM_REQUIRE(in && w_head && bias && disp && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(gin0 >= 0 && gin0 + C / 8 <= Gin_total && (!up_out || (up_g >= 0 && up_g < up_Gtotal)), CNM_ERR_BAD_ARG);
    const long long total = (long long)N * H * W;
    if (total >= 8ll * 96 * 128 && W >= 62 && C <= 128) {                  // enough tiles to fill the chip: the row-walking kernel
        constexpr int HR = 8;
        const int tilesX = cnm_ceil_div(W, 62), tilesY = cnm_ceil_div(H, HR);
        head_sigmoid_c8h_rows_kernel<HR><<<(unsigned)(N * tilesX * tilesY), 256, 0, cnm_stream(stream)>>>(
            static_cast<const float*>(in), Gin_total, gin0, C / 8, w_head, bias, scale, disp, static_cast<float*>(up_out), up_Gtotal, up_g, N, H, W, tilesX, tilesY);
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
    if (C >= 256) head_sigmoid_c8h_kernel<16><<<(unsigned)cnm_ceil_div_ll(total, 64), 1024, 0, cnm_stream(stream)>>>(
        static_cast<const float*>(in), Gin_total, gin0, C / 8, w_head, bias, scale, disp, static_cast<float*>(up_out), up_Gtotal, up_g, N, H, W);
    else head_sigmoid_c8h_kernel<4><<<(unsigned)cnm_ceil_div_ll(total, 64), 256, 0, cnm_stream(stream)>>>(
        static_cast<const float*>(in), Gin_total, gin0, C / 8, w_head, bias, scale, disp, static_cast<float*>(up_out), up_Gtotal, up_g, N, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ refine input assembly (S sources, S even; S = 2 is the plain case)
// x = [mean_even(f) + mean_odd(f) (C channels), id_even, id_odd, |id_even - id_odd|, 0 x5]   (depthNet_model.py:332-333, eval.py:656-663)
__global__ __launch_bounds__(256) void refine_assemble_multi_c8h_kernel(const float* __restrict__ idp, const float* __restrict__ f,
                                                                        float* __restrict__ x, int B, int S, int G, int HW) {
    const long long total = (long long)B * (G + 1) * HW;
    const int h = S / 2;
    const float inv = h == 1 ? 1.f : (h == 2 ? 0.5f : 1.f / (float)h);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % (G + 1)), b = (int)(r / (G + 1));
        f16x8 v;
        if (g < G) {
            float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < h; ++k) {
                const f16x8 a = ld8(f, c4_offset(b * S + 2 * k, G, g, HW, pix)), c = ld8(f, c4_offset(b * S + 2 * k + 1, G, g, HW, pix));
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += (float)a[j]; s2[j] += (float)c[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (_Float16)(s1[j] * inv + s2[j] * inv);
        } else {
            float a = 0.f, c = 0.f;
            for (int k = 0; k < h; ++k) { a += idp[(size_t)(b * S + 2 * k) * HW + pix]; c += idp[(size_t)(b * S + 2 * k + 1) * HW + pix]; }
            a *= inv; c *= inv;
            v = (f16x8){(_Float16)a, (_Float16)c, (_Float16)fabsf(a - c), 0, 0, 0, 0, 0};
        }
        st8(x, c4_offset(b, G + 1, g, HW, pix), v);
    }
}

extern "C" int cnm_refine_assemble_multi_c8_f16(const float* idepth_pairs, const void* feat_pairs_c8, void* x,
                                                int B, int S, int C, int H, int W, void* stream) {
    CNM_REQUIRE(idepth_pairs && feat_pairs_c8 && x && B > 0 && S >= 2 && S % 2 == 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const long long total = (long long)B * (C / 8 + 1) * H * W;
    refine_assemble_multi_c8h_kernel<<<(int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384), 256, 0, cnm_stream(stream)>>>(
        idepth_pairs, static_cast<const float*>(feat_pairs_c8), static_cast<float*>(x), B, S, C / 8, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ layout converters NCHW fp32 <-> c8 fp16
__global__ __launch_bounds__(256) void nchw_to_c8h_kernel(const float* __restrict__ src, float* __restrict__ dst, int Gt, int g0, int N, int C, int HW) {
    const int G = (C + 7) / 8;
    const long long total = (long long)N * G * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % G), n = (int)(r / G);
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = 8 * g + j; v[j] = (_Float16)(c < C ? src[((size_t)n * C + c) * HW + pix] : 0.f); }
        st8(dst, c4_offset(n, Gt, g0 + g, HW, pix), v);
    }
}

__global__ __launch_bounds__(256) void c8h_to_nchw_kernel(const float* __restrict__ src, int Gt, int g0, float* __restrict__ dst, int N, int C, int HW) {
    const int G = (C + 7) / 8;
    const long long total = (long long)N * G * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % G), n = (int)(r / G);
        const f16x8 v = ld8(src, c4_offset(n, Gt, g0 + g, HW, pix));
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = 8 * g + j; if (c < C) dst[((size_t)n * C + c) * HW + pix] = (float)v[j]; }
    }
}

extern "C" int cnm_nchw_to_c8_f16(const float* nchw, void* c8, int G_total, int g0, int N, int C, int H, int W, void* stream) {
    CNM_REQUIRE(nchw && c8 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 7) / 8 <= G_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * ((C + 7) / 8) * H * W;
    nchw_to_c8h_kernel<<<(int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384), 256, 0, cnm_stream(stream)>>>(nchw, static_cast<float*>(c8), G_total, g0, N, C, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_c8_to_nchw_f16(const void* c8, int G_total, int g0, float* nchw, int N, int C, int H, int W, void* stream) {
    CNM_REQUIRE(nchw && c8 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 7) / 8 <= G_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * ((C + 7) / 8) * H * W;
    c8h_to_nchw_kernel<<<(int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384), 256, 0, cnm_stream(stream)>>>(static_cast<const float*>(c8), G_total, g0, nchw, N, C, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
