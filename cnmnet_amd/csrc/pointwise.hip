// K3-K5 and boundary layout kernels (all HBM-bound, one float4 per lane, coalesced).
//   upsample2x   : nn.Upsample(scale_factor=2, bilinear, align_corners=False)  depthNet_model.py:94,105
//   head         : depth_layer conv3x3 C->1 + bias + sigmoid, * idepth_scale    :82-84,246,251,256,261,351,365
//                  + F.upsample(disp, 2) nearest written into the consumer's concat slot (:247,252,257)
//   refine input : cat(id1, id2, |id1-id2|, iconv01+iconv02)                    :332-333
//   nchw <-> c4  : module-boundary converters
#include "cnm_common.h"

// ------------------------------------------------------------------ bilinear x2 (c4)
// src = (dst+0.5)/2 - 0.5 clamped at 0; i0 = floor, i1 = min(i0+1, n-1), lambda = src - i0
// (torch upsample_bilinear2d, align_corners=False).
// One thread per LOW-resolution (pixel, channel group): its 2 x 2 output pixels come from the 3 x 3 neighbourhood (9 loads
// per 4 outputs instead of 16, one 32-bit index decomposition instead of four 64-bit ones).  Per output the expression of
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False): src = (dst + 0.5) / 2 - 0.5 clamped at 0, i.e. weights
// 0.75 / 0.25; output row 2y + a reads neighbourhood rows (a, a + 1); at y = 0 the weight of the second row is 0 (the
// reference reads row 1 there, this kernel row 0 again: the same value for finite data).  Columns alike.
__global__ __launch_bounds__(256) void upsample2x_c4_kernel(const float* __restrict__ in, int Gin_tot, int gin0,
                                                            float* __restrict__ out, int Gout_tot, int gout0,
                                                            int N, int G, int H, int W) {
    const int Wo = 2 * W, Ho = 2 * H;
    const unsigned total = (unsigned)N * G * H * W;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int x = (int)(idx % (unsigned)W);
        unsigned r = idx / (unsigned)W;
        const int y = (int)(r % (unsigned)H); r /= (unsigned)H;
        const int g = (int)(r % (unsigned)G), n = (int)(r / (unsigned)G);
        const int ys[3] = {max(y - 1, 0), y, min(y + 1, H - 1)}, xs[3] = {max(x - 1, 0), x, min(x + 1, W - 1)};
        const float4* base = reinterpret_cast<const float4*>(in + c4_offset(n, Gin_tot, gin0 + g, H * W, 0));
        float4 p[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) p[i][j] = base[ys[i] * W + xs[j]];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float ly = a ? 0.25f : (y > 0 ? 0.75f : 0.f), lx = b ? 0.25f : (x > 0 ? 0.75f : 0.f), hy = 1.f - ly, hx = 1.f - lx;
                const float4 p00 = p[a][b], p01 = p[a][b + 1], p10 = p[a + 1][b], p11 = p[a + 1][b + 1];
                float4 v;
                v.x = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
                v.y = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
                v.z = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
                v.w = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
                *reinterpret_cast<float4*>(out + c4_offset(n, Gout_tot, gout0 + g, Ho * Wo, (2 * y + a) * Wo + 2 * x + b)) = v;
            }
    }
}

extern "C" int cnm_upsample2x_c4_f32(const float* in, int Gin_total, int gin0,
                                     float* out, int Gout_total, int gout0,
                                     int N, int G, int H, int W, void* stream) {
    CNM_REQUIRE(in && out && N > 0 && G > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(gin0 >= 0 && gin0 + G <= Gin_total && gout0 >= 0 && gout0 + G <= Gout_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * G * H * W;                      // one thread per low-resolution (pixel, group)
    CNM_REQUIRE(total < (1ll << 31), CNM_ERR_BAD_ARG);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    upsample2x_c4_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(in, Gin_total, gin0, out, Gout_total, gout0, N, G, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ disparity head
__global__ void pack_head_kernel(const float* __restrict__ w, int C, float* __restrict__ wh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // wh[tap][c] <- w[0][c][tap]
    if (i >= 9 * C) return;
    const int tap = i / C, c = i - tap * C;
    wh[i] = w[c * 9 + tap];
}

extern "C" int cnm_pack_head_f32(const float* w_oihw, int C, float* w_head, void* stream) {
    CNM_REQUIRE(w_oihw && w_head && C > 0 && C % 4 == 0, CNM_ERR_BAD_ARG);
    pack_head_kernel<<<cnm_ceil_div(9 * C, 256), 256, 0, cnm_stream(stream)>>>(w_oihw, C, w_head);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// Workgroup = 64 consecutive output pixels x 4 channel slices (one wave per slice): consecutive lanes =
// consecutive x, so every float4 tap load is coalesced across the wave; the 9*C weights of a slice are
// wave-uniform (scalar loads); the four partial sums meet in LDS in a fixed order.  (One lane per pixel over all
// channels left the low-resolution heads with a few dozen workgroups and a 4608-tap serial loop.)
template <int NS>                                                        // channel slices = waves per workgroup (4, or 16 for the 256- / 512-channel heads)
__global__ __launch_bounds__(64 * NS) void head_sigmoid_c4_kernel(const float* __restrict__ in, int Gin_tot, int gin0, int G,
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
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int g = gbeg; g < gend; ++g) {
        const float4* base = reinterpret_cast<const float4*>(in + c4_offset(n, Gin_tot, gin0 + g, HW, 0));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + kx - 1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = base[iy * W + ix];
                const float4 w = *reinterpret_cast<const float4*>(wh + (size_t)(ky * 3 + kx) * (G * 4) + g * 4);
                acc0 = fmaf(v.x, w.x, acc0); acc1 = fmaf(v.y, w.y, acc1);
                acc2 = fmaf(v.z, w.z, acc2); acc3 = fmaf(v.w, w.w, acc3);
            }
        }
    }
    part[slice][lane] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (slice != 0 || !live) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NS; q += 4) s += (part[q][lane] + part[q + 1][lane]) + (part[q + 2][lane] + part[q + 3][lane]);   // fixed order
    s += bias[0];
    const float d = scale / (1.f + expf(-s));
    disp[idx] = d;
    if (up_out) {
        const int Wo = 2 * W;
        float* o = up_out + c4_offset(n, up_Gtot, up_g, 4 * HW, (2 * y) * Wo + 2 * x);
        const float4 v = make_float4(d, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(o) = v; *reinterpret_cast<float4*>(o + 4) = v;
        *reinterpret_cast<float4*>(o + (size_t)Wo * 4) = v; *reinterpret_cast<float4*>(o + (size_t)Wo * 4 + 4) = v;
    }
}

// High-resolution heads: the kernel above reads every input texel nine times through L1 (three rows x three columns
// per output), which is what bounds it at 192x256.  Here a lane owns one image column of a 62 x HR pixel tile and walks
// down the rows: each texel is loaded ONCE per tile (float4, coalesced), its left / right neighbours come from the
// adjacent lanes (wave shift), and a loaded row feeds the three output rows it touches from registers.  Lanes 0 and 63
// only supply halo columns (62 outputs per 64 lanes).  Waves = 4 channel slices, partial sums meet in LDS.
// whole-wave shifts by one lane on the VALU (DPP wave_shr / wave_shl, gfx9): lane i receives lane i-1 / i+1
__device__ __forceinline__ float lane_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false)); }
__device__ __forceinline__ float lane_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false)); }

template <int HR>
__global__ __launch_bounds__(256) void head_sigmoid_c4_rows_kernel(const float* __restrict__ in, int Gin_tot, int gin0, int G,
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
        const float4* base = reinterpret_cast<const float4*>(in + c4_offset(n, Gin_tot, gin0 + g, HW, 0));
        float4 row[HR + 2];
#pragma unroll
        for (int i = 0; i < HR + 2; ++i) {                               // input rows y0-1 .. y0+HR: all loads first
            const int iy = y0 + i - 1;
            row[i] = (xin && (unsigned)iy < (unsigned)H) ? base[iy * W + x] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(wh + (size_t)k * (G * 4) + g * 4);   // wave-uniform
#pragma unroll
        for (int i = 0; i < HR + 2; ++i) {
            const float4 c = row[i];
            float4 l, r;                                                 // columns x-1 / x+1 from lanes-1 / +1 (lanes 0 / 63 get garbage: halo lanes, no output)
            l.x = lane_shr1(c.x); l.y = lane_shr1(c.y); l.z = lane_shr1(c.z); l.w = lane_shr1(c.w);
            r.x = lane_shl1(c.x); r.y = lane_shl1(c.y); r.z = lane_shl1(c.z); r.w = lane_shl1(c.w);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int o = i - ky;                                    // output row of the tile this input row feeds through kernel row ky
                if (o < 0 || o >= HR) continue;
                const float4 w0 = wk[ky * 3 + 0], w1 = wk[ky * 3 + 1], w2 = wk[ky * 3 + 2];
                float a = acc[o];
                a = fmaf(l.x, w0.x, a); a = fmaf(l.y, w0.y, a); a = fmaf(l.z, w0.z, a); a = fmaf(l.w, w0.w, a);
                a = fmaf(c.x, w1.x, a); a = fmaf(c.y, w1.y, a); a = fmaf(c.z, w1.z, a); a = fmaf(c.w, w1.w, a);
                a = fmaf(r.x, w2.x, a); a = fmaf(r.y, w2.y, a); a = fmaf(r.z, w2.z, a); a = fmaf(r.w, w2.w, a);
                acc[o] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) part[slice][r][lane] = acc[r];
    __syncthreads();
    // 256 threads finish HR x 64 sums: thread t -> (row t >> 6 + 4 k, lane)
    for (int r = slice; r < HR; r += 4) {
        const int y = y0 + r;
        if (lane == 0 || lane == 63 || !xin || y >= H) continue;
        const float s = ((part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane])) + bias[0];
        const float d = scale / (1.f + expf(-s));
        disp[(size_t)n * HW + y * W + x] = d;
        if (up_out) {
            const int Wo = 2 * W;
            float* o = up_out + c4_offset(n, up_Gtot, up_g, 4 * HW, (2 * y) * Wo + 2 * x);
            const float4 v = make_float4(d, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(o) = v; *reinterpret_cast<float4*>(o + 4) = v;
            *reinterpret_cast<float4*>(o + (size_t)Wo * 4) = v; *reinterpret_cast<float4*>(o + (size_t)Wo * 4 + 4) = v;
        }
    }
}

extern "C" int cnm_head_sigmoid_c4_f32(const float* in, int Gin_total, int gin0, int C,
                                       const float* w_head, const float* bias, float scale,
                                       float* disp, float* up_out, int up_Gtotal, int up_g,
                                       int N, int H, int W, void* stream) {
    CNM_REQUIRE(in && w_head && bias && disp && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(gin0 >= 0 && gin0 + C / 4 <= Gin_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!up_out || (up_g >= 0 && up_g < up_Gtotal), CNM_ERR_BAD_ARG);
    const long long total = (long long)N * H * W;
    if (total >= 8ll * 96 * 128 && W >= 62) {                             // enough tiles to fill the chip: the row-walking kernel
        constexpr int HR = 8;
        const int tilesX = cnm_ceil_div(W, 62), tilesY = cnm_ceil_div(H, HR);
        head_sigmoid_c4_rows_kernel<HR><<<(unsigned)(N * tilesX * tilesY), 256, 0, cnm_stream(stream)>>>(
            in, Gin_total, gin0, C / 4, w_head, bias, scale, disp, up_out, up_Gtotal, up_g, N, H, W, tilesX, tilesY);
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
    if (C >= 256)                                                        // deep low-resolution heads: 16 waves share the 9*C-tap reduction of 64 pixels
        head_sigmoid_c4_kernel<16><<<(unsigned)cnm_ceil_div_ll(total, 64), 1024, 0, cnm_stream(stream)>>>(
            in, Gin_total, gin0, C / 4, w_head, bias, scale, disp, up_out, up_Gtotal, up_g, N, H, W);
    else head_sigmoid_c4_kernel<4><<<(unsigned)cnm_ceil_div_ll(total, 64), 256, 0, cnm_stream(stream)>>>(
        in, Gin_total, gin0, C / 4, w_head, bias, scale, disp, up_out, up_Gtotal, up_g, N, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ refine input assembly
__global__ __launch_bounds__(256) void refine_assemble_c4_kernel(const float* __restrict__ id1, const float* __restrict__ id2, long long ids,
                                                                 const float* __restrict__ f1, int G1t, int g1,
                                                                 const float* __restrict__ f2, int G2t, int g2,
                                                                 float* __restrict__ x, int N, int G, int HW) {
    const long long total = (long long)N * (G + 1) * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % (G + 1)), n = (int)(r / (G + 1));
        float4 v;
        if (g < G) {
            const float4 a = *reinterpret_cast<const float4*>(f1 + c4_offset(n, G1t, g1 + g, HW, pix));
            const float4 b = *reinterpret_cast<const float4*>(f2 + c4_offset(n, G2t, g2 + g, HW, pix));
            v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        } else {
            const float a = id1[(size_t)n * ids + pix], b = id2[(size_t)n * ids + pix];
            v = make_float4(a, b, fabsf(a - b), 0.f);
        }
        *reinterpret_cast<float4*>(x + c4_offset(n, G + 1, g, HW, pix)) = v;
    }
}

extern "C" int cnm_refine_assemble_c4_f32(const float* idepth01, const float* idepth02, long long idepth_stride,
                                          const float* f1, int G1_total, int g1,
                                          const float* f2, int G2_total, int g2,
                                          float* x, int N, int C, int H, int W, void* stream) {
    CNM_REQUIRE(idepth01 && idepth02 && f1 && f2 && x && N > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(g1 >= 0 && g1 + C / 4 <= G1_total && g2 >= 0 && g2 + C / 4 <= G2_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(idepth_stride >= (long long)H * W, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * (C / 4 + 1) * H * W;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    refine_assemble_c4_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(idepth01, idepth02, idepth_stride, f1, G1_total, g1, f2, G2_total, g2,
                                                                     x, N, C / 4, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// Multi-source variant (reference eval.py:656-663 for S=4, :917-929 for S=6): the S pairs of frame b
// (p = b*S + s) are averaged into the two refine sides -- even sources -> side 1, odd -> side 2 --
// as (a + c) * 0.5 resp. (a + c + e) / 3, then assembled as above.  S = 2 reduces to the plain case.
__global__ __launch_bounds__(256) void refine_assemble_multi_c4_kernel(const float* __restrict__ idp, const float* __restrict__ f,
                                                                       float* __restrict__ x, int B, int S, int G, int HW) {
    const long long total = (long long)B * (G + 1) * HW;
    const int h = S / 2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % (G + 1)), b = (int)(r / (G + 1));
        float4 v;
        if (g < G) {
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
            for (int k = 0; k < h; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(f + c4_offset(b * S + 2 * k, G, g, HW, pix));
                const float4 c = *reinterpret_cast<const float4*>(f + c4_offset(b * S + 2 * k + 1, G, g, HW, pix));
                s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w; s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            }
            if (h == 2) { s1.x *= 0.5f; s1.y *= 0.5f; s1.z *= 0.5f; s1.w *= 0.5f; s2.x *= 0.5f; s2.y *= 0.5f; s2.z *= 0.5f; s2.w *= 0.5f; }
            else if (h > 2) { const float d = (float)h; s1.x /= d; s1.y /= d; s1.z /= d; s1.w /= d; s2.x /= d; s2.y /= d; s2.z /= d; s2.w /= d; }
            v = make_float4(s1.x + s2.x, s1.y + s2.y, s1.z + s2.z, s1.w + s2.w);
        } else {
            float a = 0.f, c = 0.f;
            for (int k = 0; k < h; ++k) { a += idp[(size_t)(b * S + 2 * k) * HW + pix]; c += idp[(size_t)(b * S + 2 * k + 1) * HW + pix]; }
            if (h == 2) { a *= 0.5f; c *= 0.5f; } else if (h > 2) { a /= (float)h; c /= (float)h; }
            v = make_float4(a, c, fabsf(a - c), 0.f);
        }
        *reinterpret_cast<float4*>(x + c4_offset(b, G + 1, g, HW, pix)) = v;
    }
}

extern "C" int cnm_refine_assemble_multi_c4_f32(const float* idepth_pairs, const float* feat_pairs_c4, float* x,
                                                int B, int S, int C, int H, int W, void* stream) {
    CNM_REQUIRE(idepth_pairs && feat_pairs_c4 && x && B > 0 && S >= 2 && S % 2 == 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const long long total = (long long)B * (C / 4 + 1) * H * W;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    refine_assemble_multi_c4_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(idepth_pairs, feat_pairs_c4, x, B, S, C / 4, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ layout converters
__global__ __launch_bounds__(256) void nchw_to_c4_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int Gt, int g0, int N, int C, int HW) {
    const int G = (C + 3) / 4;
    const long long total = (long long)N * G * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % G), n = (int)(r / G);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * g + j;
            v[j] = c < C ? src[((size_t)n * C + c) * HW + pix] : 0.f;
        }
        *reinterpret_cast<float4*>(dst + c4_offset(n, Gt, g0 + g, HW, pix)) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void c4_to_nchw_kernel(const float* __restrict__ src, int Gt, int g0,
                                                         float* __restrict__ dst, int N, int C, int HW) {
    const int G = (C + 3) / 4;
    const long long total = (long long)N * G * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const long long r = idx / HW;
        const int g = (int)(r % G), n = (int)(r / G);
        const float4 v = *reinterpret_cast<const float4*>(src + c4_offset(n, Gt, g0 + g, HW, pix));
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * g + j;
            if (c < C) dst[((size_t)n * C + c) * HW + pix] = vv[j];
        }
    }
}

extern "C" int cnm_nchw_to_c4_f32(const float* nchw, float* c4, int G_total, int g0, int N, int C, int H, int W, void* stream) {
    CNM_REQUIRE(nchw && c4 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 3) / 4 <= G_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * ((C + 3) / 4) * H * W;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    nchw_to_c4_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(nchw, c4, G_total, g0, N, C, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_c4_to_nchw_f32(const float* c4, int G_total, int g0, float* nchw, int N, int C, int H, int W, void* stream) {
    CNM_REQUIRE(nchw && c4 && N > 0 && C > 0 && H > 0 && W > 0 && g0 >= 0 && g0 + (C + 3) / 4 <= G_total, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * ((C + 3) / 4) * H * W;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    c4_to_nchw_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(c4, G_total, g0, nchw, N, C, H * W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
