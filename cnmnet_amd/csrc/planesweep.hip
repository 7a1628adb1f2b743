// K0 + K1: camera prep and the fused plane-sweep warp + L1 cost volume.
//
// K0 replaces get_pixel_coordinates + process_camera_parameters
//    (reference depthnet/depth_util.py:13-56): 12 floats per (ref,src) pair; the pixel
//    grid is implicit in the thread index and KRKiUV [B,3,H*W] is never materialised.
// K1 replaces depthNet.getVolume (reference depthnet/depthNet_model.py:185-224), i.e.
//    64 x ~12 ATen launches per pair, with ONE launch over all pairs, planes and pixels.
//
// K1 mapping (gfx950, 64-lane waves): workgroup = 4 waves = 64x4 pixel tile, one lane per
// reference pixel, all D planes walked by that lane; the (u,v,1) homography product and the
// reference RGB stay in registers for the whole sweep.  Planes are taken in groups of 8:
// for a group the workgroup computes the bounding box of the tile's footprint in the source
// image (projective map => extremes at the 4 tile corners x 2 end planes), stages that box
// into LDS as interleaved (r,g,b,0) float4 texels with a 2-texel zero border where it leaves
// the image, and every bilinear tap becomes one ds_read_b128 -- no per-corner bounds tests.
// Double-buffered staging => one barrier per group.  If a footprint does not fit (extreme
// geometry, points behind the source camera) the group falls back to bounds-checked global
// gathers with identical arithmetic.  Output leaves the registers as coalesced stores:
// float4 (4 planes of one pixel) in the c4 layout, or one float per plane for NCHW.
// HBM-bound by design: algorithmic bytes per pair = 3HW*4 (ref) + 3HW*4 (src) + D*HW*4 (volume)
// (+ 4HW*4 for the ref group when emitting the concatenated conv input).
#include "cnm_common.h"

#define CNM_MAX_PLANES 128
#define SWEEP_TW 64
#define SWEEP_TH 4
#define SWEEP_PG 8            // planes per staging group
#define SWEEP_CAP 1024        // texels per LDS staging buffer (16 KB)

struct SweepArgs {
    const float* ref; const float* src; const float* hmkt; float* out;
    int B, S, H, W, D;
    float z[CNM_MAX_PLANES];
};

// ------------------------------------------------------------------ K0
__device__ static bool inv_nxn(double* A, double* Ai, int n) {   // Gauss-Jordan, partial pivoting
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ai[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; ++c) {
        int piv = c; double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[r * n + c]) > best) { best = fabs(A[r * n + c]); piv = r; }
        if (piv != c) for (int j = 0; j < n; ++j) {
            double t = A[c * n + j]; A[c * n + j] = A[piv * n + j]; A[piv * n + j] = t;
            t = Ai[c * n + j]; Ai[c * n + j] = Ai[piv * n + j]; Ai[piv * n + j] = t;
        }
        const double d = 1.0 / A[c * n + c];
        for (int j = 0; j < n; ++j) { A[c * n + j] *= d; Ai[c * n + j] *= d; }
        for (int r = 0; r < n; ++r) if (r != c) {
            const double f = A[r * n + c];
            for (int j = 0; j < n; ++j) { A[r * n + j] -= f * A[c * n + j]; Ai[r * n + j] -= f * Ai[c * n + j]; }
        }
    }
    return true;
}

__global__ void homography_terms_kernel(const float* __restrict__ ref_cam, const float* __restrict__ src_cam,
                                        float* __restrict__ hmkt, int B, int S) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B * S) return;
    const float* lc = ref_cam + (size_t)(p / S) * 32;
    const float* rc = src_cam + (size_t)p * 32;
    double El[16], Eli[16], Kl[9], Kli[9], rel[16];
    for (int i = 0; i < 16; ++i) El[i] = lc[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Kl[i * 3 + j] = lc[16 + i * 4 + j];
    inv_nxn(El, Eli, 4);
    inv_nxn(Kl, Kli, 3);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {     // right2left = E_r @ E_l^-1  (depth_util.py:37)
        double s = 0; for (int k = 0; k < 4; ++k) s += (double)rc[i * 4 + k] * Eli[k * 4 + j];
        rel[i * 4 + j] = s;
    }
    double RKi[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {     // R @ K_l^-1                 (depth_util.py:42)
        double s = 0; for (int k = 0; k < 3; ++k) s += rel[i * 4 + k] * Kli[k * 3 + j];
        RKi[i * 3 + j] = s;
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += (double)rc[16 + i * 4 + k] * RKi[k * 3 + j];
            hmkt[(size_t)p * 12 + i * 3 + j] = (float)s;          // Hm = K_r R K_l^-1
        }
        double s = 0; for (int k = 0; k < 3; ++k) s += (double)rc[16 + i * 4 + k] * rel[k * 4 + 3];
        hmkt[(size_t)p * 12 + 9 + i] = (float)s;                  // KT = K_r T               (depth_util.py:46-50)
    }
}

extern "C" int cnm_homography_terms_f32(const float* ref_cam, const float* src_cam, float* hmkt,
                                        int B, int S, void* stream) {
    CNM_REQUIRE(ref_cam && src_cam && hmkt && B > 0 && S > 0, CNM_ERR_BAD_ARG);
    homography_terms_kernel<<<cnm_ceil_div(B * S, 64), 64, 0, cnm_stream(stream)>>>(ref_cam, src_cam, hmkt, B, S);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_idepth_range_host(double idepth_scale, double* idepth_min, double* idepth_max) {
    CNM_REQUIRE(idepth_min && idepth_max, CNM_ERR_BAD_ARG);
    if (idepth_scale == 2.0) { *idepth_min = 0.02; *idepth_max = 2.0; return CNM_OK; }
    if (idepth_scale == 3.0) { *idepth_min = 0.1; *idepth_max = 3.0; return CNM_OK; }
    return CNM_ERR_BAD_SCALE;
}

// ------------------------------------------------------------------ K1
__device__ static inline float fast_div(float n, float d) {       // v_rcp_f32 + one Newton step
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return n * r;
}

template <int LAYOUT>   // 0: volume [P,D,H,W]   1: c4 [P,D/4+1,H,W,4]
__global__ __launch_bounds__(256) void planesweep_kernel(const SweepArgs a) {
    __shared__ float4 tex[2][SWEEP_CAP];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx0 = blockIdx.x * SWEEP_TW, ty0 = blockIdx.y * SWEEP_TH;
    const int x = tx0 + lane, y = ty0 + wave;
    const int p = blockIdx.z, b = p / a.S;
    const int H = a.H, W = a.W, HW = H * W, D = a.D;
    const bool pvalid = x < W && y < H;

    const float* hk = a.hmkt + (size_t)p * 12;
    const float h00 = hk[0], h01 = hk[1], h02 = hk[2], h10 = hk[3], h11 = hk[4], h12 = hk[5];
    const float h20 = hk[6], h21 = hk[7], h22 = hk[8], k0 = hk[9], k1 = hk[10], k2 = hk[11];
    const float fx_ = (float)x, fy_ = (float)y;
    const float a0 = fmaf(h00, fx_, fmaf(h01, fy_, h02));
    const float a1 = fmaf(h10, fx_, fmaf(h11, fy_, h12));
    const float a2 = fmaf(h20, fx_, fmaf(h21, fy_, h22));

    const float* refp = a.ref + (size_t)b * 3 * HW + (size_t)y * W + x;
    float rr = 0.f, rg = 0.f, rb = 0.f;
    if (pvalid) { rr = refp[0]; rg = refp[HW]; rb = refp[2 * HW]; }
    const float* srcp = a.src + (size_t)p * 3 * HW;

    // tile corners for the footprint box: lanes 0..7 = 4 corners x {first,last plane of group}
    const int cxi = (lane & 1) ? min(tx0 + SWEEP_TW - 1, W - 1) : tx0;
    const int cyi = (lane & 2) ? min(ty0 + SWEEP_TH - 1, H - 1) : ty0;
    const float cxf = (float)cxi, cyf = (float)cyi;
    const float ca0 = fmaf(h00, cxf, fmaf(h01, cyf, h02));
    const float ca1 = fmaf(h10, cxf, fmaf(h11, cyf, h12));
    const float ca2 = fmaf(h20, cxf, fmaf(h21, cyf, h22));

    float cost4[4];
    const int ngroups = (D + SWEEP_PG - 1) / SWEEP_PG;
    for (int g = 0; g < ngroups; ++g) {
        const int d0 = g * SWEEP_PG, d1 = min(d0 + SWEEP_PG, D) - 1;
        // ---- footprint box (computed redundantly by every wave: identical, no exchange needed)
        float umin, umax, vmin, vmax; int okc;
        {
            const float zc = a.z[(lane & 4) ? d1 : d0];
            const float den = fmaf(ca2, zc, k2) + 1e-6f;
            const float u = fast_div(fmaf(ca0, zc, k0), den), v = fast_div(fmaf(ca1, zc, k1), den);
            okc = (den > 1e-4f) && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f);
            umin = umax = u; vmin = vmax = v;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                umin = fminf(umin, __shfl_xor(umin, m, 8)); umax = fmaxf(umax, __shfl_xor(umax, m, 8));
                vmin = fminf(vmin, __shfl_xor(vmin, m, 8)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 8));
                okc &= __shfl_xor(okc, m, 8);
            }
            umin = __shfl(umin, 0); umax = __shfl(umax, 0); vmin = __shfl(vmin, 0); vmax = __shfl(vmax, 0);
            okc = __shfl(okc, 0);
        }
        // sample corners x0 = floor(u-0.5) .. x0+1, with one texel of safety margin either side
        int rx0 = (int)fminf(fmaxf(floorf(umin - 0.5f) - 1.f, -2.f), (float)W);
        int rx1 = (int)fminf(fmaxf(floorf(umax - 0.5f) + 2.f, (float)(rx0 + 1)), (float)(W + 1));
        int ry0 = (int)fminf(fmaxf(floorf(vmin - 0.5f) - 1.f, -2.f), (float)H);
        int ry1 = (int)fminf(fmaxf(floorf(vmax - 0.5f) + 2.f, (float)(ry0 + 1)), (float)(H + 1));
        rx0 = __builtin_amdgcn_readfirstlane(rx0); rx1 = __builtin_amdgcn_readfirstlane(rx1);
        ry0 = __builtin_amdgcn_readfirstlane(ry0); ry1 = __builtin_amdgcn_readfirstlane(ry1);
        const int rw = rx1 - rx0 + 1, rh = ry1 - ry0 + 1;
        const bool staged = __builtin_amdgcn_readfirstlane(okc) && (rw * rh <= SWEEP_CAP);
        float4* tb = tex[g & 1];

        if (staged) {
            for (int r = wave; r < rh; r += 4) {
                const int iy = ry0 + r;
                const bool rowin = (unsigned)iy < (unsigned)H;
                for (int c = lane; c < rw; c += 64) {
                    const int ix = rx0 + c;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rowin && (unsigned)ix < (unsigned)W) {
                        const float* s = srcp + (size_t)iy * W + ix;
                        v.x = s[0]; v.y = s[HW]; v.z = s[2 * HW];
                    }
                    tb[r * rw + c] = v;
                }
            }
        }
        __syncthreads();   // unconditional: also orders buffer reuse when a group was not staged

#pragma unroll
        for (int j = 0; j < SWEEP_PG; ++j) {
            const int dd = d0 + j;
            if (dd > d1) break;
            const float z = a.z[dd];
            const float den = fmaf(a2, z, k2) + 1e-6f;                      // depthNet_model.py:210-212
            const float ix = fast_div(fmaf(a0, z, k0), den) - 0.5f;         // :213 + grid_sample unnormalise
            const float iy = fast_div(fmaf(a1, z, k1), den) - 0.5f;
            const float flx = floorf(ix), fly = floorf(iy);
            const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            float4 p00, p01, p10, p11;
            if (staged) {
                const int xi = (int)fminf(fmaxf(flx, (float)rx0), (float)(rx1 - 1)) - rx0;
                const int yi = (int)fminf(fmaxf(fly, (float)ry0), (float)(ry1 - 1)) - ry0;
                const float4* q = tb + yi * rw + xi;
                p00 = q[0]; p01 = q[1]; p10 = q[rw]; p11 = q[rw + 1];
            } else {
                p00 = p01 = p10 = p11 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fabsf(ix) < 1e7f && fabsf(iy) < 1e7f) {                  // false for NaN/inf as well
                    const int xi = (int)flx, yi = (int)fly;
                    const bool x0in = (unsigned)xi < (unsigned)W, x1in = (unsigned)(xi + 1) < (unsigned)W;
                    const bool y0in = (unsigned)yi < (unsigned)H, y1in = (unsigned)(yi + 1) < (unsigned)H;
                    const float* s = srcp + (ptrdiff_t)yi * W + xi;
                    if (y0in && x0in) { p00.x = s[0]; p00.y = s[HW]; p00.z = s[2 * HW]; }
                    if (y0in && x1in) { p01.x = s[1]; p01.y = s[HW + 1]; p01.z = s[2 * HW + 1]; }
                    if (y1in && x0in) { p10.x = s[W]; p10.y = s[HW + W]; p10.z = s[2 * HW + W]; }
                    if (y1in && x1in) { p11.x = s[W + 1]; p11.y = s[HW + W + 1]; p11.z = s[2 * HW + W + 1]; }
                }
            }
            const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
            const float wr = fmaf(w11, p11.x, fmaf(w10, p10.x, fmaf(w01, p01.x, w00 * p00.x)));
            const float wg = fmaf(w11, p11.y, fmaf(w10, p10.y, fmaf(w01, p01.y, w00 * p00.y)));
            const float wb = fmaf(w11, p11.z, fmaf(w10, p10.z, fmaf(w01, p01.z, w00 * p00.z)));
            const float cost = fabsf(wr - rr) + fabsf(wg - rg) + fabsf(wb - rb);   // :222-223
            if (LAYOUT == 0) {
                if (pvalid) a.out[((size_t)p * D + dd) * HW + (size_t)y * W + x] = cost;
            } else {
                cost4[j & 3] = cost;
                if ((j & 3) == 3 && pvalid)
                    *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, dd >> 2, HW, y * W + x)) =
                        make_float4(cost4[0], cost4[1], cost4[2], cost4[3]);
            }
        }
    }
    if (LAYOUT == 1 && pvalid)
        *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, D / 4, HW, y * W + x)) = make_float4(rr, rg, rb, 0.f);
}

static int sweep_launch(int layout, const float* ref, const float* src, const float* hmkt, float* out,
                        int B, int S, int H, int W, int D, double idepth_min, double idepth_max, void* stream) {
    CNM_REQUIRE(ref && src && hmkt && out, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= CNM_MAX_PLANES, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout == 0 || D % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)B * S <= 65535, CNM_ERR_BAD_ARG);
    SweepArgs a;
    a.ref = ref; a.src = src; a.hmkt = hmkt; a.out = out;
    a.B = B; a.S = S; a.H = H; a.W = W; a.D = D;
    const double step = (idepth_max - idepth_min) / (D - 1.0);              // depthNet_model.py:194
    for (int d = 0; d < CNM_MAX_PLANES; ++d)
        a.z[d] = d < D ? (float)(1.0 / (idepth_min + d * step)) : 0.f;      // :209 (python double -> fp32)
    dim3 grid(cnm_ceil_div(W, SWEEP_TW), cnm_ceil_div(H, SWEEP_TH), B * S);
    if (layout == 0) planesweep_kernel<0><<<grid, 256, 0, cnm_stream(stream)>>>(a);
    else planesweep_kernel<1><<<grid, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_planesweep_volume_nchw_f32(const float* ref, const float* src, const float* hmkt, float* volume,
                                              int B, int S, int H, int W, int D,
                                              double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(0, ref, src, hmkt, volume, B, S, H, W, D, idepth_min, idepth_max, stream);
}

extern "C" int cnm_planesweep_cat_c4_f32(const float* ref, const float* src, const float* hmkt, float* x,
                                         int B, int S, int H, int W, int D,
                                         double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(1, ref, src, hmkt, x, B, S, H, W, D, idepth_min, idepth_max, stream);
}
