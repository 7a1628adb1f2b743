// K0 + K1: camera prep and the fused plane-sweep warp + L1 cost volume.
//
// K0 replaces get_pixel_coordinates + process_camera_parameters
//    (reference depthnet/depth_util.py:13-56): 12 floats per (ref,src) pair; the pixel
//    grid is implicit in the thread index and KRKiUV [B,3,H*W] is never materialised.
// K1 replaces depthNet.getVolume (reference depthnet/depthNet_model.py:185-224), i.e.
//    64 x ~12 ATen launches per pair, with ONE launch over all pairs, planes and pixels.
//
// K1 = ONE launch on the caller's stream (gfx950, 64-lane waves):
//   workgroup = 8 waves = 64 x 8 pixel tile, one lane per reference pixel, all D planes walked by that lane;
//   the (u,v,1) homography product and the reference RGB stay in registers for the whole sweep.
//   The footprint of the tile in the source image over a run of planes (projective map => extremes at the
//   4 tile corners x the 2 end planes of every 8-plane octet) is staged ONCE into LDS as pre-differenced
//   texels: 12 floats (P, dP/dx, dP/dy, d2P/dxdy per channel, zero outside the image) = three 16-byte
//   units, built from the planar source by bounds-checked buffer loads (no texture pre-pass, no workspace).
//   A bilinear sample is then P + wu*dx + wv*dy + wu*wv*dxy: three ds_read_b128 of ONE texel and 9 FMAs
//   for the three channels instead of four taps and 12 weighted products, and every blend instruction is a
//   full-rate scalar v_fma_f32 (packed fp32 issues at half rate on gfx950).  The run of planes per box is
//   chosen per workgroup: the largest power-of-two number of octets whose boxes all fit the LDS budget
//   (all 64 planes at the benchmark geometry), so there is one barrier pair per box instead of one per
//   16 planes.  Octets whose footprint does not fit (extreme geometry, points behind the source camera)
//   gather the same texels from global memory with the same arithmetic.  Output leaves the registers as
//   coalesced stores: float4 (4 planes of one pixel) in the c4 layout, one float per plane for NCHW.
// HBM-bound by design: algorithmic bytes per pair = 3HW*4 (ref) + 3HW*4 (src) + D*HW*4 (volume)
// (+ 4HW*4 for the ref group when emitting the concatenated conv input).
#include "cnm_common.h"

#define CNM_MAX_PLANES 128
#define SWEEP_TW 64                 // tile width  (pixels, lanes along x)
#define SWEEP_TH 8                  // tile height
#define SWEEP_NT (SWEEP_TW * SWEEP_TH)
#ifndef SWEEP_CAP
#define SWEEP_CAP 1672              // texels per LDS box (3 x 16 B each: 80 256 B, two workgroups per CU)
#endif
#ifndef SWEEP_MINW
#define SWEEP_MINW 4                // waves per SIMD the register allocation must allow (2 workgroups x 8 waves per CU)
#endif
#ifndef SWEEP_AHEAD
#define SWEEP_AHEAD 2               // samples whose texel reads are in flight ahead of the blend
#endif
#define SWEEP_MAX_OCT (CNM_MAX_PLANES / 8)

struct SweepArgs {
    const float* ref; const float* src; const float* hmkt; float* out;
    int B, S, H, W, D;
    double idmin, idstep;           // plane d lies at depth 1 / (idmin + d * idstep)
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
typedef _Float16 sw_f16x8 __attribute__((ext_vector_type(8)));

// depth of plane d exactly as depthNet_model.py:193-194,209: python doubles, then one rounding to fp32
// (separately rounded multiply / add / divide: no contraction)
__device__ __forceinline__ float sweep_depth(const SweepArgs& a, int d) {
#pragma clang fp contract(off)
    const double m = (double)d * a.idstep;
    const double s = a.idmin + m;
    return (float)(1.0 / s);
}

// One pre-differenced texel of the zero-extended source image at (x, y):
//   u0 = (Pr, Pg, Pb, dxr)  u1 = (dxg, dxb, dyr, dyg)  u2 = (dyb, dxyr, dxyg, dxyb)
//   dx = P(x+1,y) - P(x,y), dy = P(x,y+1) - P(x,y), dxy = (P(x+1,y+1) - P(x,y+1)) - dx; P = 0 outside the image,
// so grid_sample's per-corner zeros padding (depthNet_model.py:220) is carried by the data.
struct SweepTexel { float4 u0, u1, u2; };

__device__ __forceinline__ SweepTexel sweep_texel(__amdgpu_buffer_rsrc_t rsrc, int x, int y, int W, int H, unsigned chan_bytes) {
    const bool x0 = (unsigned)x < (unsigned)W, x1 = (unsigned)(x + 1) < (unsigned)W;
    const bool y0 = (unsigned)y < (unsigned)H, y1 = (unsigned)(y + 1) < (unsigned)H;
    const unsigned o = (unsigned)(y * W + x) * 4u, row = (unsigned)W * 4u;
    const unsigned a00 = (x0 && y0) ? o : 0xFFFFFFFFu, a01 = (x1 && y0) ? o + 4u : 0xFFFFFFFFu;      // out of range: the
    const unsigned a10 = (x0 && y1) ? o + row : 0xFFFFFFFFu, a11 = (x1 && y1) ? o + row + 4u : 0xFFFFFFFFu;   // load returns 0
    float p00[3], dx[3], dy[3], dxy[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned so = c * chan_bytes;
        const float v00 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a00, so, 0));
        const float v01 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a01, so, 0));
        const float v10 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a10, so, 0));
        const float v11 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a11, so, 0));
        p00[c] = v00; dx[c] = v01 - v00; dy[c] = v10 - v00; dxy[c] = (v11 - v10) - dx[c];
    }
    SweepTexel t;
    t.u0 = make_float4(p00[0], p00[1], p00[2], dx[0]);
    t.u1 = make_float4(dx[1], dx[2], dy[0], dy[1]);
    t.u2 = make_float4(dy[2], dxy[0], dxy[1], dxy[2]);
    return t;
}

// box: rx0, ry0 = image coordinates of box texel (0,0); rw x rh texels.  The box is the tile's footprint clipped to
// [-2, W] x [-2, H]; columns -2 and W (rows -2 and H) hold all-zero texels, so clamping a sample's coordinates
// into the box reproduces "no contribution" for everything outside the image.
struct SweepBox { int rx0, ry0, rw, rh; };
#ifdef SWEEP_STATS
__device__ unsigned int sweep_stats[4];     // debug builds only: workgroups, staged boxes, octets gathered from global, box texels
#endif

// A sample = coordinates (sweep_coords), three 16-byte reads of one texel, blend (sweep_blend).  The sweep loop
// issues the reads SWEEP_AHEAD samples before the blend that consumes them.
struct SweepCoord { unsigned xi, yi; float wu, wv; };

__device__ __forceinline__ SweepCoord sweep_coords(float cu, float cv, float umax, float vmax, float a0, float a1, float a2,
                                                   float k0, float k1, float k2e, float z) {
    const float den = fmaf(a2, z, k2e);                                      // depthNet_model.py:210-212 (k2e = k2 + 1e-6)
    float r = __builtin_amdgcn_rcpf(den);
    r = fmaf(fmaf(-den, r, 1.0f), r, r);                                     // one Newton step: ~0.5 ulp reciprocal
    float iu = fmaf(fmaf(a0, z, k0), r, cu);                                 // :213 + grid_sample unnormalise (u' - 0.5),
    float iv = fmaf(fmaf(a1, z, k1), r, cv);                                 // relative to the box origin
    iu = __builtin_amdgcn_fmed3f(iu, 0.f, umax);
    iv = __builtin_amdgcn_fmed3f(iv, 0.f, vmax);
    SweepCoord c;
    c.xi = (unsigned)iu; c.yi = (unsigned)iv;                                // floor (coordinates are >= 0 here)
    c.wu = __builtin_amdgcn_fractf(iu); c.wv = __builtin_amdgcn_fractf(iv);
    return c;
}

__device__ __forceinline__ float sweep_blend(const float4 u0, const float4 u1, const float4 u2, float wu, float wv,
                                             float rr, float rg, float rb) {
    const float wuv = wu * wv;
    const float er = fmaf(wuv, u2.y, fmaf(wv, u1.z, fmaf(wu, u0.w, u0.x))) - rr;
    const float eg = fmaf(wuv, u2.z, fmaf(wv, u1.w, fmaf(wu, u1.x, u0.y))) - rg;
    const float eb = fmaf(wuv, u2.w, fmaf(wv, u2.x, fmaf(wu, u1.y, u0.z))) - rb;
    return (__builtin_fabsf(er) + __builtin_fabsf(eg)) + __builtin_fabsf(eb);   // :222-223
}

template <int LAYOUT>   // 0: volume [P,D,H,W] fp32   1: c4 [P,D/4+1,H,W,4] fp32   2: c8 [P,D/8+1,H,W,8] fp16
__global__ __launch_bounds__(SWEEP_NT) __attribute__((amdgpu_waves_per_eu(SWEEP_MINW, SWEEP_MINW))) void planesweep_kernel(const SweepArgs a) {
    // one LDS object: texel box | plane depths | plane groups | header
    __shared__ float4 smem[3 * SWEEP_CAP + CNM_MAX_PLANES / 4 + 2 * SWEEP_MAX_OCT + 1];
    float4* const box = smem;
    float* const zsh = reinterpret_cast<float*>(smem + 3 * SWEEP_CAP);
    int (*const grp)[8] = reinterpret_cast<int (*)[8]>(smem + 3 * SWEEP_CAP + CNM_MAX_PLANES / 4);
    int* const hdr = reinterpret_cast<int*>(smem + 3 * SWEEP_CAP + CNM_MAX_PLANES / 4 + 2 * SWEEP_MAX_OCT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int tx0 = blockIdx.x * SWEEP_TW, ty0 = blockIdx.y * SWEEP_TH;
    const int x = tx0 + lane, y = ty0 + (tid >> 6);
    const int p = blockIdx.z, b = p / a.S;
    const int H = a.H, W = a.W, HW = H * W, D = a.D;
    const bool pvalid = x < W && y < H;
    const int noct = (D + 7) >> 3;
    if (tid < CNM_MAX_PLANES) zsh[tid] = tid < D ? sweep_depth(a, tid) : 0.f;

    const float* hk = a.hmkt + (size_t)p * 12;
    const float h00 = hk[0], h01 = hk[1], h02 = hk[2], h10 = hk[3], h11 = hk[4], h12 = hk[5];
    const float h20 = hk[6], h21 = hk[7], h22 = hk[8], k0 = hk[9], k1 = hk[10], k2 = hk[11];
    const float k2e = k2 + 1e-6f;
    const float fx_ = (float)x, fy_ = (float)y;
    const float a0 = fmaf(h00, fx_, fmaf(h01, fy_, h02)), a1 = fmaf(h10, fx_, fmaf(h11, fy_, h12));
    const float a2 = fmaf(h20, fx_, fmaf(h21, fy_, h22));

    float rr = 0.f, rg = 0.f, rb = 0.f;
    if (pvalid) {
        const float* refp = a.ref + (size_t)b * 3 * HW + (size_t)y * W + x;
        rr = refp[0]; rg = refp[HW]; rb = refp[2 * HW];
    }
    const unsigned chan_bytes = (unsigned)HW * 4u;
    const unsigned long long srcb = reinterpret_cast<unsigned long long>(a.src + (size_t)p * 3 * HW);
    const unsigned src_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)srcb);       // descriptor pinned to SGPRs
    const unsigned src_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(srcb >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>((unsigned long long)src_lo | ((unsigned long long)src_hi << 32)), 0, 3 * chan_bytes, 0x00020000);

    // ---- footprints, wave 0: lane (8j + c) projects tile corner (c & 3) on the first (c < 4) / last plane of
    // octet j; an 8-lane min/max gives the octet's box.  Lane o then owns octet o, and four xor-merges give the
    // boxes of every aligned run of 2, 4, 8, 16 octets.  The longest run whose boxes all fit the LDS budget wins.
    if (tid < 64) {
        const int cxi = (lane & 1) ? min(tx0 + SWEEP_TW - 1, W - 1) : tx0;
        const int cyi = (lane & 2) ? min(ty0 + SWEEP_TH - 1, H - 1) : ty0;
        const float cxf = (float)cxi, cyf = (float)cyi;
        const float ca0 = fmaf(h00, cxf, fmaf(h01, cyf, h02));
        const float ca1 = fmaf(h10, cxf, fmaf(h11, cyf, h12));
        const float ca2 = fmaf(h20, cxf, fmaf(h21, cyf, h22));
        int ox0 = 0, oy0 = 0, ox1 = 0, oy1 = 0, ook = 0;                     // lane o (< 16): footprint of octet o
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int o = min(pass * 8 + (lane >> 3), noct - 1);
            const int d0 = o * 8, d1 = min(d0 + 8, D) - 1;
            const float zc = sweep_depth(a, (lane & 4) ? d1 : d0);
            const float den = fmaf(ca2, zc, k2) + 1e-6f;
            float rc = __builtin_amdgcn_rcpf(den);
            rc = fmaf(fmaf(-den, rc, 1.0f), rc, rc);
            const float u = fmaf(ca0, zc, k0) * rc, v = fmaf(ca1, zc, k1) * rc;
            int okc = (den > 1e-4f) && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f);
            float umin = u, umax = u, vmin = v, vmax = v;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                umin = fminf(umin, __shfl_xor(umin, m, 8)); umax = fmaxf(umax, __shfl_xor(umax, m, 8));
                vmin = fminf(vmin, __shfl_xor(vmin, m, 8)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 8));
                okc &= __shfl_xor(okc, m, 8);
            }
            // texel indices floor(u - 0.5) of the samples, one texel of safety margin either side, clipped to
            // [-2, W] x [-2, H] (the outermost column / row of that range is all zeros)
            const int bx0 = (int)fminf(fmaxf(floorf(umin - 0.5f) - 1.f, -2.f), (float)W);
            const int bx1 = (int)fminf(fmaxf(floorf(umax - 0.5f) + 1.f, (float)bx0), (float)W);
            const int by0 = (int)fminf(fmaxf(floorf(vmin - 0.5f) - 1.f, -2.f), (float)H);
            const int by1 = (int)fminf(fmaxf(floorf(vmax - 0.5f) + 1.f, (float)by0), (float)H);
            const int srcl = 8 * (lane & 7);                                 // lanes 8j .. 8j+7 hold octet (8 pass + j)
            const int sx0 = __shfl(bx0, srcl), sy0 = __shfl(by0, srcl), sx1 = __shfl(bx1, srcl), sy1 = __shfl(by1, srcl);
            const int sok = __shfl(okc, srcl);
            if ((lane >> 3) == pass) { ox0 = sx0; oy0 = sy0; ox1 = sx1; oy1 = sy1; ook = sok; }
        }
        const bool live = lane < noct;
        if (!live) { ox0 = 1 << 28; oy0 = 1 << 28; ox1 = -(1 << 28); oy1 = -(1 << 28); ook = 1; }   // neutral for the merges
        int lx0[5], ly0[5], lx1[5], ly1[5], lok[5];
        lx0[0] = ox0; ly0[0] = oy0; lx1[0] = ox1; ly1[0] = oy1; lok[0] = ook;
#pragma unroll
        for (int L = 1; L < 5; ++L) {
            const int m = 1 << (L - 1);
            lx0[L] = min(lx0[L - 1], __shfl_xor(lx0[L - 1], m)); ly0[L] = min(ly0[L - 1], __shfl_xor(ly0[L - 1], m));
            lx1[L] = max(lx1[L - 1], __shfl_xor(lx1[L - 1], m)); ly1[L] = max(ly1[L - 1], __shfl_xor(ly1[L - 1], m));
            lok[L] = lok[L - 1] & __shfl_xor(lok[L - 1], m);
        }
        int level = 0;
        int gx0 = lx0[0], gy0 = ly0[0], gx1 = lx1[0], gy1 = ly1[0], gst = 0;
#pragma unroll
        for (int L = 0; L < 5; ++L) {
            const int fits = lok[L] && (lx1[L] - lx0[L] + 1) * (ly1[L] - ly0[L] + 1) <= SWEEP_CAP;
            if (L == 0) gst = fits;
            const bool all_fit = __ballot(live && !fits) == 0;
            if (L > 0 && all_fit) { level = L; gx0 = lx0[L]; gy0 = ly0[L]; gx1 = lx1[L]; gy1 = ly1[L]; gst = 1; }
        }
        // (all_fit is monotone: a run that fits implies its halves fit, so the last level taken is the largest)
        level = __builtin_amdgcn_readfirstlane(level);
        if (live && (lane & ((1 << level) - 1)) == 0) {
            int* gq = grp[lane >> level];
            gq[0] = gx0; gq[1] = gy0; gq[2] = gx1 - gx0 + 1; gq[3] = gy1 - gy0 + 1; gq[4] = gst;
        }
        if (lane == 0) { hdr[0] = level; }
    }
    __syncthreads();
    const int level = __builtin_amdgcn_readfirstlane(hdr[0]);
    const int ngroups = (noct + (1 << level) - 1) >> level;

    const int pix = y * W + x;
    // per-lane base of the pair's output + wave-uniform plane offsets (kept on the scalar unit)
    // octets are visited in order, so the output address is a running per-lane pointer
    float* optr = a.out + (LAYOUT == 0 ? (size_t)p * D * HW + pix
                                       : c4_offset(p, LAYOUT == 1 ? D / 4 + 1 : D / 8 + 1, 0, HW, pix));
    const size_t plane = (size_t)HW * (LAYOUT == 0 ? 1 : 4);                 // floats per plane / per 16-byte channel group
    for (int g = 0; g < ngroups; ++g) {
        SweepBox bx;
        bx.rx0 = __builtin_amdgcn_readfirstlane(grp[g][0]); bx.ry0 = __builtin_amdgcn_readfirstlane(grp[g][1]);
        bx.rw = __builtin_amdgcn_readfirstlane(grp[g][2]); bx.rh = __builtin_amdgcn_readfirstlane(grp[g][3]);
        const bool staged = __builtin_amdgcn_readfirstlane(grp[g][4]) != 0;
#ifdef SWEEP_STATS
        if (tid == 0) {
            if (g == 0) atomicAdd(&sweep_stats[0], 1u);
            if (staged) { atomicAdd(&sweep_stats[1], 1u); atomicAdd(&sweep_stats[3], (unsigned)(bx.rw * bx.rh)); }
            else atomicAdd(&sweep_stats[2], (unsigned)(min((g + 1) << level, noct) - (g << level)));
        }
#endif
        if (staged) {
            if (g > 0) __syncthreads();                                      // every wave is done with the previous box
            const int n = bx.rw * bx.rh;
            const float inv_rw = 1.0f / (float)bx.rw;
            for (int i = tid; i < n; i += SWEEP_NT) {
                const int r = (int)(((float)i + 0.5f) * inv_rw), c = i - r * bx.rw;   // exact for n <= SWEEP_CAP
                const SweepTexel t = sweep_texel(rsrc, bx.rx0 + c, bx.ry0 + r, W, H, chan_bytes);
                box[3 * i] = t.u0; box[3 * i + 1] = t.u1; box[3 * i + 2] = t.u2;
            }
            __syncthreads();
        } else {                                                             // whole zero-extended image as the "box"
            bx.rx0 = -2; bx.ry0 = -2; bx.rw = W + 3; bx.rh = H + 3;
        }
        const float cu = -(0.5f + (float)bx.rx0), cv = -(0.5f + (float)bx.ry0);
        const float umax = (float)(bx.rw - 1), vmax = (float)(bx.rh - 1);
        const int o_end = min((g + 1) << level, noct);
        for (int o = g << level; o < o_end; ++o) {
            const int d0 = o * 8;
            float cost[8];
            if (staged) {
                SweepCoord cd[8];
                float4 tx[8][3];
                float zz[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) zz[j] = zsh[d0 + j];             // broadcast reads, d0 + j < CNM_MAX_PLANES
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8 + SWEEP_AHEAD; ++j) {
                    if (j < 8) {
                        cd[j] = sweep_coords(cu, cv, umax, vmax, a0, a1, a2, k0, k1, k2e, zz[j]);
                        unsigned off;                                        // byte offset of texel (yi, xi) in the box
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(cd[j].yi), "s"(bx.rw), "v"(cd[j].xi));
                        asm("v_mul_u32_u24 %0, %1, 48" : "=v"(off) : "v"(off));
                        const float4* t = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(box) + off);
                        tx[j][0] = t[0]; tx[j][1] = t[1]; tx[j][2] = t[2];
                    }
                    if (j >= SWEEP_AHEAD) {
                        const int i = j - SWEEP_AHEAD;
                        cost[i] = sweep_blend(tx[i][0], tx[i][1], tx[i][2], cd[i].wu, cd[i].wv, rr, rg, rb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < 8; ++j) {
                    const SweepCoord cj = sweep_coords(cu, cv, umax, vmax, a0, a1, a2, k0, k1, k2e, zsh[d0 + j]);
                    const SweepTexel t = sweep_texel(rsrc, (int)cj.xi + bx.rx0, (int)cj.yi + bx.ry0, W, H, chan_bytes);
                    const float c = sweep_blend(t.u0, t.u1, t.u2, cj.wu, cj.wv, rr, rg, rb);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) if (jj == j) cost[jj] = c;
                }
            }
            if (LAYOUT == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (pvalid && d0 + j < D) *optr = cost[j];
                    optr += plane;
                }
            } else if (LAYOUT == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (pvalid && d0 + 4 * q < D)
                        *reinterpret_cast<float4*>(optr) = make_float4(cost[4 * q], cost[4 * q + 1], cost[4 * q + 2], cost[4 * q + 3]);
                    if (d0 + 4 * q < D) optr += plane;
                }
            } else {
                sw_f16x8 h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (_Float16)cost[j];
                if (pvalid) *reinterpret_cast<sw_f16x8*>(optr) = h;
                optr += plane;
            }
        }
    }
    // optr now points at the channel group behind the D planes: the reference image (depthNet_model.py:233)
    if (LAYOUT == 1 && pvalid) *reinterpret_cast<float4*>(optr) = make_float4(rr, rg, rb, 0.f);
    if (LAYOUT == 2 && pvalid) {
        const sw_f16x8 h = {(_Float16)rr, (_Float16)rg, (_Float16)rb, 0, 0, 0, 0, 0};
        *reinterpret_cast<sw_f16x8*>(optr) = h;
    }
}

// The sweep needs no scratch any more; the argument stays in the ABI (callers size it with this query).
extern "C" size_t cnm_planesweep_workspace_floats(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
    return 4;
}

static int sweep_launch(int layout, const float* ref, const float* src, const float* hmkt, float* out,
                        float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                        double idepth_min, double idepth_max, void* stream) {
    (void)ws; (void)ws_floats;
    CNM_REQUIRE(ref && src && hmkt && out, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= CNM_MAX_PLANES, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout == 0 || D % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout != 2 || D % 8 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)B * S <= 65535 && (long long)H * W * 12 < (1ll << 31), CNM_ERR_BAD_ARG);
    SweepArgs a;
    a.ref = ref; a.src = src; a.hmkt = hmkt; a.out = out;
    a.B = B; a.S = S; a.H = H; a.W = W; a.D = D;
    a.idmin = idepth_min;
    a.idstep = (idepth_max - idepth_min) / (D - 1.0);                        // depthNet_model.py:194
    dim3 grid(cnm_ceil_div(W, SWEEP_TW), cnm_ceil_div(H, SWEEP_TH), B * S);
    if (layout == 0) planesweep_kernel<0><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
    else if (layout == 1) planesweep_kernel<1><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
    else planesweep_kernel<2><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_planesweep_volume_nchw_f32(const float* ref, const float* src, const float* hmkt, float* volume,
                                              float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                                              double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(0, ref, src, hmkt, volume, ws, ws_floats, B, S, H, W, D, idepth_min, idepth_max, stream);
}

extern "C" int cnm_planesweep_cat_c4_f32(const float* ref, const float* src, const float* hmkt, float* x,
                                         float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                                         double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(1, ref, src, hmkt, x, ws, ws_floats, B, S, H, W, D, idepth_min, idepth_max, stream);
}

extern "C" int cnm_planesweep_cat_c8_f16(const float* ref, const float* src, const float* hmkt, void* x,
                                         float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                                         double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(2, ref, src, hmkt, static_cast<float*>(x), ws, ws_floats, B, S, H, W, D, idepth_min, idepth_max, stream);
}
