// K0 + K1: camera prep and the fused plane-sweep warp + L1 cost volume.
//
// K0 replaces get_pixel_coordinates + process_camera_parameters
//    (reference depthnet/depth_util.py:13-56): 12 floats per (ref,src) pair; the pixel
//    grid is implicit in the thread index and KRKiUV [B,3,H*W] is never materialised.
// K1 replaces depthNet.getVolume (reference depthnet/depthNet_model.py:185-224), i.e.
//    64 x ~12 ATen launches per pair, with ONE launch over all pairs, planes and pixels.
//
// K1 = two launches on the caller's stream:
//  (a) texture pre-pass: every source image is re-laid once as an interleaved (r,g,b,0) float4
//      texture with a 2-texel zero border, [(H+4) x (W+4)] per pair (1.4 MB/pair of extra traffic
//      against 14.3 MB/pair of algorithmic traffic; it stays in L2 / Infinity Cache).
//  (b) sweep (gfx950, 64-lane waves): workgroup = 4 waves = SWEEP_TW x SWEEP_TH pixel tile, one
//      lane per reference pixel, all D planes walked by that lane; the (u,v,1) homography product
//      and the reference RGB stay in registers for the whole sweep.  Planes are taken in groups of
//      SWEEP_PG: for a group the workgroup computes the bounding box of the tile's footprint in the
//      source image (projective map => extremes at the 4 tile corners x 2 end planes) and copies
//      that box of texels straight into LDS with global_load_lds_dwordx4 (LDS-DMA: no VGPRs, no
//      ds_write, no bounds tests thanks to the zero border).  Every bilinear tap is then one
//      ds_read_b128 and the interpolation runs on packed fp32 pairs.  The box of group g+1 is in
//      flight while group g is computed (double buffer, one barrier per group).  If a footprint
//      does not fit the LDS box (extreme geometry, points behind the source camera) the group falls
//      back to bounds-checked global gathers with the same arithmetic.  Output leaves the registers
//      as coalesced stores: float4 (4 planes of one pixel) in the c4 layout, one float per plane
//      for NCHW.
// HBM-bound by design: algorithmic bytes per pair = 3HW*4 (ref) + 3HW*4 (src) + D*HW*4 (volume)
// (+ 4HW*4 for the ref group when emitting the concatenated conv input).
#include "cnm_common.h"

#define CNM_MAX_PLANES 128
#ifndef SWEEP_TW
#define SWEEP_TW 64           // tile width  (pixels, lanes along x)
#endif
#ifndef SWEEP_TH
#define SWEEP_TH 8            // tile height (SWEEP_TW * SWEEP_TH = 256 or 512 threads)
#endif
#ifndef SWEEP_PG
#define SWEEP_PG 16           // planes per staging group (multiple of 4)
#endif
#ifndef SWEEP_CAP
#define SWEEP_CAP 1664        // texels per LDS staging buffer (26 KB; two buffers)
#endif
#ifndef SWEEP_BATCH
#define SWEEP_BATCH 2         // planes per software-pipeline stage
#endif
#ifndef SWEEP_DEFER
#define SWEEP_DEFER 0         // 1: issue the stores of group g after the barrier of group g+1
#endif
#ifndef SWEEP_ASM_DMA
#define SWEEP_ASM_DMA 0       // 1: LDS-DMA from inline asm + counted vmcnt (measured: no gain over the builtin, kept for study)
#endif
#ifndef SWEEP_HYBRID
#define SWEEP_HYBRID 0          // 1: odd planes of every batch gather their taps from the texture through L1 (TA path)
#endif                         //    while even planes read the LDS box: the two data paths share the load
#ifndef SWEEP_MINW
#define SWEEP_MINW 1           // __launch_bounds__ min waves per SIMD (register cap)
#endif
#define SWEEP_NT (SWEEP_TW * SWEEP_TH)   // threads per workgroup (256 or 512)

struct SweepArgs {
    const float* ref; const float* src; const float* hmkt; float* out; const float4* tex;
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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (a) texture pre-pass: src [P,3,H,W] planar -> tex [P][H+4][W+4] float4 (r,g,b,0), zero border of 2
__global__ __launch_bounds__(256) void sweep_texture_kernel(const float* __restrict__ src, float4* __restrict__ tex,
                                                            int P, int H, int W) {
    const int TWp = W + 4, THp = H + 4;
    const long long total = (long long)P * THp * TWp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int tx = (int)(idx % TWp);
        const long long r = idx / TWp;
        const int ty = (int)(r % THp), p = (int)(r / THp);
        const int x = tx - 2, y = ty - 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const float* s = src + (size_t)p * 3 * H * W + (size_t)y * W + x;
            v.x = s[0]; v.y = s[(size_t)H * W]; v.z = s[2 * (size_t)H * W];
        }
        tex[idx] = v;
    }
}

// Staged sampling is split in three so that the sweep loop can software-pipeline it across planes:
//   sweep_coords : homography, perspective divide, floor/fraction, clamped LDS offset of tap (0,0)
//   4 x ds_read_b128 issued by the caller one batch of planes ahead
//   sweep_blend  : bilinear blend on packed fp32 pairs, |.-ref| summed over channels
// Out-of-image texels are zeros in the staged box, coordinates are clamped into it: no bounds tests.
// The .w lane of every texel is exactly zero; it is carried through the packed math (adds 0) so the
// taps stay single 128-bit reads.
struct SweepTap { int off; f32x2 w1; float xf, yf; };      // off: texel index of tap (0,0) in the LDS box; (xf,yf) box-relative

__device__ __forceinline__ SweepTap sweep_coords(f32x2 r0f, f32x2 rmaxf, float rwf, f32x2 a01, float a2,
                                                 f32x2 k01, float k2, float z) {
    const float den = fmaf(a2, z, k2) + 1e-6f;                               // depthNet_model.py:210-212
    float r = __builtin_amdgcn_rcpf(den);
    r = fmaf(fmaf(-den, r, 1.0f), r, r);                                     // one Newton step: ~0.5 ulp reciprocal
    const f32x2 zz = {z, z}, rr2 = {r, r}, half = {0.5f, 0.5f};
    const f32x2 i = (a01 * zz + k01) * rr2 - half;                           // :213 + grid_sample unnormalise
    const f32x2 fl = {floorf(i.x), floorf(i.y)};
    const f32x2 rel = fl - r0f;
    const float xf = __builtin_amdgcn_fmed3f(rel.x, 0.f, rmaxf.x), yf = __builtin_amdgcn_fmed3f(rel.y, 0.f, rmaxf.y);
    SweepTap t;
#ifdef SWEEP_ABL_NOCOORD
    t.off = (int)(z * 3.f); t.w1 = a01 * (f32x2){z, z};
    return t;
#endif
    t.off = (int)fmaf(yf, rwf, xf);                                          // exact: integers < 2^24
    t.w1 = i - fl; t.xf = xf; t.yf = yf;
    return t;
}

__device__ __forceinline__ float sweep_blend(const float4 p00, const float4 p01, const float4 p10, const float4 p11,
                                             f32x2 w1, f32x2 ref_rg, f32x2 ref_b0) {
    const f32x2 one = {1.f, 1.f};
    const f32x2 w0 = one - w1;
    const f32x2 wx = {w0.x, w1.x};
    const f32x2 wt = wx * (f32x2){w0.y, w0.y}, wb = wx * (f32x2){w1.y, w1.y};   // (w00,w01), (w10,w11)
    const f32x2 w00 = {wt.x, wt.x}, w01 = {wt.y, wt.y}, w10 = {wb.x, wb.x}, w11 = {wb.y, wb.y};
    f32x2 lo = w00 * (f32x2){p00.x, p00.y} - ref_rg;                         // warped - ref folded into the FMA chain
    f32x2 hi = w00 * (f32x2){p00.z, p00.w} - ref_b0;
    lo = w01 * (f32x2){p01.x, p01.y} + lo; hi = w01 * (f32x2){p01.z, p01.w} + hi;
    lo = w10 * (f32x2){p10.x, p10.y} + lo; hi = w10 * (f32x2){p10.z, p10.w} + hi;
    lo = w11 * (f32x2){p11.x, p11.y} + lo; hi = w11 * (f32x2){p11.z, p11.w} + hi;
    return (__builtin_fabsf(lo.x) + __builtin_fabsf(lo.y)) + __builtin_fabsf(hi.x + hi.y);   // :222-223 (hi.y == 0 exactly)
}

// Same arithmetic, texels gathered from the planar source with per-corner bounds tests
// (grid_sample zeros padding).  Used when a footprint does not fit the LDS box.
__device__ __forceinline__ float sweep_sample_global(const float* __restrict__ srcp, int H, int W, int HW,
                                                     f32x2 a01, float a2, f32x2 k01, float k2, float z,
                                                     f32x2 ref_rg, f32x2 ref_b0) {
    const float den = fmaf(a2, z, k2) + 1e-6f;
    float r = __builtin_amdgcn_rcpf(den);
    r = fmaf(fmaf(-den, r, 1.0f), r, r);
    const f32x2 zz = {z, z}, rr2 = {r, r}, half = {0.5f, 0.5f}, one = {1.f, 1.f};
    const f32x2 t = a01 * zz + k01;
    const f32x2 i = t * rr2 - half;
    const f32x2 fl = {floorf(i.x), floorf(i.y)};
    const f32x2 w1 = i - fl, w0 = one - w1;
    float4 p00, p01, p10, p11;
    p00 = p01 = p10 = p11 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fabsf(i.x) < 1e7f && fabsf(i.y) < 1e7f) {                            // false for NaN/inf as well
        const int xi = (int)fl.x, yi = (int)fl.y;
        const bool x0in = (unsigned)xi < (unsigned)W, x1in = (unsigned)(xi + 1) < (unsigned)W;
        const bool y0in = (unsigned)yi < (unsigned)H, y1in = (unsigned)(yi + 1) < (unsigned)H;
        const float* s = srcp + (ptrdiff_t)yi * W + xi;
        if (y0in && x0in) { p00.x = s[0]; p00.y = s[HW]; p00.z = s[2 * HW]; }
        if (y0in && x1in) { p01.x = s[1]; p01.y = s[HW + 1]; p01.z = s[2 * HW + 1]; }
        if (y1in && x0in) { p10.x = s[W]; p10.y = s[HW + W]; p10.z = s[2 * HW + W]; }
        if (y1in && x1in) { p11.x = s[W + 1]; p11.y = s[HW + W + 1]; p11.z = s[2 * HW + W + 1]; }
    }
    const f32x2 wy0 = {w0.y, w0.y}, wy1 = {w1.y, w1.y};
    const f32x2 wx = {w0.x, w1.x};
    const f32x2 wt = wx * wy0, wb = wx * wy1;
    const f32x2 w00 = {wt.x, wt.x}, w01 = {wt.y, wt.y}, w10 = {wb.x, wb.x}, w11 = {wb.y, wb.y};
    f32x2 lo = w00 * (f32x2){p00.x, p00.y} - ref_rg;
    f32x2 hi = w00 * (f32x2){p00.z, p00.w} - ref_b0;
    lo = w01 * (f32x2){p01.x, p01.y} + lo; hi = w01 * (f32x2){p01.z, p01.w} + hi;
    lo = w10 * (f32x2){p10.x, p10.y} + lo; hi = w10 * (f32x2){p10.z, p10.w} + hi;
    lo = w11 * (f32x2){p11.x, p11.y} + lo; hi = w11 * (f32x2){p11.z, p11.w} + hi;
    return (__builtin_fabsf(lo.x) + __builtin_fabsf(lo.y)) + __builtin_fabsf(hi.x + hi.y);
}

struct SweepBox { int rx0, ry0, rw, rh; bool staged; };
#ifdef SWEEP_STATS
__device__ unsigned int sweep_stats[2];     // debug builds only: groups served from LDS / by the global fallback
#endif
#ifdef SWEEP_TRACE
__device__ long long sweep_trace[4096][40];  // debug builds only: s_memtime stamps of wave 0 per workgroup
#define TRACE(slot) do { if (threadIdx.x == 0 && tr_blk < 4096 && (slot) < 40) sweep_trace[tr_blk][slot] = clock64(); } while (0)
#else
#define TRACE(slot) do {} while (0)
#endif

typedef _Float16 sw_f16x8 __attribute__((ext_vector_type(8)));

template <int LAYOUT>   // 0: volume [P,D,H,W] fp32   1: c4 [P,D/4+1,H,W,4] fp32   2: c8 [P,D/8+1,H,W,8] fp16
__global__ __launch_bounds__(SWEEP_NT, SWEEP_MINW) void planesweep_kernel(const SweepArgs a) {
    static_assert((SWEEP_NT == 256 || SWEEP_NT == 512) && (SWEEP_TW & (SWEEP_TW - 1)) == 0 && SWEEP_TW <= 64 &&
                  SWEEP_PG % 4 == 0 && SWEEP_PG % SWEEP_BATCH == 0 && CNM_MAX_PLANES % SWEEP_PG == 0, "tile");
    // one LDS object (a second __shared__ array makes hipcc drain the LDS-DMA queue early): two texel
    // boxes, the per-group footprint boxes and the plane depths.  The depths arrive in the kernel
    // argument segment, which is host-visible memory: reading them group by group costs a PCIe-class
    // round trip per group, so they are copied to LDS once.
    __shared__ float4 smem[2 * SWEEP_CAP + (CNM_MAX_PLANES / 4) * 2 + CNM_MAX_PLANES / 4];
    float4 (*tex)[SWEEP_CAP] = reinterpret_cast<float4 (*)[SWEEP_CAP]>(smem);
    int (*boxes)[8] = reinterpret_cast<int (*)[8]>(smem + 2 * SWEEP_CAP);
    float* zsh = reinterpret_cast<float*>(smem + 2 * SWEEP_CAP + (CNM_MAX_PLANES / 4) * 2);
    if (threadIdx.x < CNM_MAX_PLANES) zsh[threadIdx.x] = a.z[threadIdx.x];

#ifdef SWEEP_TRACE
    const int tr_blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    TRACE(0);
    const int lane = threadIdx.x & 63;
    const int wbase = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63);
    const int tx0 = blockIdx.x * SWEEP_TW, ty0 = blockIdx.y * SWEEP_TH;
    const int x = tx0 + (threadIdx.x & (SWEEP_TW - 1)), y = ty0 + threadIdx.x / SWEEP_TW;
    const int p = blockIdx.z, b = p / a.S;
    const int H = a.H, W = a.W, HW = H * W, D = a.D, TWp = W + 4;
    const bool pvalid = x < W && y < H;

    const float* hk = a.hmkt + (size_t)p * 12;
    const float h00 = hk[0], h01 = hk[1], h02 = hk[2], h10 = hk[3], h11 = hk[4], h12 = hk[5];
    const float h20 = hk[6], h21 = hk[7], h22 = hk[8], k2 = hk[11];
    const f32x2 k01 = {hk[9], hk[10]};
    const float fx_ = (float)x, fy_ = (float)y;
    const f32x2 a01 = {fmaf(h00, fx_, fmaf(h01, fy_, h02)), fmaf(h10, fx_, fmaf(h11, fy_, h12))};
    const float a2 = fmaf(h20, fx_, fmaf(h21, fy_, h22));

    const float* refp = a.ref + (size_t)b * 3 * HW + (size_t)y * W + x;
    float rr = 0.f, rg = 0.f, rb = 0.f;
    if (pvalid) { rr = refp[0]; rg = refp[HW]; rb = refp[2 * HW]; }
    const f32x2 ref_rg = {rr, rg}, ref_b0 = {rb, 0.f};
    const float* srcp = a.src + (size_t)p * 3 * HW;
    const float4* texp = a.tex + (size_t)p * (H + 4) * TWp;

    // tile corners for the footprint box: lanes 0..7 = 4 corners x {first,last plane of group}
    const int cxi = (lane & 1) ? min(tx0 + SWEEP_TW - 1, W - 1) : tx0;
    const int cyi = (lane & 2) ? min(ty0 + SWEEP_TH - 1, H - 1) : ty0;
    const float cxf = (float)cxi, cyf = (float)cyi;
    const float ca0 = fmaf(h00, cxf, fmaf(h01, cyf, h02));
    const float ca1 = fmaf(h10, cxf, fmaf(h11, cyf, h12));
    const float ca2 = fmaf(h20, cxf, fmaf(h21, cyf, h22));

    // ---- footprint boxes of all plane groups, once per workgroup: lane (8g + c) of wave 0 projects tile
    // corner (c&3) on the first (c<4) / last (c>=4) plane of group g; an 8-lane min/max gives the box,
    // which is parked in LDS and read back wave-uniformly when the group is processed.
    const int ngroups = (D + SWEEP_PG - 1) / SWEEP_PG;
    for (int gbase = 0; gbase < ngroups; gbase += 8) {        // wave 0 only; 8 groups per pass
        if (threadIdx.x < 64) {
            const int g = min(gbase + (lane >> 3), ngroups - 1);
            const int d0 = g * SWEEP_PG, d1 = min(d0 + SWEEP_PG, D) - 1;
            const float zc = a.z[(lane & 4) ? d1 : d0];
            const float den = fmaf(ca2, zc, k2) + 1e-6f;
            const float u = fast_div(fmaf(ca0, zc, k01.x), den), v = fast_div(fmaf(ca1, zc, k01.y), den);
            int okc = (den > 1e-4f) && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f);
            float umin = u, umax = u, vmin = v, vmax = v;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                umin = fminf(umin, __shfl_xor(umin, m, 8)); umax = fmaxf(umax, __shfl_xor(umax, m, 8));
                vmin = fminf(vmin, __shfl_xor(vmin, m, 8)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 8));
                okc &= __shfl_xor(okc, m, 8);
            }
            // sample corners x0 = floor(u-0.5) .. x0+1, one texel of safety margin either side,
            // clipped to the texture's zero border [-2, W+1] x [-2, H+1]
            const int rx0 = (int)fminf(fmaxf(floorf(umin - 0.5f) - 1.f, -2.f), (float)W);
            const int rx1 = (int)fminf(fmaxf(floorf(umax - 0.5f) + 2.f, (float)(rx0 + 1)), (float)(W + 1));
            const int ry0 = (int)fminf(fmaxf(floorf(vmin - 0.5f) - 1.f, -2.f), (float)H);
            const int ry1 = (int)fminf(fmaxf(floorf(vmax - 0.5f) + 2.f, (float)(ry0 + 1)), (float)(H + 1));
            if ((lane & 7) == 0 && gbase + (lane >> 3) < ngroups) {
                int* bx = boxes[g];
                bx[0] = rx0; bx[1] = ry0; bx[2] = rx1 - rx0 + 1; bx[3] = ry1 - ry0 + 1;
                bx[4] = okc && ((rx1 - rx0 + 1) * (ry1 - ry0 + 1) <= SWEEP_CAP);
            }
        }
    }
    __syncthreads();
    auto box_of = [&](int g) {
        SweepBox bx;
#ifdef SWEEP_ABL_CONSTBOX
        bx.rx0 = tx0; bx.ry0 = ty0; bx.rw = 80; bx.rh = 10; bx.staged = true; return bx;
#endif
        bx.rx0 = __builtin_amdgcn_readfirstlane(boxes[g][0]); bx.ry0 = __builtin_amdgcn_readfirstlane(boxes[g][1]);
        bx.rw = __builtin_amdgcn_readfirstlane(boxes[g][2]); bx.rh = __builtin_amdgcn_readfirstlane(boxes[g][3]);
        bx.staged = __builtin_amdgcn_readfirstlane(boxes[g][4]) != 0;
        return bx;
    };
    // LDS-DMA copy of a box: texel i = r*rw + c of the box lands at tb[i]; lane l of a wave owns i = chunk + l
    auto stage = [&](const SweepBox& bx, float4* tb) {
        const int n = bx.rw * bx.rh;
        const float inv_rw = 1.0f / (float)bx.rw;
        const float4* base = texp + (size_t)(bx.ry0 + 2) * TWp + (bx.rx0 + 2);
        for (int i0 = wbase; i0 < n; i0 += SWEEP_NT) {
            const int i = i0 + lane;
            if (i < n) {
                const int r = (int)(((float)i + 0.5f) * inv_rw), c = i - r * bx.rw;
                const float4* gsrc = base + r * TWp + c;
#if SWEEP_ASM_DMA
                // hipcc does not see this load: it is waited for with a COUNTED vmcnt before the barrier,
                // so the output stores issued after it are never drained (guide 5.7: M0 set and used in
                // one statement, s_nop 0 between the SALU write of M0 and the DMA).
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) float4*)(tb + i0));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                 (__attribute__((address_space(3))) void*)(tb + i0), 16, 0, 0);
#endif
            }
        }
    };

    // Stores of group g are issued AFTER the barrier of group g+1: the barrier's vmcnt(0) (needed for the
    // LDS-DMA) then only ever waits for stores that had a whole compute phase to drain.
    float pend[SWEEP_PG];
    auto emit = [&](int d0) {
#ifdef SWEEP_ABL_NOSTORE
#pragma unroll
        for (int j = 0; j < SWEEP_PG; ++j) asm volatile("" :: "v"(pend[j]));
        if (a.D >= 0) return;
#endif
        if (!pvalid) return;
        if (LAYOUT == 0) {
#pragma unroll
            for (int j = 0; j < SWEEP_PG; ++j)
                if (d0 + j < D) a.out[((size_t)p * D + d0 + j) * HW + (size_t)y * W + x] = pend[j];
        } else if (LAYOUT == 1) {
#pragma unroll
            for (int q = 0; q < SWEEP_PG / 4; ++q)
                if (d0 + 4 * q < D)
                    *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, (d0 >> 2) + q, HW, y * W + x)) =
                        make_float4(pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < SWEEP_PG / 8; ++q)
                if (d0 + 8 * q < D) {
                    sw_f16x8 h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) h[j] = (_Float16)pend[8 * q + j];
                    *reinterpret_cast<sw_f16x8*>(a.out + c4_offset(p, D / 8 + 1, (d0 >> 3) + q, HW, y * W + x)) = h;
                }
        }
    };
    TRACE(1);
    // counted waits need a fixed number of stores per wave and group: full tiles and full groups only
    [[maybe_unused]] const bool counted = __builtin_amdgcn_readfirstlane((tx0 + SWEEP_TW <= W) && (ty0 + SWEEP_TH <= H) && (D % SWEEP_PG == 0) && !SWEEP_DEFER);
    SweepBox cur = box_of(0);
    if (cur.staged) stage(cur, tex[0]);
    TRACE(2);
    for (int g = 0; g < ngroups; ++g) {
        const int d0 = g * SWEEP_PG;
        SweepBox nxt = cur;
#ifndef SWEEP_ABL_NOBOX
        if (g + 1 < ngroups) nxt = box_of(g + 1);
#endif
        TRACE(3 + 4 * g);
#ifndef SWEEP_ABL_NOGROUPSYNC
#if SWEEP_ASM_DMA
        // DMA(g) was issued before the stores of group g-1: wait for it but not for those stores
        if (counted && g > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LAYOUT == 0 ? SWEEP_PG : SWEEP_PG / 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();   // box g has landed; every wave is done reading the other buffer
#endif
        TRACE(4 + 4 * g);
        if (SWEEP_DEFER && g > 0) emit(d0 - SWEEP_PG);
#ifndef SWEEP_ABL_NODMA
        if (g + 1 < ngroups && nxt.staged) stage(nxt, tex[(g + 1) & 1]);
#endif
        TRACE(5 + 4 * g);
        const float4* tb = tex[g & 1];

#ifdef SWEEP_STATS
        if (threadIdx.x == 0) atomicAdd(&sweep_stats[cur.staged ? 0 : 1], 1u);
#endif
        float zs[SWEEP_PG];
#pragma unroll
        for (int j = 0; j < SWEEP_PG; ++j) zs[j] = zsh[d0 + j];             // LDS broadcast reads, d0+j < CNM_MAX_PLANES
        float cost[SWEEP_PG];
        if (cur.staged) {
            const f32x2 r0f = {(float)cur.rx0, (float)cur.ry0}, rmaxf = {(float)(cur.rw - 2), (float)(cur.rh - 2)};
            const float rwf = (float)cur.rw;
            const int rw = cur.rw;
            // software pipeline over batches of SWEEP_BATCH planes: the taps of batch k+1 are in flight
            // while batch k is blended
            constexpr int NB = SWEEP_PG / SWEEP_BATCH, BT = SWEEP_BATCH;
            SweepTap tap[2][BT];
            float4 tx[2][BT][4];
#pragma unroll
            for (int u = 0; u < BT; ++u) tap[0][u] = sweep_coords(r0f, rmaxf, rwf, a01, a2, k01, k2, zs[u]);
            const float twpf = (float)TWp;
            const float4* tbox = texp + (size_t)(cur.ry0 + 2) * TWp + (cur.rx0 + 2);      // texture address of the box origin
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                if (SWEEP_HYBRID && (u & 1)) {
                    const float4* q = tbox + (int)fmaf(tap[0][u].yf, twpf, tap[0][u].xf);
                    tx[0][u][0] = q[0]; tx[0][u][1] = q[1]; tx[0][u][2] = q[TWp]; tx[0][u][3] = q[TWp + 1];
                } else {
                    const float4* q = tb + tap[0][u].off;
                    tx[0][u][0] = q[0]; tx[0][u][1] = q[1]; tx[0][u][2] = q[rw]; tx[0][u][3] = q[rw + 1];
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int c = k & 1, n = c ^ 1;
                if (k + 1 < NB) {
#pragma unroll
                    for (int u = 0; u < BT; ++u) tap[n][u] = sweep_coords(r0f, rmaxf, rwf, a01, a2, k01, k2, zs[BT * (k + 1) + u]);
#pragma unroll
                    for (int u = 0; u < BT; ++u) {
                        if (SWEEP_HYBRID && (u & 1)) {
                            const float4* q = tbox + (int)fmaf(tap[n][u].yf, twpf, tap[n][u].xf);
                            tx[n][u][0] = q[0]; tx[n][u][1] = q[1]; tx[n][u][2] = q[TWp]; tx[n][u][3] = q[TWp + 1];
                        } else {
                            const float4* q = tb + tap[n][u].off;
                            tx[n][u][0] = q[0]; tx[n][u][1] = q[1]; tx[n][u][2] = q[rw]; tx[n][u][3] = q[rw + 1];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < BT; ++u)
#if defined(SWEEP_ABL_NOTAPS)
                    cost[BT * k + u] = tap[c][u].w1.x + (float)tap[c][u].off;
#elif defined(SWEEP_ABL_NOBLEND)
                    cost[BT * k + u] = tap[c][u].w1.x + tx[c][u][0].x + tx[c][u][1].x + tx[c][u][2].x + tx[c][u][3].x;
#else
                    cost[BT * k + u] = sweep_blend(tx[c][u][0], tx[c][u][1], tx[c][u][2], tx[c][u][3], tap[c][u].w1, ref_rg, ref_b0);
#endif
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < SWEEP_PG; ++j) {
                float zj = zs[0];
#pragma unroll
                for (int jj = 1; jj < SWEEP_PG; ++jj) if (jj == j) zj = zs[jj];
                const float c = sweep_sample_global(srcp, H, W, HW, a01, a2, k01, k2, zj, ref_rg, ref_b0);
#pragma unroll
                for (int jj = 0; jj < SWEEP_PG; ++jj) if (jj == j) cost[jj] = c;
            }
        }
#pragma unroll
        for (int j = 0; j < SWEEP_PG; ++j) pend[j] = cost[j];
        if (!SWEEP_DEFER) emit(d0);
        TRACE(6 + 4 * g);
        cur = nxt;
    }
    if (SWEEP_DEFER) emit((ngroups - 1) * SWEEP_PG);
    TRACE(39);
    if (LAYOUT == 1 && pvalid)
        *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, D / 4, HW, y * W + x)) = make_float4(rr, rg, rb, 0.f);
    if (LAYOUT == 2 && pvalid) {
        const sw_f16x8 h = {(_Float16)rr, (_Float16)rg, (_Float16)rb, 0, 0, 0, 0, 0};
        *reinterpret_cast<sw_f16x8*>(a.out + c4_offset(p, D / 8 + 1, D / 8, HW, y * W + x)) = h;
    }
}

// ------------------------------------------------------------------ K1, direct variant (no LDS)
// Same arithmetic; the four taps of a sample are 16-byte loads from the zero-bordered RGBA texture through L1/L2.
// No footprint boxes, no staging, no barriers, low register count (8 waves per SIMD hide the gather latency).
#ifndef SWEEP_DIRECT_UNROLL
#define SWEEP_DIRECT_UNROLL 4
#endif
template <int LAYOUT>
__global__ __launch_bounds__(256) void planesweep_direct_kernel(const SweepArgs a) {
    __shared__ float zsh[CNM_MAX_PLANES];
    if (threadIdx.x < CNM_MAX_PLANES) zsh[threadIdx.x] = a.z[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int p = blockIdx.z, b = p / a.S;
    const int H = a.H, W = a.W, HW = H * W, D = a.D, TWp = W + 4;
    if (x >= W || y >= H) return;
    const float* hk = a.hmkt + (size_t)p * 12;
    const float fx_ = (float)x, fy_ = (float)y;
    const f32x2 a01 = {fmaf(hk[0], fx_, fmaf(hk[1], fy_, hk[2])), fmaf(hk[3], fx_, fmaf(hk[4], fy_, hk[5]))};
    const float a2 = fmaf(hk[6], fx_, fmaf(hk[7], fy_, hk[8])), k2 = hk[11];
    const f32x2 k01 = {hk[9], hk[10]};
    const float* refp = a.ref + (size_t)b * 3 * HW + (size_t)y * W + x;
    const float rr = refp[0], rg = refp[HW], rb = refp[2 * HW];
    const f32x2 ref_rg = {rr, rg}, ref_b0 = {rb, 0.f};
    const float4* texp = a.tex + (size_t)p * (H + 4) * TWp + 2 * TWp + 2;          // texel (0,0)
    const f32x2 r0f = {-2.f, -2.f}, rmaxf = {(float)(W + 2), (float)(H + 2)};      // tap (0,0) clamped to [-2, W] x [-2, H]
    const float rwf = (float)TWp;
    for (int d0 = 0; d0 < D; d0 += SWEEP_DIRECT_UNROLL) {
        SweepTap tap[SWEEP_DIRECT_UNROLL];
        float4 tx[SWEEP_DIRECT_UNROLL][4];
        float cost[SWEEP_DIRECT_UNROLL];
#pragma unroll
        for (int u = 0; u < SWEEP_DIRECT_UNROLL; ++u) tap[u] = sweep_coords(r0f, rmaxf, rwf, a01, a2, k01, k2, zsh[d0 + u]);
#pragma unroll
        for (int u = 0; u < SWEEP_DIRECT_UNROLL; ++u) {
            const float4* q = texp + (tap[u].off - 2 * TWp - 2);                   // off is relative to texel (-2,-2)
            tx[u][0] = q[0]; tx[u][1] = q[1]; tx[u][2] = q[TWp]; tx[u][3] = q[TWp + 1];
        }
#pragma unroll
        for (int u = 0; u < SWEEP_DIRECT_UNROLL; ++u)
            cost[u] = sweep_blend(tx[u][0], tx[u][1], tx[u][2], tx[u][3], tap[u].w1, ref_rg, ref_b0);
        if (LAYOUT == 0) {
#pragma unroll
            for (int u = 0; u < SWEEP_DIRECT_UNROLL; ++u)
                if (d0 + u < D) a.out[((size_t)p * D + d0 + u) * HW + (size_t)y * W + x] = cost[u];
        } else {
#pragma unroll
            for (int q = 0; q < SWEEP_DIRECT_UNROLL / 4; ++q)
                *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, (d0 >> 2) + q, HW, y * W + x)) =
                    make_float4(cost[4 * q], cost[4 * q + 1], cost[4 * q + 2], cost[4 * q + 3]);
        }
    }
    if (LAYOUT == 1)
        *reinterpret_cast<float4*>(a.out + c4_offset(p, D / 4 + 1, D / 4, HW, y * W + x)) = make_float4(rr, rg, rb, 0.f);
}

extern "C" size_t cnm_planesweep_workspace_floats(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * S * (H + 4) * (W + 4) * 4;
}

static int sweep_launch(int layout, const float* ref, const float* src, const float* hmkt, float* out,
                        float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                        double idepth_min, double idepth_max, void* stream) {
    CNM_REQUIRE(ref && src && hmkt && out && ws, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(((uintptr_t)ws & 15) == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= CNM_MAX_PLANES, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout == 0 || D % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout != 2 || (D % 8 == 0 && SWEEP_PG % 8 == 0), CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)B * S <= 65535, CNM_ERR_BAD_ARG);
    SweepArgs a;
    a.ref = ref; a.src = src; a.hmkt = hmkt; a.out = out;
    a.B = B; a.S = S; a.H = H; a.W = W; a.D = D;
    const double step = (idepth_max - idepth_min) / (D - 1.0);              // depthNet_model.py:194
    for (int d = 0; d < CNM_MAX_PLANES; ++d)
        a.z[d] = d < D ? (float)(1.0 / (idepth_min + d * step)) : 0.f;      // :209 (python double -> fp32)
    CNM_REQUIRE(ws_floats >= cnm_planesweep_workspace_floats(B, S, H, W), CNM_ERR_WORKSPACE);
    a.tex = reinterpret_cast<const float4*>(ws);
    {
        const long long total = (long long)B * S * (H + 4) * (W + 4);
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        sweep_texture_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(src, reinterpret_cast<float4*>(ws), B * S, H, W);
    }
    dim3 grid(cnm_ceil_div(W, SWEEP_TW), cnm_ceil_div(H, SWEEP_TH), B * S);
#ifdef SWEEP_USE_DIRECT
    if (layout <= 1 && D % SWEEP_DIRECT_UNROLL == 0) {
        dim3 g2(cnm_ceil_div(W, 64), cnm_ceil_div(H, 4), B * S);
        if (layout == 0) planesweep_direct_kernel<0><<<g2, 256, 0, cnm_stream(stream)>>>(a);
        else planesweep_direct_kernel<1><<<g2, 256, 0, cnm_stream(stream)>>>(a);
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
#endif
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
