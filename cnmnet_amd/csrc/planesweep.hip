// K0 + K1: camera prep and the fused plane-sweep warp + L1 cost volume.
//
// K0 replaces get_pixel_coordinates + process_camera_parameters
//    (reference depthnet/depth_util.py:13-56): 12 floats per (ref,src) pair; the pixel
//    grid is implicit in the thread index and KRKiUV [B,3,H*W] is never materialised.
// K1 replaces depthNet.getVolume (reference depthnet/depthNet_model.py:185-224), i.e.
//    64 x ~12 ATen launches per pair, with ONE launch over all pairs, planes and pixels.
//
// K1 = ONE launch on the caller's stream (gfx950, 64-lane waves), persistent workgroups, [r5] built for EIGHT waves per SIMD:
//   two 16-wave workgroups per CU at <= 64 VGPRs, each with its own 78 KB LDS box, so that one workgroup's footprint pass,
//   box staging and barriers are covered by the other one's sweep (rounds 2-4 ran 4 waves per SIMD at 128 VGPRs and were
//   latency-bound: 34 % of the wave-cycles issuing).  A workgroup sweeps 64 x 16 pixel tiles drawn from a ticket counter;
//   one lane per reference pixel walks the planes with the per-pixel camera terms and the (negated) reference RGB in registers.
//   Per tile, wave 0 projects the 4 tile corners on the 2 end planes of every 8-plane octet (projective map => extremes
//   there): the footprint of the tile in the source image per octet, merged (DPP) into runs of 2, 4, 8, 16 octets; the
//   longest run whose footprint fits the LDS budget is staged ONCE into LDS as **column texels**: 6 floats
//   (P, dP/dy per channel; zero outside the image) = 24 bytes, built from the planar source by range-checked buffer loads
//   (6 per texel, no halo, no cross-lane traffic, no texture pre-pass).  [r5] Half the bytes of rounds 2-4's 48-byte
//   (P, dx, dy, dxy) texel: the same box holds twice the footprint, which is what lets two workgroups share a CU's LDS
//   without staging the source twice as often.  A bilinear sample reads the texels at xi and xi + 1 (adjacent: one address,
//   six ds_read_b64), L = P + wv dP/dy for both, and (1 - wu) L + wu R - ref as two FMAs whose addend carries the
//   reference pixel: 12 FMAs + 1 for the three channels (the 48-byte form: 9 FMAs + 3 subtractions).
//   Coordinates use the parallax form u' = a0/a2 + (k0 - (a0/a2) k2) / (a2 z + k2): one v_rcp_f32 and two FMAs
//   per plane (the reciprocal's rounding is scaled by the parallax, not by the coordinate).  The sample loop holds two
//   samples in flight (texel reads of sample j + 1 issued before the blend of sample j), four costs, and stores a float4
//   per four planes through a raw-buffer descriptor (one VGPR of address, the plane stride in an SGPR).
//   Octets whose footprint does not fit (extreme geometry, points behind the source camera, a2 near zero) gather the
//   same texels from global memory with the general division form.
// HBM-bound by design: algorithmic bytes per pair = 3HW*4 (ref) + 3HW*4 (src) + D*HW*4 (volume)
// (+ 4HW*4 for the ref group when emitting the concatenated conv input).
#include "cnm_common.h"
#include <algorithm>
#include <cstddef>

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
                                        float* __restrict__ hmkt, int B, int S, long long rc_bstride, long long sc_bstride) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B * S) return;
    const float* lc = ref_cam + (size_t)(p / S) * (size_t)rc_bstride;
    const float* rc = src_cam + (size_t)(p / S) * (size_t)sc_bstride + (size_t)(p % S) * 32;
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
    homography_terms_kernel<<<cnm_ceil_div(B * S, 64), 64, 0, cnm_stream(stream)>>>(ref_cam, src_cam, hmkt, B, S, 32, 32ll * S);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
// [r6] ... with the cameras as views: floats between consecutive frames of ref_cam [B][2][4][4] / src_cam [B][S][2][4][4] (0 = dense)
extern "C" int cnm_homography_terms_strided_f32(const float* ref_cam, long long rc_bstride, const float* src_cam, long long sc_bstride, float* hmkt,
                                                int B, int S, void* stream) {
    CNM_REQUIRE(ref_cam && src_cam && hmkt && B > 0 && S > 0 && (rc_bstride == 0 || rc_bstride >= 32) && (sc_bstride == 0 || sc_bstride >= 32ll * S), CNM_ERR_BAD_ARG);
    homography_terms_kernel<<<cnm_ceil_div(B * S, 64), 64, 0, cnm_stream(stream)>>>(ref_cam, src_cam, hmkt, B, S, rc_bstride ? rc_bstride : 32, sc_bstride ? sc_bstride : 32ll * S);
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
#define CNM_MAX_PLANES 128
#define SWEEP_TW 64                 // tile width  (pixels, lanes along x)
#ifndef SWEEP_TH
#define SWEEP_TH 16                 // tile height = waves per workgroup
#endif
#define SWEEP_NT (SWEEP_TW * SWEEP_TH)
#ifndef SWEEP_MINW
#define SWEEP_MINW 8                // waves per SIMD the register allocation must allow: two 16-wave workgroups per CU
#endif
#define SWEEP_WG_PER_CU (SWEEP_MINW * 256 / SWEEP_NT)
#ifndef SWEEP_CAP
#define SWEEP_CAP (SWEEP_WG_PER_CU == 1 ? 6600 : SWEEP_WG_PER_CU == 2 ? 3264 : 1600)   // texels per LDS box (24 B each)
#endif
// The queue ends with small units (guided self-scheduling): the last tiles are handed out as 2, 4, 8 partial sweeps of
// their plane range.  Zone sizes in SIXTEENTHS of the grid size (tiles, not units).
#ifndef SWEEP_TAIL_HALVES
#define SWEEP_TAIL_HALVES 6
#endif
#ifndef SWEEP_TAIL_QUARTERS
#define SWEEP_TAIL_QUARTERS 2
#endif
#ifndef SWEEP_TAIL_EIGHTHS
#define SWEEP_TAIL_EIGHTHS 0
#endif
#ifndef SWEEP_AHEAD
#define SWEEP_AHEAD 1               // 1: two samples' texels in registers (24), reads one sample ahead of the blend; 0: one sample's (12)
#endif
#ifndef SWEEP_ROLL_QUADS
#define SWEEP_ROLL_QUADS 0          // 1: the two quads of an octet as a loop (half the code of the sample loops; not for the fp16 layout, which packs eight planes)
#endif
#ifndef SWEEP_STORE_AUX
#define SWEEP_STORE_AUX 2           // the non-default cache policy of the output stores (gfx950: 1 sc0, 2 nt, 16 sc1), the kernel's AUX template argument next to 0: see sweep_store_policy()
#endif
#define SWEEP_MAX_OCT (CNM_MAX_PLANES / 8)
#define SWEEP_PASSES ((SWEEP_CAP + SWEEP_NT - 1) / SWEEP_NT)   // staging passes a full box needs

struct SweepArgs {
    const float* ref; const float* src; const float* hmkt; float* out;
    int frame_strides;              // [r6] images between consecutive frames of ref [B][3][H][W] (low 16 bits; dense: 1) and of src [B][S][3][H][W] (high 16 bits; dense: S) -- the views
                                    // frames[:, 0] / frames[:, 1:] of one [B][1 + S][3][H][W] tensor (1 + S each) need no copy.  One packed word: two 64-bit strides cost the launch 0.5 us of argument re-loads
    unsigned int* queue;            // [0] tile tickets, [1] workgroups that have left; zero between launches
    int B, S, H, W, D;
    // launch constants worked out on the host (sweep_launch): 512 workgroups x 16 waves need not each derive them
    int noct, ntx, tiles_per_pair, nfull, nh, nq, ne, nunits;                // work units: whole tiles, then halves / quarters / eighths
    float inv_tpp, inv_ntx, idmin_f, idstep_f;
    double idmin, idstep;           // plane d lies at depth 1 / (idmin + d * idstep)
#ifdef SWEEP_Z_ARGS
    float z[CNM_MAX_PLANES];        // experiment: the depths as 512 bytes of kernel arguments (measured: the launches get SLOWER in the step)
#endif
};

typedef _Float16 sw_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned sw_u32x4 __attribute__((ext_vector_type(4)));

// depth of plane d exactly as depthNet_model.py:193-194,209: python doubles, then one rounding to fp32
// (separately rounded multiply / add / divide: no contraction)
__host__ __device__ __forceinline__ float sweep_depth(double idmin, double idstep, int d) {
#pragma clang fp contract(off)
    const double m = (double)d * idstep;
    const double s = idmin + m;
    return (float)(1.0 / s);
}

template <int CTRL> __device__ __forceinline__ float sweep_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ int sweep_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
#define SWEEP_DPP_XOR1 0xB1         // quad_perm:[1,0,3,2]
#define SWEEP_DPP_XOR2 0x4E         // quad_perm:[2,3,0,1]
#define SWEEP_DPP_HALF_MIRROR 0x141 // lane 7-l of the 8-lane half row
#define SWEEP_DPP_ROR8 0x128        // row_ror:8 = lane l^8 of the 16-lane row

// A column texel of the zero-extended source image at (x, y): 24 bytes = three 8-byte units
//   (Pr, Pg) (dyr, dyg) (Pb, dyb),   dy = P(x, y+1) - P(x, y),   P = 0 outside the image,
// so grid_sample's per-corner zeros padding (depthNet_model.py:220) is carried by the data.  A bilinear sample at
// integer corner (xi, yi), fractions (wu, wv), reads the texels xi (L) and xi + 1 (R) of row yi - adjacent in the box:
//   sample - ref = wu (P_R + wv dy_R) + ((1 - wu) (P_L + wv dy_L) - ref).
// The (r, g) pairs sit in adjacent registers as the reads deliver them, so their four FMAs are v_pk_fma_f32.
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
struct SweepTexels { sw_f32x2 lp, ld, lb, rp, rd, rb; };
#ifndef SWEEP_PK
#define SWEEP_PK 0                 // 1: (r, g) pairs on v_pk_fma_f32, 2: also the two coordinates - measured [r5]: a packed FMA costs what two plain ones cost, no gain
#endif

__device__ __forceinline__ float sweep_blend(const SweepTexels& t, float wu, float wv, float nr, float ng, float nb) {
    const float nwu = 1.0f - wu;
#if SWEEP_PK
    const sw_f32x2 wv2 = {wv, wv}, wu2 = {wu, wu}, nwu2 = {nwu, nwu}, nrg = {nr, ng};
    const sw_f32x2 lrg = __builtin_elementwise_fma(wv2, t.ld, t.lp), srg = __builtin_elementwise_fma(wv2, t.rd, t.rp);
    const float lb = fmaf(wv, t.lb.y, t.lb.x), sb = fmaf(wv, t.rb.y, t.rb.x);
    const sw_f32x2 erg = __builtin_elementwise_fma(wu2, srg, __builtin_elementwise_fma(nwu2, lrg, nrg));   // nr = -ref: the subtraction rides in the addend
    const float eb = fmaf(wu, sb, fmaf(nwu, lb, nb));
    return (__builtin_fabsf(erg.x) + __builtin_fabsf(erg.y)) + __builtin_fabsf(eb);   // :222-223
#else
    const float lr = fmaf(wv, t.ld.x, t.lp.x), lg = fmaf(wv, t.ld.y, t.lp.y), lb = fmaf(wv, t.lb.y, t.lb.x);
    const float sr = fmaf(wv, t.rd.x, t.rp.x), sg = fmaf(wv, t.rd.y, t.rp.y), sb = fmaf(wv, t.rb.y, t.rb.x);
    const float er = fmaf(wu, sr, fmaf(nwu, lr, nr));                        // nr = -ref: the subtraction rides in the addend
    const float eg = fmaf(wu, sg, fmaf(nwu, lg, ng));
    const float eb = fmaf(wu, sb, fmaf(nwu, lb, nb));
    return (__builtin_fabsf(er) + __builtin_fabsf(eg)) + __builtin_fabsf(eb);   // :222-223
#endif
}

// the same two texels gathered from global memory (octets whose footprint does not fit the LDS box);
// out-of-image corners: buffer offset 0xFFFFFFFF is out of range and the load returns 0
__device__ __forceinline__ SweepTexels sweep_texels_global(__amdgpu_buffer_rsrc_t rsrc, int x, int y, int W, int H, unsigned chan_bytes) {
    const bool x0 = (unsigned)x < (unsigned)W, x1 = (unsigned)(x + 1) < (unsigned)W;
    const bool y0 = (unsigned)y < (unsigned)H, y1 = (unsigned)(y + 1) < (unsigned)H;
    const unsigned o = (unsigned)(y * W + x) * 4u, row = (unsigned)W * 4u;
    const unsigned a00 = (x0 && y0) ? o : 0xFFFFFFFFu, a01 = (x1 && y0) ? o + 4u : 0xFFFFFFFFu;
    const unsigned a10 = (x0 && y1) ? o + row : 0xFFFFFFFFu, a11 = (x1 && y1) ? o + row + 4u : 0xFFFFFFFFu;
    float p00[3], p01[3], d0[3], d1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned so = c * chan_bytes;
        p00[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a00, so, 0));
        p01[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a01, so, 0));
        d0[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a10, so, 0)) - p00[c];
        d1[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a11, so, 0)) - p01[c];
    }
    SweepTexels t;
    t.lp = sw_f32x2{p00[0], p00[1]}; t.ld = sw_f32x2{d0[0], d0[1]}; t.lb = sw_f32x2{p00[2], d0[2]};
    t.rp = sw_f32x2{p01[0], p01[1]}; t.rd = sw_f32x2{d1[0], d1[1]}; t.rb = sw_f32x2{p01[2], d1[2]};
    return t;
}

// box: rx0, ry0 = image coordinates of box texel (0,0); rw x rh texels.  The box is the tile's footprint (floor
// coordinates of the samples, one texel of margin) clipped to [-2, W] x [-2, H], plus one more column on the right (the
// R texel of the rightmost sample): rw = footprint width + 1.  Columns -2, -1, W, W + 1 and rows -2, H hold all-zero
// texels, so clamping a sample's coordinates into the box reproduces "no contribution" for everything outside the image.
struct SweepBox { int rx0, ry0, rw, rh; };
#ifdef SWEEP_SPAN
// debug builds (tools/k1_bench.hip -DSWEEP_SPAN): per workgroup, s_memrealtime at kernel entry, first tile start, last tile end; tiles done
__device__ unsigned int sweep_unit_ticks[4096];   // duration of every unit (tile), 100 MHz ticks
__device__ unsigned long long sweep_span[1024][4];
#endif
#ifdef SWEEP_TRACE
// debug builds (tools/k1_bench.hip -DSWEEP_TRACE): one record per unit: workgroup, unit, HW_ID, s_memrealtime at its start / after the footprints' barrier / end
__device__ unsigned long long sweep_trace[8192][4];
__device__ unsigned int sweep_trace_n;
__device__ unsigned long long sweep_trace_first[1024][6];   // first unit of a workgroup: kernel entry, loop top, footprints done (wave 0), barrier passed, first box staged, first octet stored
#endif
#ifdef SWEEP_BOXTIME
__device__ unsigned long long sweep_boxtime[4];   // debug builds (tools/k1_bench.hip -DSWEEP_BOXTIME): 100 MHz ticks summed over workgroups (thread 0's view): footprints + barrier, box staging, sweeps, units
#endif
#ifdef SWEEP_STATS
__device__ unsigned int sweep_stats[4];     // debug builds only: workgroups, staged boxes, octets gathered from global, box texels
#endif

// a wave-uniform float, pinned to an SGPR (float arithmetic leaves its results in VGPRs; hoisted out of the tile loop they
// would each hold a vector register for the whole kernel)
__device__ __forceinline__ float sweep_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

struct SweepCoord { unsigned off; float wu, wv; };       // byte offset of texel (yi, xi) in the box, fractions

template <bool CLAMP>
__device__ __forceinline__ void sweep_split(float iu, float iv, float umax, float vmax, unsigned& xi, unsigned& yi, float& wu, float& wv) {
    if (CLAMP) { iu = __builtin_amdgcn_fmed3f(iu, 0.f, umax); iv = __builtin_amdgcn_fmed3f(iv, 0.f, vmax); }
    xi = (unsigned)iu; yi = (unsigned)iv;                                    // floor (coordinates are >= 0 here)
    wu = __builtin_amdgcn_fractf(iu); wv = __builtin_amdgcn_fractf(iv);
}

// Parallax form of the map depthNet_model.py:210-213: u' = a0/a2 + (k0 - (a0/a2) k2e) / (a2 z + k2e) = U + A r.  U (the
// image of the point at infinity) and A are per-pixel constants, so a plane costs one reciprocal and two FMAs, and the
// reciprocal's rounding is scaled by the parallax |A r| instead of the coordinate |u'|: the plain v_rcp_f32 (1 ulp) is
// enough.  CLAMP = false for boxes that were not clipped by the image: every sample of a valid pixel then lies inside
// the box (margins included) and the two v_med3_f32 are saved; a lane of a ragged tile outside the image may then
// compute any address - LDS reads beyond the allocation return 0 and its result is never stored.
template <bool CLAMP>
__device__ __forceinline__ SweepCoord sweep_coords_parallax(float ug, float vg, float umax, float vmax, unsigned rwv, float pa, float pb,
                                                            float a2, float k2e, float z) {
    const float r = __builtin_amdgcn_rcpf(fmaf(a2, z, k2e));
    unsigned xi, yi; SweepCoord c;
#if SWEEP_PK >= 2
    const sw_f32x2 r2 = {r, r}, pab = {pa, pb}, uvg = {ug, vg};
    const sw_f32x2 iuv = __builtin_elementwise_fma(pab, r2, uvg);            // one v_pk_fma_f32 for both coordinates
    sweep_split<CLAMP>(iuv.x, iuv.y, umax, vmax, xi, yi, c.wu, c.wv);
#else
    sweep_split<CLAMP>(fmaf(pa, r, ug), fmaf(pb, r, vg), umax, vmax, xi, yi, c.wu, c.wv);
#endif
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c.off) : "v"(yi), "v"(rwv), "v"(xi));
    asm("v_mul_u32_u24 %0, %1, 24" : "=v"(c.off) : "v"(c.off));
    return c;
}

__device__ __forceinline__ SweepTexels sweep_texels_lds(const char* box, unsigned off) {
    typedef const volatile __attribute__((address_space(3))) sw_f32x2* lds_cv2;
    lds_cv2 t = (lds_cv2)(box + off);   // volatile: six ds_read_b64 (256 B/clk), not three ds_read2_b64 (128 B/clk)
    SweepTexels s;
    s.lp = t[0]; s.ld = t[1]; s.lb = t[2]; s.rp = t[3]; s.rd = t[4]; s.rb = t[5];
    return s;
}

// Four planes of one pixel from a staged box: the texel reads of sample j + 1 are issued before the blend of sample j
// (counted lgkmcnt waits, one scheduling region per sample); with eight waves per SIMD the rest of the LDS latency is
// the other waves' issue time.
template <bool CLAMP>
__device__ __forceinline__ void sweep_quad(const char* __restrict__ box, const float* __restrict__ zs, float (&cost)[4],
                                           float ug, float vg, float umax, float vmax, unsigned rwv, float pa, float pb,
                                           float a2, float k2v, float nr, float ng, float nb) {
    const float4 zq = *reinterpret_cast<const float4*>(zs);                  // broadcast read
    const float zz[4] = {zq.x, zq.y, zq.z, zq.w};
#if SWEEP_AHEAD
    SweepCoord cd[2];
    SweepTexels tx[2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (j < 4) {
            cd[j & 1] = sweep_coords_parallax<CLAMP>(ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, zz[j]);
            tx[j & 1] = sweep_texels_lds(box, cd[j & 1].off);
        }
        if (j >= 1) {
            const int i = j - 1;
            cost[i] = sweep_blend(tx[i & 1], cd[i & 1].wu, cd[i & 1].wv, nr, ng, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    // one sample's texels in registers; the coordinates of sample j + 1 are worked out while the reads of sample j travel
    SweepCoord cd = sweep_coords_parallax<CLAMP>(ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, zz[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const SweepTexels tx = sweep_texels_lds(box, cd.off);
        const float wu = cd.wu, wv = cd.wv;
        if (j < 3) cd = sweep_coords_parallax<CLAMP>(ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, zz[j + 1]);
        cost[j] = sweep_blend(tx, wu, wv, nr, ng, nb);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

// One LDS object per workgroup: texel box | plane depths | plane groups (two tile parities) | header (level, next tile) x 2.
// File scope, because the cold phases below are separate functions: the sample loop is the only code whose register
// allocation matters, and as callees the footprint pass, the box staging and the global-memory fallback get their own
// (the compiler otherwise hoists their invariants across the tile loop and spills around the sample loop's 64 registers).
#define SWEEP_BOX16 ((SWEEP_CAP * 24 + 15) / 16)
__shared__ float4 sweep_smem[SWEEP_BOX16 + CNM_MAX_PLANES / 4 + 2 * 2 * SWEEP_MAX_OCT + 1];
#define SWEEP_LDS_BOX (reinterpret_cast<char*>(sweep_smem))
#define SWEEP_LDS_Z (reinterpret_cast<float*>(sweep_smem + SWEEP_BOX16))
#define SWEEP_LDS_GRP (reinterpret_cast<int (*)[8]>(sweep_smem + SWEEP_BOX16 + CNM_MAX_PLANES / 4))
#define SWEEP_LDS_HDR (reinterpret_cast<int*>(sweep_smem + SWEEP_BOX16 + CNM_MAX_PLANES / 4 + 4 * SWEEP_MAX_OCT))

__device__ __forceinline__ int sweep_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sweep_rsrc(unsigned lo, unsigned hi, unsigned bytes) {
    // base assembled from two SGPR halves; the low half widened as UNSIGNED
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((unsigned long long)lo | ((unsigned long long)hi << 32)), 0, bytes, 0x00020000);
}

// The 12 camera terms of a pair through the SCALAR cache (s_load: lgkmcnt), straight into SGPRs.  A vector load would queue
// behind the wave's own output stores - vmcnt counts loads and stores in one in-order counter on gfx9 - and a tile would start
// by waiting for the previous tile's stores to be acknowledged by a memory system that is busy with exactly those.
typedef float sw_f32x4 __attribute__((ext_vector_type(4)));
struct SweepTerms { sw_f32x4 q0, q1, q2; };                       // h00 h01 h02 h10 | h11 h12 h20 h21 | h22 k0 k1 k2
__device__ __forceinline__ SweepTerms sweep_load_terms(const float* hmkt_pair) {
    SweepTerms t;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t.q0), "=&s"(t.q1), "=&s"(t.q2) : "s"(hmkt_pair) : "memory");
    return t;
}
#define SWEEP_TERMS(t) const float h00 = t.q0.x, h01 = t.q0.y, h02 = t.q0.z, h10 = t.q0.w, h11 = t.q1.x, h12 = t.q1.y, h20 = t.q1.z, h21 = t.q1.w, \
                                   h22 = t.q2.x, k0 = t.q2.y, k1 = t.q2.z, k2 = t.q2.w
__device__ __forceinline__ const float* sweep_uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const float*>((unsigned long long)lo | ((unsigned long long)hi << 32));
}

// ---- footprints of one unit, wave 0 only (serial work stays on one wave; the CU's other workgroup fills the gap).
// Lane 8j + c of pass q projects tile corner (c & 3) on the first (c < 4) / last plane of the unit's octet 8q + j; an
// 8-lane min / max (DPP) gives the octet's box in all eight lanes.  Merging with the lanes 8, 16, 32 away and with the
// other pass gives the boxes of every aligned run of 2, 4, 8, 16 octets; the longest run whose boxes all fit the LDS
// budget wins.  Results go to LDS (groups, level).
// NQ = passes of eight octets: 1 for units of up to 64 planes (half the merging work), 2 up to 128.
template <int NQ>
__device__ __forceinline__ void sweep_footprints_body(const float* hmkt_pair, int tx0_, int ty0_, int obeg_, int ocnt_, int W_, int H_, int D_,
                                                      float idmin_, float idstep_, int parity_) {
    const int tx0 = sweep_sgpr(tx0_), ty0 = sweep_sgpr(ty0_), obeg = sweep_sgpr(obeg_), ocnt = sweep_sgpr(ocnt_);
    const int W = sweep_sgpr(W_), H = sweep_sgpr(H_), D = sweep_sgpr(D_), parity = sweep_sgpr(parity_);
    const float idmin = sweep_uniform(idmin_), idstep = sweep_uniform(idstep_);
    int (*const grp)[8] = SWEEP_LDS_GRP;
    int* const hdr = SWEEP_LDS_HDR;
    const int lane = threadIdx.x & 63;
    const SweepTerms terms = sweep_load_terms(sweep_uniform_ptr(hmkt_pair));
    SWEEP_TERMS(terms);
    const int cxi = (lane & 1) ? min(tx0 + SWEEP_TW - 1, W - 1) : tx0;
    const int cyi = (lane & 2) ? min(ty0 + SWEEP_TH - 1, H - 1) : ty0;
    const float cxf = (float)cxi, cyf = (float)cyi;
    const float ca0 = fmaf(h00, cxf, fmaf(h01, cyf, h02));
    const float ca1 = fmaf(h10, cxf, fmaf(h11, cyf, h12));
    const float ca2 = fmaf(h20, cxf, fmaf(h21, cyf, h22));
    // the parallax form needs a2 (linear over the tile: extremes at the corners) away from zero, one sign
    const bool parallax_ok = __ballot(!(fabsf(ca2) >= 0.25f)) == 0 && (__ballot(ca2 < 0.f) == 0 || __ballot(ca2 > 0.f) == 0);
    int bx0[NQ], by0[NQ], bx1[NQ], by1[NQ], bok[NQ], bcl[NQ];        // box, footprint usable, box clipped by the image
    bool live[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int oct = q * 8 + (lane >> 3);                         // octet of this unit
        live[q] = oct < ocnt;
        const int o = obeg + min(oct, ocnt - 1);
        const int d0 = o * 8, d1 = min(d0 + 8, D) - 1;
        const float zc = __builtin_amdgcn_rcpf(fmaf((float)((lane & 4) ? d1 : d0), idstep, idmin));   // fp32 depths are enough for a box with margins
        const float den = fmaf(ca2, zc, k2) + 1e-6f;
        float rc = __builtin_amdgcn_rcpf(den);
        rc = fmaf(fmaf(-den, rc, 1.0f), rc, rc);
        const float u = fmaf(ca0, zc, k0) * rc, v = fmaf(ca1, zc, k1) * rc;
        int okc = (den > 1e-4f) && (fabsf(u) < 1e6f) && (fabsf(v) < 1e6f);
        float umin = u, umax = u, vmin = v, vmax = v;
#define SWEEP_RED8(CTRL) \
        umin = fminf(umin, sweep_dpp<CTRL>(umin)); umax = fmaxf(umax, sweep_dpp<CTRL>(umax)); \
        vmin = fminf(vmin, sweep_dpp<CTRL>(vmin)); vmax = fmaxf(vmax, sweep_dpp<CTRL>(vmax)); okc &= sweep_dpp<CTRL>(okc);
        SWEEP_RED8(SWEEP_DPP_XOR1) SWEEP_RED8(SWEEP_DPP_XOR2) SWEEP_RED8(SWEEP_DPP_HALF_MIRROR)
#undef SWEEP_RED8
        // texel indices floor(u - 0.5) of the samples, one texel of safety margin either side, clipped to
        // [-2, W] x [-2, H] (the outermost columns / rows of that range are all zeros)
        bx0[q] = (int)fminf(fmaxf(floorf(umin - 0.5f) - 1.f, -2.f), (float)W);
        bx1[q] = (int)fminf(fmaxf(floorf(umax - 0.5f) + 1.f, (float)bx0[q]), (float)W);
        by0[q] = (int)fminf(fmaxf(floorf(vmin - 0.5f) - 1.f, -2.f), (float)H);
        by1[q] = (int)fminf(fmaxf(floorf(vmax - 0.5f) + 1.f, (float)by0[q]), (float)H);
        bok[q] = okc;
        bcl[q] = !(floorf(umin - 0.5f) - 1.f >= -2.f && floorf(umax - 0.5f) + 1.f <= (float)W &&
                   floorf(vmin - 0.5f) - 1.f >= -2.f && floorf(vmax - 0.5f) + 1.f <= (float)H);
        if (!live[q]) { bx0[q] = 1 << 28; by0[q] = 1 << 28; bx1[q] = -(1 << 28); by1[q] = -(1 << 28); bok[q] = 1; bcl[q] = 0; }   // neutral
    }
    int level = 0, gx0[NQ], gy0[NQ], gx1[NQ], gy1[NQ], gst[NQ], gcl[NQ];
#pragma unroll
    for (int L = 0; L < 3 + NQ; ++L) {
        if (L == 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                bx0[q] = min(bx0[q], sweep_dpp<SWEEP_DPP_ROR8>(bx0[q])); by0[q] = min(by0[q], sweep_dpp<SWEEP_DPP_ROR8>(by0[q]));
                bx1[q] = max(bx1[q], sweep_dpp<SWEEP_DPP_ROR8>(bx1[q])); by1[q] = max(by1[q], sweep_dpp<SWEEP_DPP_ROR8>(by1[q]));
                bok[q] &= sweep_dpp<SWEEP_DPP_ROR8>(bok[q]); bcl[q] |= sweep_dpp<SWEEP_DPP_ROR8>(bcl[q]);
            }
        } else if (L == 2 || L == 3) {
            const int m = L == 2 ? 16 : 32;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                bx0[q] = min(bx0[q], __shfl_xor(bx0[q], m)); by0[q] = min(by0[q], __shfl_xor(by0[q], m));
                bx1[q] = max(bx1[q], __shfl_xor(bx1[q], m)); by1[q] = max(by1[q], __shfl_xor(by1[q], m));
                bok[q] &= __shfl_xor(bok[q], m); bcl[q] |= __shfl_xor(bcl[q], m);
            }
        } else if (L == 4) {
            bx0[0] = bx0[NQ - 1] = min(bx0[0], bx0[NQ - 1]); by0[0] = by0[NQ - 1] = min(by0[0], by0[NQ - 1]);
            bx1[0] = bx1[NQ - 1] = max(bx1[0], bx1[NQ - 1]); by1[0] = by1[NQ - 1] = max(by1[0], by1[NQ - 1]);
            bok[0] = bok[NQ - 1] = bok[0] & bok[NQ - 1]; bcl[0] = bcl[NQ - 1] = bcl[0] | bcl[NQ - 1];
        }
        bool bad = false;
        int fits[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // capacity in texels: (footprint width + the R column) x height
            const int rw = bx1[q] - bx0[q] + 2, rh = by1[q] - by0[q] + 1;
            const int rwc = min(max(rw, 0), SWEEP_CAP + 1), rhc = min(max(rh, 0), SWEEP_CAP + 1);   // 24-bit products
            fits[q] = bok[q] && __mul24(rwc, rhc) <= SWEEP_CAP;
            const bool run_live = ((q * 8 + (lane >> 3)) & ~((1 << L) - 1)) < ocnt;
            bad |= run_live && !fits[q];
        }
        const bool all_fit = __ballot(bad) == 0;
        if (L == 0 || all_fit) {                                          // monotone: a run that fits implies its halves fit
            level = L;
#pragma unroll
            for (int q = 0; q < NQ; ++q) { gx0[q] = bx0[q]; gy0[q] = by0[q]; gx1[q] = bx1[q]; gy1[q] = by1[q]; gst[q] = fits[q]; gcl[q] = bcl[q]; }
        }
    }
    level = __builtin_amdgcn_readfirstlane(level);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int oct = q * 8 + (lane >> 3);
        if ((lane & 7) == 0 && oct < ocnt && (oct & ((1 << level) - 1)) == 0) {
            int* gq = grp[parity * SWEEP_MAX_OCT + (oct >> level)];
            gq[0] = gx0[q]; gq[1] = gy0[q]; gq[2] = gx1[q] - gx0[q] + 2; gq[3] = gy1[q] - gy0[q] + 1;
            gq[4] = gst[q] && parallax_ok; gq[5] = gcl[q];
        }
    }
    if (lane == 0) hdr[2 * parity] = level;
}

__device__ __attribute__((noinline)) void sweep_footprints8(const float* hmkt_pair, int tx0, int ty0, int obeg, int ocnt, int W, int H, int D, float idmin, float idstep, int parity) {
    sweep_footprints_body<1>(hmkt_pair, tx0, ty0, obeg, ocnt, W, H, D, idmin, idstep, parity);
}
__device__ __attribute__((noinline)) void sweep_footprints16(const float* hmkt_pair, int tx0, int ty0, int obeg, int ocnt, int W, int H, int D, float idmin, float idstep, int parity) {
    sweep_footprints_body<2>(hmkt_pair, tx0, ty0, obeg, ocnt, W, H, D, idmin, idstep, parity);
}

// ---- stage a box: rh rows of rw column texels, item i = r rw + c, SWEEP_NT items per pass.  A lane loads its column of
// image rows y and y + 1 for the three channels; all loads of the box are issued first and travel while the slower waves
// finish the previous box (barrier_first: this is not the tile's first box, whose predecessor the tile barrier retired).
__device__ __attribute__((noinline)) void sweep_stage_box(unsigned src_lo_, unsigned src_hi_, int rx0_, int ry0_, int rw_, int rh_, int W_, int H_, int barrier_first_) {
    const unsigned src_lo = (unsigned)sweep_sgpr((int)src_lo_), src_hi = (unsigned)sweep_sgpr((int)src_hi_);
    const int rx0 = sweep_sgpr(rx0_), ry0 = sweep_sgpr(ry0_), rw = sweep_sgpr(rw_), rh = sweep_sgpr(rh_), W = sweep_sgpr(W_), H = sweep_sgpr(H_);
    const int barrier_first = sweep_sgpr(barrier_first_);
#if defined(SWEEP_NOSTAGE) && SWEEP_NOSTAGE == 1                          // debug builds (wrong output): staging free -- no loads, no LDS writes, no barriers
    return;
#endif
    const unsigned chan_bytes = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = sweep_rsrc(src_lo, src_hi, 3 * chan_bytes);
    char* const box = SWEEP_LDS_BOX;
    const int tid = threadIdx.x;
    const int n = rw * rh;
    const float inv_rw = sweep_uniform(1.0f / (float)rw);
    const int origin4 = (ry0 * W + rx0) * 4;                                 // byte offset of box texel (0,0) in a channel plane
    float p0[SWEEP_PASSES][3], p1[SWEEP_PASSES][3];
#pragma unroll
    for (int k = 0; k < SWEEP_PASSES; ++k) {
        if (k * SWEEP_NT >= n) break;                                        // wave-uniform: passes the box does not need
        const int i = k * SWEEP_NT + tid;
        const int r = (int)(((float)i + 0.5f) * inv_rw);                    // exact for i < 2^21 / rw
        int c, t;                                                           // 24-bit multiply-adds (v_mul_lo_u32 is quarter rate)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c) : "v"(r), "s"(-rw), "v"(i));          // c = i - r rw
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(r), "s"(W), "v"(c));            // texel offset from the origin
        const int xx = rx0 + c, yy = ry0 + r;
        const bool xin = i < n && (unsigned)xx < (unsigned)W;
        const unsigned o = (unsigned)(t * 4 + origin4);
        const unsigned o0 = (xin && (unsigned)yy < (unsigned)H) ? o : 0xFFFFFFFFu;
        const unsigned o1 = (xin && (unsigned)(yy + 1) < (unsigned)H) ? o + (unsigned)W * 4u : 0xFFFFFFFFu;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
#if defined(SWEEP_NOSTAGE) && SWEEP_NOSTAGE == 3                          // debug builds (wrong output): the staging's barriers and LDS writes without its loads
            p0[k][ch] = (float)o0; p1[k][ch] = (float)o1;
#else
            p0[k][ch] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o0, ch * chan_bytes, 0));
            p1[k][ch] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o1, ch * chan_bytes, 0));
#endif
        }
    }
    if (barrier_first) __syncthreads();                                      // every wave is done with the previous box
#pragma unroll
    for (int k = 0; k < SWEEP_PASSES; ++k) {
        if (k * SWEEP_NT >= n) break;
        const int i = k * SWEEP_NT + tid;
        if (i < n) {
            unsigned off;
            asm("v_mul_u32_u24 %0, %1, 24" : "=v"(off) : "v"(i));
            float2* tb = reinterpret_cast<float2*>(box + off);
            tb[0] = make_float2(p0[k][0], p0[k][1]);
            tb[1] = make_float2(p1[k][0] - p0[k][0], p1[k][1] - p0[k][1]);
            tb[2] = make_float2(p0[k][2], p1[k][2] - p0[k][2]);
        }
    }
    __syncthreads();
}

// ---- four planes of one pixel gathered from global memory (octets whose footprint does not fit the LDS box): general
// form, depthNet_model.py:210-213: (u', v') = (a0 z + k0, a1 z + k1) / (a2 z + k2 + 1e-6), minus the half pixel of
// grid_sample's unnormalisation; coordinates clamped into the zero-extended image [-2, W + 1] x [-2, H]
__device__ __attribute__((noinline)) float4 sweep_quad_global(unsigned src_lo_, unsigned src_hi_, int W_, int H_, const float* hmkt_pair, int x, int y, int d0_,
                                                              float nr, float ng, float nb) {
    const unsigned src_lo = (unsigned)sweep_sgpr((int)src_lo_), src_hi = (unsigned)sweep_sgpr((int)src_hi_);
    const int W = sweep_sgpr(W_), H = sweep_sgpr(H_), d0 = sweep_sgpr(d0_);
    const unsigned chan_bytes = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = sweep_rsrc(src_lo, src_hi, 3 * chan_bytes);
    const SweepTerms terms = sweep_load_terms(sweep_uniform_ptr(hmkt_pair));
    SWEEP_TERMS(terms);
    const float fx = (float)x, fy = (float)y;
    const float a0 = fmaf(h00, fx, fmaf(h01, fy, h02)), a1 = fmaf(h10, fx, fmaf(h11, fy, h12)), a2 = fmaf(h20, fx, fmaf(h21, fy, h22));
    const float k2e = k2 + 1e-6f;
    const float umax = (float)(W + 2), vmax = (float)(H + 2);              // box = columns -2 .. W + 1, rows -2 .. H
    float cost[4];
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const float z = SWEEP_LDS_Z[d0 + j];
        const float den = fmaf(a2, z, k2e);
        float r = __builtin_amdgcn_rcpf(den);
        r = fmaf(fmaf(-den, r, 1.0f), r, r);                                 // one Newton step: ~0.5 ulp reciprocal
        unsigned xi, yi; float wu, wv;
        sweep_split<true>(fmaf(fmaf(a0, z, k0), r, 1.5f), fmaf(fmaf(a1, z, k1), r, 1.5f), umax, vmax, xi, yi, wu, wv);
        const SweepTexels t = sweep_texels_global(rsrc, (int)xi - 2, (int)yi - 2, W, H, chan_bytes);
        const float c = sweep_blend(t, wu, wv, nr, ng, nb);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) if (jj == j) cost[jj] = c;
    }
    return make_float4(cost[0], cost[1], cost[2], cost[3]);
}

// Persistent workgroups: the grid is sized to the chip (two 16-wave workgroups per CU) and a workgroup sweeps tile
// blockIdx.x, then tiles drawn from a ticket counter in the caller's workspace (ws[0]: tickets, ws[1]: exits; both
// are zero between launches - the last workgroup to leave resets them).  Per tile: footprints (wave 0) ->
// [stage box -> sweep its planes]*.  The ticket and the camera terms of the NEXT tile are fetched while the current one is swept.
template <int LAYOUT, int AUX>   // LAYOUT 0: volume [P,D,H,W] fp32   1: c4 [P,D/4+1,H,W,4] fp32   2: c8 [P,D/8+1,H,W,8] fp16;  AUX: cache policy of the output stores
__global__ __launch_bounds__(SWEEP_NT) __attribute__((amdgpu_waves_per_eu(SWEEP_MINW, SWEEP_MINW))) void planesweep_kernel(const SweepArgs a) {
    char* const box = SWEEP_LDS_BOX;
    float* const zsh = SWEEP_LDS_Z;
    int (*const grp)[8] = SWEEP_LDS_GRP;
    int* const hdr = SWEEP_LDS_HDR;

#ifdef SWEEP_SPAN
    const unsigned long long span_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
#ifdef SWEEP_TRACE
    if (tid == 0 && blockIdx.x < 1024) sweep_trace_first[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
    bool trace_first = true;
#endif
    const int H = a.H, W = a.W, HW = H * W, D = a.D;
    const int noct = a.noct, ntx = a.ntx, tiles_per_pair = a.tiles_per_pair;
    {   // plane depths: kernel arguments -> LDS (not by wave 0, whose footprint pass everybody waits for)
        const int zt = tid - (SWEEP_NT >= 2 * CNM_MAX_PLANES ? CNM_MAX_PLANES : 0);
#ifdef SWEEP_Z_ARGS
        // (read through the kernarg segment pointer: indexing the by-value struct leaves a dead 48-byte stack object behind, and with it a scratch set-up per launch)
        const float* const zarg = reinterpret_cast<const float*>(static_cast<const char*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(SweepArgs, z));
        if (zt >= 0 && zt < CNM_MAX_PLANES) zsh[zt] = zarg[zt];
#else
        if (zt >= 0 && zt < CNM_MAX_PLANES) zsh[zt] = zt < D ? sweep_depth(a.idmin, a.idstep, zt) : 0.f;
#endif
    }
    // Work units: whole tiles first, then tiles cut into 2, 4, 8 partial sweeps: the launch ends with small units, not
    // with a tile-long tail of half-empty CUs.
    const int nh = a.nh, nq = a.nq, nfull = a.nfull, nunits = a.nunits;
    const float inv_tpp = a.inv_tpp, inv_ntx = a.inv_ntx, idmin_f = a.idmin_f, idstep_f = a.idstep_f;
#if defined(SWEEP_SPAN)
    struct Unit { int p, tx0, ty0, obeg, ocnt, tile; };
#else
    struct Unit { int p, tx0, ty0, obeg, ocnt; };
#endif
    auto decode = [&](int u) {                                               // unit -> pair, tile origin, octet range
        Unit q; q.obeg = 0; q.ocnt = noct;
        int t = u;
        if (u >= nfull) {
            int v = u - nfull, sh = 1;
            if (v < 2 * nh) t = nfull;
            else if (v < 2 * nh + 4 * nq) { v -= 2 * nh; sh = 2; t = nfull + nh; }
            else { v -= 2 * nh + 4 * nq; sh = 3; t = nfull + nh + nq; }
            t += v >> sh;
            const int part = v & ((1 << sh) - 1);
            q.obeg = (part * noct) >> sh;
            q.ocnt = (((part + 1) * noct) >> sh) - q.obeg;
        }
#if defined(SWEEP_SPAN)
        q.tile = t;
#endif
        // fp32 reciprocals (v_rcp_f32, 1 ulp: exact quotients for t < 2^20, checked by the launcher); back to SGPRs so the rest is scalar arithmetic
        q.p = __builtin_amdgcn_readfirstlane((int)(((float)t + 0.5f) * inv_tpp));
        const int rem = t - q.p * tiles_per_pair, tyi = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * inv_ntx));
        q.tx0 = (rem - tyi * ntx) * SWEEP_TW; q.ty0 = tyi * SWEEP_TH;
        return q;
    };
    // The queue.  A workgroup's first two units are static (blockIdx.x, blockIdx.x + gridDim.x: no burst of 512 atomics on one
    // word when the launch starts - a word serves ~85 tickets per us).  From its second unit on, the LAST wave draws the unit
    // after the current one at the top of the current one, while wave 0 works out the footprints and everybody else waits at
    // that barrier anyway, through the SCALAR cache path (s_atomic_add: lgkmcnt; a vector atomic would return behind the
    // wave's own output stores in vmcnt order), and parks it in the LDS header; the barrier publishes it, the waves read it
    // when the unit is done.
    int unit = blockIdx.x;
    bool first_unit = true;
    Unit cur = decode(min(unit, nunits - 1));

#ifdef SWEEP_SPAN
    if (tid == 0 && blockIdx.x < 1024) { sweep_span[blockIdx.x][0] = span_entry; sweep_span[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); sweep_span[blockIdx.x][3] = 0; }
#endif
    for (int parity = 0; unit < nunits; parity ^= 1) {
#ifdef SWEEP_SPAN
        const unsigned long long span_u0 = __builtin_amdgcn_s_memrealtime();
        const int cur_tile_span = cur.tile;
#endif
#ifdef SWEEP_TRACE
        const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (tid >= SWEEP_NT - 64) {
            unsigned nxt = (unsigned)unit + gridDim.x;                         // the first unit's successor, and every one without a queue: a fixed stride
            if (a.queue && !first_unit) {
                nxt = 1u;
                asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(nxt) : "s"(a.queue) : "memory");
                nxt += 2u * gridDim.x;
            }
            if (tid == SWEEP_NT - 64) hdr[2 * parity + 1] = (int)nxt;
        }
        first_unit = false;
        const int p = cur.p, tx0 = cur.tx0, ty0 = cur.ty0, obeg = cur.obeg, ocnt = cur.ocnt;
        int tu = tid;
        asm volatile("" : "+v"(tu));                                         // per-lane values are derived per tile, not hoisted and held
        const int lane = tu & 63, wave = tu >> 6;
        const int x = tx0 + lane, y = ty0 + wave;
        const bool pvalid = x < W && y < H;
        // the reference pixel, negated (the blend's addend): needed at the first blend only, i.e. after the first staging
        float nr = 0.f, ng = 0.f, nb = 0.f;
        if (pvalid) {
#ifdef SWEEP_DENSE   // A/B build: the dense addressing of rounds 1-5 (tools/k1_stride_ab.sh)
            const float* refp = a.ref + (size_t)(p / a.S) * 3 * HW + (size_t)y * W + x;
#else
            const float* refp = a.ref + (size_t)((p / a.S) * (a.frame_strides & 0xFFFF)) * 3 * HW + (size_t)y * W + x;
#endif
            nr = -refp[0]; ng = -refp[HW]; nb = -refp[2 * HW];
        }
#ifdef SWEEP_DENSE
        const unsigned long long srcb = reinterpret_cast<unsigned long long>(a.src + (size_t)p * 3 * HW);
#else
        const unsigned long long srcb = reinterpret_cast<unsigned long long>(a.src + (size_t)((p / a.S) * (int)((unsigned)a.frame_strides >> 16) + p % a.S) * 3 * HW);
#endif
        const unsigned src_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)srcb);
        const unsigned src_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(srcb >> 32));

        const float* const hmkt_pair = a.hmkt + (size_t)p * 12;
#ifdef SWEEP_TRACE
        if (trace_first && tid == 0 && blockIdx.x < 1024) sweep_trace_first[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SWEEP_BOXTIME
        unsigned long long bt_foot = __builtin_amdgcn_s_memrealtime(), bt_stage = 0, bt_sweep = 0;
#endif
        if (tid < 64) {
            if (ocnt <= 8) sweep_footprints8(hmkt_pair, tx0, ty0, obeg, ocnt, W, H, D, idmin_f, idstep_f, parity);
            else sweep_footprints16(hmkt_pair, tx0, ty0, obeg, ocnt, W, H, D, idmin_f, idstep_f, parity);
        }
#ifdef SWEEP_TRACE
        if (trace_first && tid == 0 && blockIdx.x < 1024) sweep_trace_first[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();   // groups parked; every wave has left the previous tile (its box is free)
#ifdef SWEEP_TRACE
        if (trace_first && tid == 0 && blockIdx.x < 1024) sweep_trace_first[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SWEEP_TRACE
        const unsigned long long trace_t1 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SWEEP_BOXTIME
        bt_foot = __builtin_amdgcn_s_memrealtime() - bt_foot;
#endif
        const int level = __builtin_amdgcn_readfirstlane(hdr[2 * parity]);
        const int ngroups = (ocnt + (1 << level) - 1) >> level;
        const SweepTerms terms = sweep_load_terms(hmkt_pair);              // this tile's camera terms: SGPRs
        SWEEP_TERMS(terms);
        const float k2e = sweep_uniform(k2 + 1e-6f);

        // output: a raw-buffer descriptor over this pair's slice; per lane ONE byte offset (out of range for lanes outside
        // the image: their stores are dropped), the plane / channel-group stride advances in an SGPR
        const unsigned ounit = LAYOUT == 0 ? 4u : 16u;                       // bytes per pixel of one plane / channel group
        const size_t pair_floats = LAYOUT == 0 ? (size_t)D * HW : LAYOUT == 1 ? (size_t)(D / 4 + 1) * HW * 4 : (size_t)(D / 8 + 1) * HW * 4;
        const unsigned long long outb = reinterpret_cast<unsigned long long>(a.out + (size_t)p * pair_floats);
        const unsigned out_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)outb);
        const unsigned out_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(outb >> 32));
        const __amdgpu_buffer_rsrc_t orsrc = sweep_rsrc(out_lo, out_hi, (unsigned)(pair_floats * 4));
        const unsigned ovoff = pvalid ? (unsigned)(y * W + x) * ounit : 0xFFFFFFFFu;
        const unsigned ostride = (unsigned)HW * ounit;
        unsigned osoff = (unsigned)(LAYOUT == 0 ? 8 * obeg : LAYOUT == 1 ? 2 * obeg : obeg) * ostride;

        for (int g = 0; g < ngroups; ++g) {
            SweepBox bx;
            bx.rx0 = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][0]); bx.ry0 = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][1]);
            bx.rw = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][2]); bx.rh = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][3]);
            const bool staged = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][4]) != 0;
            const bool clamp = __builtin_amdgcn_readfirstlane(grp[parity * SWEEP_MAX_OCT + g][5]) != 0;   // box clipped by the image
#ifdef SWEEP_STATS
            if (tid == 0) {
                if (g == 0) atomicAdd(&sweep_stats[0], 1u);
                if (staged) { atomicAdd(&sweep_stats[1], 1u); atomicAdd(&sweep_stats[3], (unsigned)(bx.rw * bx.rh)); }
                else atomicAdd(&sweep_stats[2], (unsigned)(min((g + 1) << level, ocnt) - (g << level)));
            }
#endif
#ifdef SWEEP_BOXTIME
            const unsigned long long bt0 = __builtin_amdgcn_s_memrealtime();
#endif
            if (staged) sweep_stage_box(src_lo, src_hi, bx.rx0, bx.ry0, bx.rw, bx.rh, W, H, g > 0);
#ifdef SWEEP_BOXTIME
            const unsigned long long bt1 = __builtin_amdgcn_s_memrealtime();
            bt_stage += bt1 - bt0;
#endif
#ifdef SWEEP_TRACE
            if (trace_first && g == 0 && tid == 0 && blockIdx.x < 1024) sweep_trace_first[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();
#endif
            // per-pixel map terms and parallax-form constants (only used with staged boxes, i.e. when a2 is safely non-zero),
            // worked out per box and AFTER its staging: ~25 instructions per box instead of registers held across the call
            int xv = x, yv = y;
            asm volatile("" : "+v"(xv), "+v"(yv));
            const float fx_ = (float)xv, fy_ = (float)yv;
            float a2 = fmaf(h20, fx_, fmaf(h21, fy_, h22));
            const float a2s = fabsf(a2) >= 0.125f ? a2 : 1.f;
            float ra = __builtin_amdgcn_rcpf(a2s);
            ra = fmaf(fmaf(-a2s, ra, 1.0f), ra, ra);
            float pu, pv;
            {
                const float a0 = fmaf(h00, fx_, fmaf(h01, fy_, h02)), a1 = fmaf(h10, fx_, fmaf(h11, fy_, h12));
                pu = a0 * ra; pv = a1 * ra;
                pu = fmaf(fmaf(-a2s, pu, a0), ra, pu); pv = fmaf(fmaf(-a2s, pv, a1), ra, pv);   // correctly rounded quotients but for rare ties
            }
            const float pa = fmaf(-pu, k2e, k0), pb = fmaf(-pv, k2e, k1);
            float k2v = k2e;
            asm("" : "+v"(k2v));                                             // VALU operands from VGPRs: an SGPR source
            asm("" : "+v"(a2));                                              // costs the FMA its full issue rate on gfx950
            const float ug = pu - (0.5f + (float)bx.rx0), vg = pv - (0.5f + (float)bx.ry0);
            float umax = (float)(bx.rw - 2), vmax = (float)(bx.rh - 1);
            unsigned rwv = (unsigned)bx.rw;
            asm("" : "+v"(umax)); asm("" : "+v"(vmax)); asm("" : "+v"(rwv));
            const int o_end = obeg + min((g + 1) << level, ocnt);
            for (int o = obeg + (g << level); o < o_end; ++o) {
                const int d0 = o * 8;
                sw_f16x2 hh[4];
#if SWEEP_ROLL_QUADS
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int q = 0; q < 2; ++q) {
                    float cost[4];
                    if (staged) {
                        if (clamp) sweep_quad<true>(box, zsh + d0 + 4 * q, cost, ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, nr, ng, nb);
                        else sweep_quad<false>(box, zsh + d0 + 4 * q, cost, ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, nr, ng, nb);
                    } else {
                        const float4 c = sweep_quad_global(src_lo, src_hi, W, H, hmkt_pair, x, y, d0 + 4 * q, nr, ng, nb);
                        cost[0] = c.x; cost[1] = c.y; cost[2] = c.z; cost[3] = c.w;
                    }
                    if (LAYOUT == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (d0 + 4 * q + j < D) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cost[j]), orsrc, ovoff, osoff, AUX);
                            osoff += ostride;
                        }
                    } else if (LAYOUT == 1) {
#ifdef SWEEP_NOSTORE
                        if (cost[0] == 12345.678f)                               // debug builds: the sweep without its output stream
#endif
                        if (d0 + 4 * q < D) {
                            const sw_u32x4 v = {__float_as_uint(cost[0]), __float_as_uint(cost[1]), __float_as_uint(cost[2]), __float_as_uint(cost[3])};
#ifdef SWEEP_STORE_LOCAL
                            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovoff, 0, AUX);   // debug builds: every store lands in the pair's first channel group (cache-resident)
#else
                            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovoff, osoff, AUX);
#endif
                            // (a 16-byte store with an SGPR offset reads its data registers over several cycles and hipcc pads no hazard
                            // for that form -- conv_winograd4s.hip found the tail of such a store leaving with the NEXT values when the
                            // registers were rewritten at once; here the next write to them is a sample away, and the wait states cost
                            // this wave nothing that the other seven on its SIMD do not fill)
                            asm volatile("s_nop 7" ::: "memory");
                            osoff += ostride;
                        }
                    } else {
                        hh[2 * q] = sw_f16x2{(_Float16)cost[0], (_Float16)cost[1]};
                        hh[2 * q + 1] = sw_f16x2{(_Float16)cost[2], (_Float16)cost[3]};
                    }
                }
#ifdef SWEEP_TRACE
                if (trace_first && tid == 0 && blockIdx.x < 1024) { sweep_trace_first[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime(); trace_first = false; }
#endif
                if (LAYOUT == 2) {
                    const sw_u32x4 v = {__builtin_bit_cast(unsigned, hh[0]), __builtin_bit_cast(unsigned, hh[1]), __builtin_bit_cast(unsigned, hh[2]), __builtin_bit_cast(unsigned, hh[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovoff, osoff, AUX);
                    osoff += ostride;
                }
            }
#ifdef SWEEP_BOXTIME
            bt_sweep += __builtin_amdgcn_s_memrealtime() - bt1;
#endif
        }
#ifdef SWEEP_BOXTIME
        if (tid == 0) { atomicAdd(&sweep_boxtime[0], bt_foot); atomicAdd(&sweep_boxtime[1], bt_stage); atomicAdd(&sweep_boxtime[2], bt_sweep); atomicAdd(&sweep_boxtime[3], 1ull); }
#endif
        // osoff now points at the channel group behind the D planes: the reference image (depthNet_model.py:233)
        if (LAYOUT == 1 && obeg + ocnt == noct) {
            const sw_u32x4 v = {__float_as_uint(-nr), __float_as_uint(-ng), __float_as_uint(-nb), 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovoff, osoff, AUX);
        }
        if (LAYOUT == 2 && obeg + ocnt == noct) {
            const sw_f16x2 h0 = {(_Float16)(-nr), (_Float16)(-ng)}, h1 = {(_Float16)(-nb), (_Float16)0.f};
            const sw_u32x4 v = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovoff, osoff, AUX);
        }
#ifdef SWEEP_SPAN
        if (tid == 0 && blockIdx.x < 1024) { sweep_span[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime(); sweep_span[blockIdx.x][3] += 1;
                                             if (cur_tile_span < 4096) sweep_unit_ticks[cur_tile_span] = (unsigned)(sweep_span[blockIdx.x][2] - span_u0); }
#endif
#ifdef SWEEP_TRACE
        if (tid == 0) {
            const unsigned slot = atomicAdd(&sweep_trace_n, 1u);
            if (slot < 8192) {
                sweep_trace[slot][0] = ((unsigned long long)blockIdx.x << 32) | (unsigned)unit;
                sweep_trace[slot][1] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_ID, XCC_ID
                sweep_trace[slot][2] = trace_t0;
                sweep_trace[slot][3] = (__builtin_amdgcn_s_memrealtime() << 20) | ((trace_t1 - trace_t0) & 0xFFFFF);
            }
        }
#endif
        unit = __builtin_amdgcn_readfirstlane(hdr[2 * parity + 1]);          // (parked before this unit's first barrier; rewritten two units from now)
        cur = decode(min(unit, nunits - 1));
    }
    // the last workgroup to leave rearms the counters
    if (tid == 0 && a.queue && atomicAdd(a.queue + 1, 1u) == gridDim.x - 1) { a.queue[0] = 0u; a.queue[1] = 0u; }
}

// Scratch of the sweep: the tile queue of the persistent workgroups, ws[0] = tickets drawn, ws[1] = workgroups that
// have left.  Contract: the words are ZERO when a call starts and the call leaves them zero (the last workgroup to
// leave rearms them), so a workspace is zeroed once, when it is allocated.  ws == nullptr selects a fixed
// tile-to-workgroup stride instead (no scratch, slower when tiles differ in cost).
extern "C" size_t cnm_planesweep_workspace_floats(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
    return 4;
}

// workgroups the chip holds at once (SWEEP_WG_PER_CU per CU); one query per device
static int sweep_resident_workgroups() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
        cached[dev] = cus * SWEEP_WG_PER_CU;
    }
    return cached[dev];
}

// Measurement hook (bench.py): the next n plane-sweep launches of this process are each bracketed by a pair of HIP events recorded
// on the launch's own stream right around the kernel -- the launch as it runs INSIDE the caller's step, between its real
// neighbours.  The events are created WITHOUT the system-scope fence of a default hipEventRecord (hipEventDisableSystemFence): with
// it the closing event first writes the launch's 210 MB of output back to system scope, and the bracket measures that flush.
// [r6] The launch path looks at ONE relaxed atomic (armed launches left); everything else happens under the hook's mutex, and a launch
// under stream capture is never bracketed.
#include <vector>
#include <atomic>
#include <mutex>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
static std::mutex g_sweep_ev_mu;
static std::vector<hipEvent_t> g_sweep_ev;                              // 2 per armed launch
static int g_sweep_ev_next = 0;
static std::atomic<int> g_sweep_ev_left{0};
extern "C" int cnm_debug_sweep_timing_arm(int n) {
    std::lock_guard<std::mutex> lock(g_sweep_ev_mu);
    g_sweep_ev_left.store(0, std::memory_order_relaxed);
    for (hipEvent_t e : g_sweep_ev) (void)hipEventDestroy(e);
    g_sweep_ev.clear(); g_sweep_ev_next = 0;
    if (n <= 0) return CNM_OK;
    for (int i = 0; i < 2 * n; ++i) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); return CNM_ERR_LAUNCH; }
        g_sweep_ev.push_back(e);
    }
    g_sweep_ev_left.store(n, std::memory_order_relaxed);
    return CNM_OK;
}
// elapsed milliseconds of the armed launches recorded so far (waits for the last one); returns how many were written to ms[]
extern "C" int cnm_debug_sweep_timing_read(float* ms, int n) {
    std::lock_guard<std::mutex> lock(g_sweep_ev_mu);
    int k = 0;
    for (; k < n && k < g_sweep_ev_next; ++k) {
        if (hipEventSynchronize(g_sweep_ev[2 * k + 1]) != hipSuccess || hipEventElapsedTime(ms + k, g_sweep_ev[2 * k], g_sweep_ev[2 * k + 1]) != hipSuccess) { (void)hipGetLastError(); break; }
    }
    return k;
}
static void sweep_timing_events(hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {   // cold: only while a measurement is armed
    *e0 = *e1 = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_sweep_ev_mu);
    if (g_sweep_ev_left.load(std::memory_order_relaxed) <= 0 || 2 * g_sweep_ev_next + 1 >= (int)g_sweep_ev.size()) return;
    *e0 = g_sweep_ev[2 * g_sweep_ev_next]; *e1 = g_sweep_ev[2 * g_sweep_ev_next + 1]; ++g_sweep_ev_next;
    g_sweep_ev_left.fetch_sub(1, std::memory_order_relaxed);
}

// Cache policy of the output stores: a per-device DECISION [r6], not a side effect of launching.  The volume is written once (13 MB per
// pair); whether `nt` stores or plain ones are faster INSIDE a step depends on the box: the launch displaces the dirty lines its
// predecessors left in the memory-side cache (tools/k1_context_probe.py: 55 us into its own still-cached buffer, 63 us after any kernel
// that wrote 400 MB), and on the pool's boxes that costs nt 55-57 us against 60 us plain on some and 65 against 60 on others.  Round 5
// sampled the first 24 large launches of a process in place -- events and a process-global mutex in the launch path, the timed steps
// of a benchmark not the steady state, and launches under stream capture pinned to the default.  Now
//   * a launch reads ONE relaxed atomic per device: forced (cnm_tune_sweep_store, or CNM_SWEEP_STORE = plain | nt | 0 | 2 in the
//     environment, read once) > calibrated > the default (nt);
//   * cnm_calibrate_sweep_store(scratch, stream) is the measurement, explicit and blocking: 24 launches of the 16-pair 192 x 256 x 64
//     shape on caller-provided scratch, alternating policies, each behind a 400 MB fill (the predecessor that matters) and between two
//     fence-free events; the lower median becomes the device's policy.  The Python modules run it once per device when a depthNet
//     allocates its workspace (never under capture), i.e. before any graph is captured and before any timed region.
// Both policies write the same bytes (tests/test_gpu_parity.py).
struct SweepStoreState { std::atomic<int> forced{-1}, chosen{-1}; float median_us[2] = {0.f, 0.f}; };
static SweepStoreState g_sweep_store[64];
static std::mutex g_sweep_store_mu;                                     // calibration and the tuning knob only
static int sweep_store_env() {                                          // CNM_SWEEP_STORE, read once: -1 = not set
    static const int v = [] {
        const char* e = getenv("CNM_SWEEP_STORE");
        if (!e || !*e) return -1;
        if (!strcmp(e, "plain") || !strcmp(e, "0")) return 0;
        if (!strcmp(e, "nt") || !strcmp(e, "2")) return SWEEP_STORE_AUX;
        return -1;
    }();
    return v;
}
static int sweep_store_policy(bool beyond_cache) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return SWEEP_STORE_AUX; }
    const int f = g_sweep_store[dev].forced.load(std::memory_order_relaxed);
    if (f >= 0) return f;
    const int e = sweep_store_env();
    if (e >= 0) return e;
    if (beyond_cache) return SWEEP_STORE_AUX;
    const int c = g_sweep_store[dev].chosen.load(std::memory_order_relaxed);
    return c >= 0 ? c : SWEEP_STORE_AUX;
}
// n = 0 / SWEEP_STORE_AUX: force that policy on the current device; n = -1: drop the forced policy AND the calibration (back to the
// default until the next cnm_calibrate_sweep_store); anything else only queries.  Returns the policy launches on the current device
// use now: forced, from the environment, calibrated -- or -1 when none of these has decided (launches then use nt).  median_us (may be
// NULL): [plain, nt] of the last calibration.
extern "C" int cnm_tune_sweep_store(int n, float* median_us) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return -1; }
    std::lock_guard<std::mutex> lock(g_sweep_store_mu);
    SweepStoreState& t = g_sweep_store[dev];
    if (n == 0 || n == SWEEP_STORE_AUX) t.forced.store(n, std::memory_order_relaxed);
    else if (n == -1) { t.forced.store(-1, std::memory_order_relaxed); t.chosen.store(-1, std::memory_order_relaxed); t.median_us[0] = t.median_us[1] = 0.f; }
    if (median_us) { median_us[0] = t.median_us[0]; median_us[1] = t.median_us[1]; }
    const int f = t.forced.load(std::memory_order_relaxed);
    if (f >= 0) return f;
    if (sweep_store_env() >= 0) return sweep_store_env();
    return t.chosen.load(std::memory_order_relaxed);
}

// A caller that measured the launch inside ITS OWN step (cnmnet_amd/pipeline.py: both policies forced in turn, the launch timed by the
// hook above between its real neighbours) records the decision here: it replaces the scratch calibration's (more faithful: what the
// launch's stores meet depends on the kernels around it).  policy = 0 / 2; median_us (may be NULL) = [plain, nt] as measured.
extern "C" int cnm_decide_sweep_store(int policy, const float* median_us) {
    CNM_REQUIRE(policy == 0 || policy == SWEEP_STORE_AUX, CNM_ERR_BAD_ARG);
    int dev = 0;
    CNM_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, CNM_ERR_LAUNCH);
    std::lock_guard<std::mutex> lock(g_sweep_store_mu);
    SweepStoreState& t = g_sweep_store[dev];
    t.chosen.store(policy, std::memory_order_relaxed);
    if (median_us) { t.median_us[0] = median_us[0]; t.median_us[1] = median_us[1]; }
    return CNM_OK;
}

static int sweep_launch(int layout, const float* ref, const float* src, const float* hmkt, float* out,
                        float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                        double idepth_min, double idepth_max, void* stream, int force_policy = -1, long long ref_bstride = 0, long long src_bstride = 0) {
    CNM_REQUIRE(ref && src && hmkt && out, CNM_ERR_BAD_ARG);
    // frame strides: whole images (multiples of 3 H W), at most 65535 images apart
    const long long img = 3ll * H * W, rfs = ref_bstride ? ref_bstride / img : 1, sfs = src_bstride ? src_bstride / img : S;
    CNM_REQUIRE(rfs * img == (ref_bstride ? ref_bstride : img) && sfs * img == (src_bstride ? src_bstride : (long long)S * img) && rfs >= 1 && rfs < 65536 && sfs >= S && sfs < 65536, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)B * (rfs > sfs ? rfs : sfs) < (1ll << 24), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ws == nullptr || (((uintptr_t)ws & 15) == 0 && ws_floats >= 4), CNM_ERR_WORKSPACE);
    CNM_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= CNM_MAX_PLANES, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout == 0 || D % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout != 2 || D % 8 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)H * W * 12 < (1ll << 31), CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)H * W * 4 * (D + 8) < (1ll << 32) - 1, CNM_ERR_BAD_ARG);          // a pair's output slice sits behind one raw-buffer descriptor
    const long long ntiles = (long long)cnm_ceil_div(W, SWEEP_TW) * cnm_ceil_div(H, SWEEP_TH) * B * S;
    CNM_REQUIRE(ntiles < (1ll << 20), CNM_ERR_BAD_ARG);                      // the kernel decodes tile ids with fp32 reciprocals
    SweepArgs a;
    a.ref = ref; a.src = src; a.hmkt = hmkt; a.out = out;
    a.frame_strides = (int)(((unsigned)sfs << 16) | (unsigned)rfs);
    a.queue = reinterpret_cast<unsigned int*>(ws);
    a.B = B; a.S = S; a.H = H; a.W = W; a.D = D;
    const double idstep = (idepth_max - idepth_min) / (D - 1.0);             // depthNet_model.py:194
    a.idmin = idepth_min; a.idstep = idstep;
#ifdef SWEEP_Z_ARGS
    for (int d = 0; d < CNM_MAX_PLANES; ++d) a.z[d] = d < D ? sweep_depth(idepth_min, idstep, d) : 0.f;
#endif
    a.idmin_f = (float)idepth_min; a.idstep_f = (float)idstep;              // fp32 depths are enough for a box with margins
    const int slots = sweep_resident_workgroups();
    CNM_REQUIRE(slots > 0, CNM_ERR_LAUNCH);
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
    // the queue ends with small units (zone sizes in sixteenths of the grid; a part never has less than one octet) - when there
    // is a queue to speak of: a launch that fits the chip at once sweeps whole tiles only (partial units get boxes of their own,
    // i.e. other box origins and last-bit differences in the sample coordinates: small launches stay independent of how the
    // caller batches its pairs, which the fp16 engine's tests ask for)
    const int nt = (int)ntiles, g = ntiles > slots ? (int)grid.x : 0;
    a.noct = (D + 7) >> 3; a.ntx = cnm_ceil_div(W, SWEEP_TW); a.tiles_per_pair = a.ntx * cnm_ceil_div(H, SWEEP_TH);
    a.ne = a.noct >= 8 ? std::min(nt, g * SWEEP_TAIL_EIGHTHS / 16) : 0;
    a.nq = a.noct >= 4 ? std::min(nt - a.ne, g * SWEEP_TAIL_QUARTERS / 16) : 0;
    a.nh = a.noct >= 2 ? std::min(nt - a.ne - a.nq, g * SWEEP_TAIL_HALVES / 16) : 0;
    a.nfull = nt - a.nh - a.nq - a.ne; a.nunits = a.nfull + 2 * a.nh + 4 * a.nq + 8 * a.ne;
    a.inv_tpp = 1.0f / (float)a.tiles_per_pair; a.inv_ntx = 1.0f / (float)a.ntx;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_sweep_ev_left.load(std::memory_order_relaxed) > 0) sweep_timing_events(cnm_stream(stream), &ev0, &ev1);   // bench.py's measurement hook, armed explicitly
    // the device's policy decides for outputs that fit the 256 MB memory-side cache (what it was measured on); a larger volume (config 4: 2 GB per
    // launch) streams to HBM whatever surrounds the launch, and there non-temporal stores win by 10 % on every box seen (503 against 554-566 us,
    // profiles/r6_k1_store_aux.txt) -- unless a policy is FORCED (cnm_tune_sweep_store / CNM_SWEEP_STORE)
    const size_t out_bytes = (size_t)B * S * H * W * (layout == 2 ? 2 : 4) * (size_t)(D + (layout == 0 ? 0 : layout == 1 ? 4 : 8));
    const int aux = force_policy >= 0 ? force_policy : sweep_store_policy(out_bytes > ((size_t)256 << 20));
    if (ev0) (void)hipEventRecord(ev0, cnm_stream(stream));
    if (aux == 0) {
        if (layout == 0) planesweep_kernel<0, 0><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
        else if (layout == 1) planesweep_kernel<1, 0><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
        else planesweep_kernel<2, 0><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
    } else {
        if (layout == 0) planesweep_kernel<0, SWEEP_STORE_AUX><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
        else if (layout == 1) planesweep_kernel<1, SWEEP_STORE_AUX><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
        else planesweep_kernel<2, SWEEP_STORE_AUX><<<grid, SWEEP_NT, 0, cnm_stream(stream)>>>(a);
    }
    if (ev1) (void)hipEventRecord(ev1, cnm_stream(stream));
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// The calibration (see above).  scratch: cnm_calibrate_sweep_store_floats() floats of device memory, 16-byte aligned, contents
// irrelevant.  Blocking (synchronises `stream`), not allowed under stream capture.  Returns the chosen policy (0 / 2) or a negative status.
static constexpr int kCalB = 8, kCalS = 2, kCalH = 192, kCalW = 256, kCalD = 64, kCalN = 24;
static constexpr size_t kCalOut = (size_t)kCalB * kCalS * (kCalD / 4 + 1) * kCalH * kCalW * 4, kCalRef = (size_t)kCalB * 3 * kCalH * kCalW,
                        kCalSrc = kCalRef * kCalS, kCalDirty = (size_t)100 << 20, kCalSmall = 256;   // floats; small = tile queue (4) + 16 pairs x 12 terms
extern "C" size_t cnm_calibrate_sweep_store_floats(void) { return kCalOut + kCalRef + kCalSrc + kCalDirty + kCalSmall; }
extern "C" int cnm_calibrate_sweep_store(float* scratch, size_t scratch_floats, void* stream, float* median_us) {
    CNM_REQUIRE(scratch && ((uintptr_t)scratch & 15) == 0 && scratch_floats >= cnm_calibrate_sweep_store_floats(), CNM_ERR_WORKSPACE);
    hipStream_t s = cnm_stream(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return CNM_ERR_BAD_ARG; }
    int dev = 0;
    CNM_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, CNM_ERR_LAUNCH);
    std::lock_guard<std::mutex> lock(g_sweep_store_mu);
    float* out = scratch; float* ref = out + kCalOut; float* src = ref + kCalRef; float* dirty = src + kCalSrc; float* small = dirty + kCalDirty;
    float hm[kCalB * kCalS * 12];                                        // identity homography, 0.1 m baseline at f = 288: u' = x + 28.8 / z
    for (int p = 0; p < kCalB * kCalS; ++p) { const float t[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, (p & 1) ? -28.8f : 28.8f, 0, 0}; memcpy(hm + 12 * p, t, sizeof(t)); }
    bool ok = hipMemsetAsync(ref, 0, (kCalRef + kCalSrc) * 4, s) == hipSuccess && hipMemsetAsync(small, 0, kCalSmall * 4, s) == hipSuccess &&
              hipMemcpyAsync(small + 16, hm, sizeof(hm), hipMemcpyHostToDevice, s) == hipSuccess;
    hipEvent_t ev[kCalN][2] = {};
    for (int i = 0; ok && i < kCalN; ++i)
        for (int j = 0; ok && j < 2; ++j) ok = hipEventCreateWithFlags(&ev[i][j], hipEventDisableSystemFence) == hipSuccess;
    int rc = CNM_OK;
    for (int i = 0; ok && rc == CNM_OK && i < kCalN + 2; ++i) {         // two untimed launches first
        const int k = i - 2;
        ok = hipMemsetAsync(dirty, i & 0xFF, kCalDirty * 4, s) == hipSuccess;   // the predecessor that matters: 400 MB of dirty lines in the memory-side cache
        if (ok && k >= 0) ok = hipEventRecord(ev[k][0], s) == hipSuccess;
        if (ok) rc = sweep_launch(1, ref, src, small + 16, out, small, 4, kCalB, kCalS, kCalH, kCalW, kCalD, 0.1, 3.0, stream, (i & 1) ? SWEEP_STORE_AUX : 0);
        if (ok && rc == CNM_OK && k >= 0) ok = hipEventRecord(ev[k][1], s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    int chosen = -1;
    if (ok && rc == CNM_OK) {
        float v[2][kCalN / 2]; int n[2] = {0, 0};
        for (int k = 0; k < kCalN; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[k][0], ev[k][1]) == hipSuccess) v[k & 1][n[k & 1]++] = ms * 1e3f; else (void)hipGetLastError();
        }
        SweepStoreState& t = g_sweep_store[dev];
        for (int k = 0; k < 2; ++k) { std::sort(v[k], v[k] + n[k]); t.median_us[k] = n[k] ? v[k][n[k] / 2] : 0.f; }
        if (n[0] && n[1]) { chosen = t.median_us[0] < t.median_us[1] ? 0 : SWEEP_STORE_AUX; t.chosen.store(chosen, std::memory_order_relaxed); }
        if (median_us) { median_us[0] = t.median_us[0]; median_us[1] = t.median_us[1]; }
    }
    for (int i = 0; i < kCalN; ++i) for (int j = 0; j < 2; ++j) if (ev[i][j]) (void)hipEventDestroy(ev[i][j]);
    if (!ok) { (void)hipGetLastError(); return CNM_ERR_LAUNCH; }
    return rc != CNM_OK ? rc : (chosen >= 0 ? chosen : CNM_ERR_LAUNCH);
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
// [r6] the same two with the images as VIEWS: ref_bstride / src_bstride = floats between consecutive frames (0 = dense) -- ref = frames[:, 0] and
// src = frames[:, 1:] of one [B][1 + S][3][H][W] tensor are read where they lie.  The strides must be whole images (multiples of 3 H W), < 65536 images
extern "C" int cnm_planesweep_cat_strided_c4_f32(const float* ref, long long ref_bstride, const float* src, long long src_bstride, const float* hmkt, float* x,
                                                 float* ws, size_t ws_floats, int B, int S, int H, int W, int D, double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(1, ref, src, hmkt, x, ws, ws_floats, B, S, H, W, D, idepth_min, idepth_max, stream, -1, ref_bstride, src_bstride);
}
extern "C" int cnm_planesweep_cat_strided_c8_f16(const float* ref, long long ref_bstride, const float* src, long long src_bstride, const float* hmkt, void* x,
                                                 float* ws, size_t ws_floats, int B, int S, int H, int W, int D, double idepth_min, double idepth_max, void* stream) {
    return sweep_launch(2, ref, src, hmkt, static_cast<float*>(x), ws, ws_floats, B, S, H, W, D, idepth_min, idepth_max, stream, -1, ref_bstride, src_bstride);
}
