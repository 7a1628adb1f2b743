// K0 + K1: camera prep and the fused plane-sweep warp + L1 cost volume.
//
// K0 replaces get_pixel_coordinates + process_camera_parameters
//    (reference depthnet/depth_util.py:13-56): 12 floats per (ref,src) pair; the pixel
//    grid is implicit in the thread index and KRKiUV [B,3,H*W] is never materialised.
// K1 replaces depthNet.getVolume (reference depthnet/depthNet_model.py:185-224), i.e.
//    64 x ~12 ATen launches per pair, with ONE launch over all pairs, planes and pixels.
//
// K1 = ONE launch on the caller's stream (gfx950, 64-lane waves), persistent workgroups:
//   The grid is sized to the chip (one 16-wave workgroup per CU; round 2: two of 8 waves).  A workgroup sweeps 64 x 16 pixel tiles drawn
//   from a ticket counter (the last tiles are handed out as two half sweeps so the launch has no tile-long
//   tail); one lane per reference pixel walks the planes, with the per-pixel camera terms and the reference RGB
//   in registers; the next tile's camera terms / reference pixel / ticket travel while the current one is swept.
//   Per tile, wave 0 projects the 4 tile corners on the 2 end planes of every 8-plane octet (projective map =>
//   extremes there): the footprint of the tile in the source image per octet, merged (DPP) into runs of 2, 4, 8,
//   16 octets; the longest run whose footprint fits the LDS budget is staged ONCE into LDS as pre-differenced
//   texels: 12 floats (P, dP/dx, dP/dy, d2P/dxdy per channel, zero outside the image) = three 16-byte units,
//   built from the planar source by range-checked buffer loads (6 per texel, the x+1 neighbour comes from the
//   next lane by DPP; no texture pre-pass).  A bilinear sample is then P + wu*dx + wv*dy + wu*wv*dxy: three
//   ds_read_b128 of ONE texel and 9 FMAs for the three channels instead of four taps and 12 weighted products.
//   Coordinates use the parallax form u' = a0/a2 + (k0 - (a0/a2) k2) / (a2 z + k2): one v_rcp_f32 and two FMAs
//   per plane (the reciprocal's rounding is scaled by the parallax, not by the coordinate).  The kernel is
//   VALU-issue bound on gfx950 (a wave64 fp32 op issues in ~3 cycles, conversions / med3 / 24-bit integer
//   multiplies in ~4.5, v_rcp_f32 in ~8.5, anything with an SGPR source in ~4.8: tools/valu_forms.hip), so the
//   design minimises issued instructions: all hot operands in VGPRs, scalar (not packed) FMAs, serial work on
//   one wave.  Octets whose footprint does not fit (extreme geometry, points behind the source camera, a2 near
//   zero) gather the same texels from global memory with the general division form.  Output leaves the registers
//   as coalesced stores: float4 (4 planes of one pixel) in the c4 layout, one float per plane for NCHW.
// HBM-bound by design: algorithmic bytes per pair = 3HW*4 (ref) + 3HW*4 (src) + D*HW*4 (volume)
// (+ 4HW*4 for the ref group when emitting the concatenated conv input).
#include "cnm_common.h"

#define CNM_MAX_PLANES 128
#define SWEEP_TW 64                 // tile width  (pixels, lanes along x)
#ifndef SWEEP_TH
#define SWEEP_TH 16                 // tile height = waves per workgroup.  [r3] 16 (one 1024-thread workgroup and one 155 KB box per CU) instead of 8
#endif                              // (two workgroups, two 80 KB boxes): a staged texel serves twice the pixels and a box holds twice the plane range, so
                                    // a third fewer texels are staged per launch (3.0 M instead of 4.5 M) -- 80 -> 69.5 us in tools/k1_bench.hip, although nothing overlaps a staging any more
#define SWEEP_NT (SWEEP_TW * SWEEP_TH)
#ifndef SWEEP_CAP
#define SWEEP_CAP 3300              // texels per LDS box (3 x 16 B each: 158 400 B, one workgroup per CU)
#endif
#ifndef SWEEP_MINW
#define SWEEP_MINW 4                // waves per SIMD the register allocation must allow (2 workgroups x 8 waves per CU)
#endif
#ifndef SWEEP_TAIL_HALVES
#define SWEEP_TAIL_HALVES 4         // tiles handed out as two half sweeps, in quarters of the grid size
#endif
#ifndef SWEEP_TAIL_QUARTERS
#define SWEEP_TAIL_QUARTERS 0       // tiles handed out as four quarter sweeps (the last ones), in quarters of the grid size
#endif
#ifndef SWEEP_AHEAD
#define SWEEP_AHEAD 1               // samples whose texel reads are in flight ahead of the blend
#endif
#define SWEEP_MAX_OCT (CNM_MAX_PLANES / 8)

struct SweepArgs {
    const float* ref; const float* src; const float* hmkt; float* out;
    unsigned int* queue;            // [0] tile tickets, [1] workgroups that have left; zero between launches
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

// lane l reads lane l+1 / a lane of its quad, half row or row (DPP: no LDS traffic)
__device__ __forceinline__ float sweep_next_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));   // wave_shl:1
}
template <int CTRL> __device__ __forceinline__ float sweep_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ int sweep_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
#define SWEEP_DPP_XOR1 0xB1         // quad_perm:[1,0,3,2]
#define SWEEP_DPP_XOR2 0x4E         // quad_perm:[2,3,0,1]
#define SWEEP_DPP_HALF_MIRROR 0x141 // lane 7-l of the 8-lane half row
#define SWEEP_DPP_ROR8 0x128        // row_ror:8 = lane l^8 of the 16-lane row

// A pre-differenced texel of the zero-extended source image at (x, y): three 16-byte units
//   u0 = (Pr, Pg, Pb, dxr)  u1 = (dxg, dxb, dyr, dyg)  u2 = (dyb, dxyr, dxyg, dxyb)
//   dx = P(x+1,y) - P(x,y), dy = P(x,y+1) - P(x,y), dxy = (P(x+1,y+1) - P(x,y+1)) - dx; P = 0 outside the image,
// so grid_sample's per-corner zeros padding (depthNet_model.py:220) is carried by the data, and a bilinear
// sample at fractions (wu, wv) is P + wu dx + wv dy + wu wv dxy.
struct SweepTexel { float4 u0, u1, u2; };

__device__ __forceinline__ SweepTexel sweep_texel_pack(const float p00[3], const float p01[3], const float p10[3], const float p11[3]) {
    float dx[3], dy[3], dxy[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dx[c] = p01[c] - p00[c]; dy[c] = p10[c] - p00[c]; dxy[c] = (p11[c] - p10[c]) - dx[c]; }
    SweepTexel t;
    t.u0 = make_float4(p00[0], p00[1], p00[2], dx[0]);
    t.u1 = make_float4(dx[1], dx[2], dy[0], dy[1]);
    t.u2 = make_float4(dy[2], dxy[0], dxy[1], dxy[2]);
    return t;
}

// the same texel gathered from global memory (octets whose footprint does not fit the LDS box);
// out-of-image corners: buffer offset 0xFFFFFFFF is out of range and the load returns 0
__device__ __forceinline__ SweepTexel sweep_texel_global(__amdgpu_buffer_rsrc_t rsrc, int x, int y, int W, int H, unsigned chan_bytes) {
    const bool x0 = (unsigned)x < (unsigned)W, x1 = (unsigned)(x + 1) < (unsigned)W;
    const bool y0 = (unsigned)y < (unsigned)H, y1 = (unsigned)(y + 1) < (unsigned)H;
    const unsigned o = (unsigned)(y * W + x) * 4u, row = (unsigned)W * 4u;
    const unsigned a00 = (x0 && y0) ? o : 0xFFFFFFFFu, a01 = (x1 && y0) ? o + 4u : 0xFFFFFFFFu;
    const unsigned a10 = (x0 && y1) ? o + row : 0xFFFFFFFFu, a11 = (x1 && y1) ? o + row + 4u : 0xFFFFFFFFu;
    float p00[3], p01[3], p10[3], p11[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned so = c * chan_bytes;
        p00[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a00, so, 0));
        p01[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a01, so, 0));
        p10[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a10, so, 0));
        p11[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, a11, so, 0));
    }
    return sweep_texel_pack(p00, p01, p10, p11);
}

// box: rx0, ry0 = image coordinates of box texel (0,0); rw x rh texels.  The box is the tile's footprint clipped to
// [-2, W] x [-2, H]; columns -2 and W (rows -2 and H) hold all-zero texels, so clamping a sample's coordinates
// into the box reproduces "no contribution" for everything outside the image.
struct SweepBox { int rx0, ry0, rw, rh; };
#ifdef SWEEP_TIMELINE
// debug builds (tools/k1_bench.hip -DSWEEP_TIMELINE): s_memtime of wave 0 / wave 15 of workgroup 0 at the stations of its first tiles
__device__ unsigned long long sweep_tl[2][64];
#define SWEEP_TL(i) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == SWEEP_TH - 1) && tlp + (i) < 64) sweep_tl[wave ? 1 : 0][tlp + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SWEEP_TL(i) do { } while (0)
#endif
#ifdef SWEEP_EMU
// experiment builds (tools/k1_bench.hip -DSWEEP_EMU -DSWEEP_SPAN): what a footprint PRE-PASS and a cost-ordered queue would buy, emulated -- a
// recording launch stores every tile's box groups; replaying launches read them back instead of working them out (mode bit 0) and / or
// draw the tiles in a given order, e.g. longest first from the recorded durations (mode bit 1).  Whole tiles only.
__device__ int sweep_emu_grp[4096][SWEEP_MAX_OCT][8];
__device__ int sweep_emu_level[4096];
__device__ int sweep_emu_order[4096];
__device__ int sweep_emu_mode;
#endif
#ifdef SWEEP_SPAN
// debug builds (tools/k1_bench.hip -DSWEEP_SPAN): per workgroup, s_memtime when its first tile starts and when its last tile ends (wave 0), tiles done
__device__ unsigned int sweep_unit_ticks[4096];   // duration of every unit (tile), 100 MHz ticks
__device__ unsigned long long sweep_span[1024][4];   // s_memrealtime (100 MHz, one base for the device): kernel entry, first tile start, last tile end; tiles
#endif
#ifdef SWEEP_STATS
__device__ unsigned int sweep_stats[4];     // debug builds only: workgroups, staged boxes, octets gathered from global, box texels
#endif

struct SweepCoord { unsigned xi, yi; float wu, wv; };

template <bool CLAMP>
__device__ __forceinline__ SweepCoord sweep_split(float iu, float iv, float umax, float vmax) {
    if (CLAMP) { iu = __builtin_amdgcn_fmed3f(iu, 0.f, umax); iv = __builtin_amdgcn_fmed3f(iv, 0.f, vmax); }
    SweepCoord c;
    c.xi = (unsigned)iu; c.yi = (unsigned)iv;                                // floor (coordinates are >= 0 here)
    c.wu = __builtin_amdgcn_fractf(iu); c.wv = __builtin_amdgcn_fractf(iv);
    return c;
}

// General form, depthNet_model.py:210-213: (u', v') = (a0 z + k0, a1 z + k1) / (a2 z + k2 + 1e-6), minus the half
// pixel of grid_sample's unnormalisation and the box origin (cu, cv).
__device__ __forceinline__ SweepCoord sweep_coords(float cu, float cv, float umax, float vmax, float a0, float a1, float a2,
                                                   float k0, float k1, float k2e, float z) {
    const float den = fmaf(a2, z, k2e);
    float r = __builtin_amdgcn_rcpf(den);
    r = fmaf(fmaf(-den, r, 1.0f), r, r);                                     // one Newton step: ~0.5 ulp reciprocal
    return sweep_split<true>(fmaf(fmaf(a0, z, k0), r, cu), fmaf(fmaf(a1, z, k1), r, cv), umax, vmax);
}

// Parallax form of the same map: u' = a0/a2 + (k0 - (a0/a2) k2e) / (a2 z + k2e) = U + A r.  U (the image of the point
// at infinity) and A are per-pixel constants, so a plane costs one reciprocal and two FMAs, and the reciprocal's
// rounding is scaled by the parallax |A r| instead of the coordinate |u'|: the plain v_rcp_f32 (1 ulp) is enough.
// CLAMP = false for boxes that were not clipped by the image: every sample of a valid pixel then lies inside the box
// (margins included) and the two v_med3_f32 are saved; a lane of a ragged tile outside the image may then compute
// any address - LDS reads beyond the allocation return 0 and its result is never stored.
template <bool CLAMP>
__device__ __forceinline__ SweepCoord sweep_coords_parallax(float ug, float vg, float umax, float vmax, float pa, float pb,
                                                            float a2, float k2e, float z) {
    const float r = __builtin_amdgcn_rcpf(fmaf(a2, z, k2e));
    return sweep_split<CLAMP>(fmaf(pa, r, ug), fmaf(pb, r, vg), umax, vmax);
}

__device__ __forceinline__ float sweep_blend(const float4 u0, const float4 u1, const float4 u2, float wu, float wv,
                                             float rr, float rg, float rb) {
    // P + wu dx + wv (dy + wu dxy): three FMAs per channel
    const float er = fmaf(wv, fmaf(wu, u2.y, u1.z), fmaf(wu, u0.w, u0.x)) - rr;
    const float eg = fmaf(wv, fmaf(wu, u2.z, u1.w), fmaf(wu, u1.x, u0.y)) - rg;
    const float eb = fmaf(wv, fmaf(wu, u2.w, u2.x), fmaf(wu, u1.y, u0.z)) - rb;
    return (__builtin_fabsf(er) + __builtin_fabsf(eg)) + __builtin_fabsf(eb);   // :222-223
}

// Persistent workgroups: the grid is sized to the chip (one 16-wave workgroup per CU) and a workgroup sweeps tile
// blockIdx.x, then tiles drawn from a ticket counter in the caller's workspace (ws[0]: tickets, ws[1]: exits; both
// are zero between launches - the last workgroup to leave resets them).  Per tile: footprints (wave 0) ->
// [stage box -> sweep its planes]*.  The ticket, the camera terms and the reference pixel of the NEXT tile are
// fetched while the current one is swept.
// Eight planes of one pixel from a staged box: the texel reads of a sample are issued SWEEP_AHEAD samples before the
// blend that consumes them (counted lgkmcnt waits, one scheduling region per sample).
template <bool CLAMP>
__device__ __forceinline__ void sweep_octet(const float4* __restrict__ box, const float* __restrict__ zs, float (&cost)[8],
                                            float ug, float vg, float umax, float vmax, unsigned rwv, float pa, float pb,
                                            float a2, float k2v, float rr, float rg, float rb) {
    SweepCoord cd[8];
    float4 tx[8][3];
    float zz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) zz[j] = zs[j];                               // broadcast reads
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8 + SWEEP_AHEAD; ++j) {
        if (j < 8) {
            cd[j] = sweep_coords_parallax<CLAMP>(ug, vg, umax, vmax, pa, pb, a2, k2v, zz[j]);
            unsigned off;                                                    // byte offset of texel (yi, xi) in the box
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(cd[j].yi), "v"(rwv), "v"(cd[j].xi));
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
}

template <int LAYOUT>   // 0: volume [P,D,H,W] fp32   1: c4 [P,D/4+1,H,W,4] fp32   2: c8 [P,D/8+1,H,W,8] fp16
__global__ __launch_bounds__(SWEEP_NT) __attribute__((amdgpu_waves_per_eu(SWEEP_MINW, SWEEP_MINW))) void planesweep_kernel(const SweepArgs a) {
    // one LDS object: texel box | plane depths | plane groups (two tile parities) | header (level, next tile) x 2
    __shared__ float4 smem[3 * SWEEP_CAP + CNM_MAX_PLANES / 4 + 2 * 2 * SWEEP_MAX_OCT + 1];
    float4* const box = smem;
    float* const zsh = reinterpret_cast<float*>(smem + 3 * SWEEP_CAP);
    int (*const grp)[8] = reinterpret_cast<int (*)[8]>(smem + 3 * SWEEP_CAP + CNM_MAX_PLANES / 4);
    int* const hdr = reinterpret_cast<int*>(smem + 3 * SWEEP_CAP + CNM_MAX_PLANES / 4 + 4 * SWEEP_MAX_OCT);

#ifdef SWEEP_SPAN
    const unsigned long long span_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, HW = H * W, D = a.D;
    const int noct = (D + 7) >> 3;
    const int ntx = (W + SWEEP_TW - 1) / SWEEP_TW, nty = (H + SWEEP_TH - 1) / SWEEP_TH, tiles_per_pair = ntx * nty;
    const int ntiles = a.B * a.S * tiles_per_pair;
    const unsigned chan_bytes = (unsigned)HW * 4u;
    const size_t plane = (size_t)HW * (LAYOUT == 0 ? 1 : 4);                 // floats per plane / per 16-byte channel group
    if (tid < CNM_MAX_PLANES) zsh[tid] = tid < D ? sweep_depth(a, tid) : 0.f;

    // Work units: whole tiles first, then tiles cut into two half sweeps, then tiles cut into quarter sweeps: the
    // launch ends with small units, not with a tile-long tail of half-empty CUs.  Zone sizes in quarters of the grid.
    const int nq = noct >= 4 ? min(ntiles, (int)gridDim.x * SWEEP_TAIL_QUARTERS / 4) : 0;
    // (no half sweeps when the tiles divide evenly among the workgroups: 768 tiles on 256 CUs ran 72 us with them, 69.5 without)
    const int nh = (noct >= 2 && ntiles % (int)gridDim.x != 0) ? min(ntiles - nq, (int)gridDim.x * SWEEP_TAIL_HALVES / 4) : 0;
    const int nfull = ntiles - nh - nq, nunits = nfull + 2 * nh + 4 * nq;
    const float inv_tpp = 1.0f / (float)tiles_per_pair, inv_ntx = 1.0f / (float)ntx;
#if defined(SWEEP_EMU) || defined(SWEEP_SPAN)
    struct Unit { int p, tx0, ty0, obeg, ocnt, tile; };
#else
    struct Unit { int p, tx0, ty0, obeg, ocnt; };
#endif
    auto decode = [&](int u) {                                               // unit -> pair, tile origin, octet range
        Unit q; q.obeg = 0; q.ocnt = noct;
        int t = u;
        if (u >= nfull) {
            int v = u - nfull, parts = 2, part;
            if (v < 2 * nh) { t = nfull + (v >> 1); part = v & 1; }
            else { v -= 2 * nh; parts = 4; t = nfull + nh + (v >> 2); part = v & 3; }
            q.obeg = part * noct / parts;
            q.ocnt = (part + 1) * noct / parts - q.obeg;
        }
#ifdef SWEEP_EMU
        if ((sweep_emu_mode & 2) && t < 4096) t = sweep_emu_order[t];
#endif
#if defined(SWEEP_EMU) || defined(SWEEP_SPAN)
        q.tile = t;
#endif
        // fp32 reciprocals (exact for t < 2^20, checked by the launcher); back to SGPRs so the rest is scalar arithmetic
        q.p = __builtin_amdgcn_readfirstlane((int)(((float)t + 0.5f) * inv_tpp));
        const int rem = t - q.p * tiles_per_pair, tyi = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * inv_ntx));
        q.tx0 = (rem - tyi * ntx) * SWEEP_TW; q.ty0 = tyi * SWEEP_TH;
        return q;
    };
    // camera terms (lane i < 12 holds term i) and reference pixel of a unit's tile
    auto tile_loads = [&](const Unit& q, float& hkv, float (&refv)[3]) {
        const int x = q.tx0 + lane, y = q.ty0 + wave;
        hkv = a.hmkt[(size_t)q.p * 12 + min(lane, 11)];
        refv[0] = refv[1] = refv[2] = 0.f;
        if (x < W && y < H) {
            const float* refp = a.ref + (size_t)(q.p / a.S) * 3 * HW + (size_t)y * W + x;
            refv[0] = refp[0]; refv[1] = refp[HW]; refv[2] = refp[2 * HW];
        }
    };
    int unit = blockIdx.x;
    float hkv = 0.f, refv[3] = {0.f, 0.f, 0.f};
    Unit cur = decode(min(unit, nunits - 1));
    if (unit < nunits) tile_loads(cur, hkv, refv);

#ifdef SWEEP_TIMELINE
    int tlp = 0;
#endif
#ifdef SWEEP_SPAN
    if (tid == 0 && blockIdx.x < 1024) { sweep_span[blockIdx.x][0] = span_entry; sweep_span[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); sweep_span[blockIdx.x][3] = 0; }
#endif
    for (int parity = 0; unit < nunits; parity ^= 1) {
        SWEEP_TL(0);                                                        // tile start
#ifdef SWEEP_SPAN
        const unsigned long long span_u0 = __builtin_amdgcn_s_memrealtime();
        const int cur_tile_span = cur.tile;
#endif
        // the ticket of the unit after this one travels while wave 0 works out the footprints
        int ticket = unit + (int)gridDim.x;                                 // without a queue: a fixed stride
        if (tid == 0 && a.queue) ticket = (int)gridDim.x + (int)atomicAdd(a.queue, 1u);
        const int p = cur.p, tx0 = cur.tx0, ty0 = cur.ty0, obeg = cur.obeg, ocnt = cur.ocnt;
        const int x = tx0 + lane, y = ty0 + wave;
        const bool pvalid = x < W && y < H;
        float hq[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) hq[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hkv), i));
        const float h00 = hq[0], h01 = hq[1], h02 = hq[2], h10 = hq[3], h11 = hq[4], h12 = hq[5];
        const float h20 = hq[6], h21 = hq[7], h22 = hq[8], k0 = hq[9], k1 = hq[10], k2 = hq[11];
        const float k2e = k2 + 1e-6f;
        const float rr = refv[0], rg = refv[1], rb = refv[2];
        const float fx_ = (float)x, fy_ = (float)y;
        const float a0 = fmaf(h00, fx_, fmaf(h01, fy_, h02)), a1 = fmaf(h10, fx_, fmaf(h11, fy_, h12));
        float a2 = fmaf(h20, fx_, fmaf(h21, fy_, h22));
        const unsigned long long srcb = reinterpret_cast<unsigned long long>(a.src + (size_t)p * 3 * HW);
        const unsigned src_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)srcb);   // descriptor pinned to SGPRs
        const unsigned src_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(srcb >> 32));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>((unsigned long long)src_lo | ((unsigned long long)src_hi << 32)), 0, 3 * chan_bytes, 0x00020000);

        // ---- footprints, wave 0 only (serial work stays on one wave: the kernel is VALU-throughput bound and the CU's
        // other workgroup fills the gap).  Lane 8j + c of pass q projects tile corner (c & 3) on the first (c < 4) / last
        // plane of the unit's octet 8q + j; an 8-lane min / max (DPP) gives the octet's box in all eight lanes.  Merging
        // with the lanes 8, 16, 32 away and with the other pass gives the boxes of every aligned run of 2, 4, 8, 16
        // octets; the longest run whose boxes all fit the LDS budget wins.  Results go to LDS (groups, level, ticket).
#ifdef SWEEP_EMU
        const bool emu_replay = (sweep_emu_mode & 1) && cur.tile < 4096;
        if (tid < 64 && emu_replay) {                                        // the groups come from the table
            const int lv = sweep_emu_level[cur.tile];
            for (int i = lane; i < SWEEP_MAX_OCT * 8; i += 64) grp[parity * SWEEP_MAX_OCT + (i >> 3)][i & 7] = sweep_emu_grp[cur.tile][i >> 3][i & 7];
            if (lane == 0) { hdr[2 * parity] = lv; hdr[2 * parity + 1] = ticket; }
        }
        if (tid < 64 && !emu_replay) {
#else
        if (tid < 64) {
#endif
            const int cxi = (lane & 1) ? min(tx0 + SWEEP_TW - 1, W - 1) : tx0;
            const int cyi = (lane & 2) ? min(ty0 + SWEEP_TH - 1, H - 1) : ty0;
            const float cxf = (float)cxi, cyf = (float)cyi;
            const float ca0 = fmaf(h00, cxf, fmaf(h01, cyf, h02));
            const float ca1 = fmaf(h10, cxf, fmaf(h11, cyf, h12));
            const float ca2 = fmaf(h20, cxf, fmaf(h21, cyf, h22));
            // the parallax form needs a2 (linear over the tile: extremes at the corners) away from zero, one sign
            const bool parallax_ok = __ballot(!(fabsf(ca2) >= 0.25f)) == 0 && (__ballot(ca2 < 0.f) == 0 || __ballot(ca2 > 0.f) == 0);
            const float idmin = (float)a.idmin, idstep = (float)a.idstep;       // fp32 depths are enough for a box with margins
            int bx0[2], by0[2], bx1[2], by1[2], bok[2], bcl[2];              // box, footprint usable, box clipped by the image
            bool live[2];
            bx0[1] = by0[1] = 1 << 28; bx1[1] = by1[1] = -(1 << 28); bok[1] = 1; bcl[1] = 0; live[1] = false;   // neutral second pass
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && ocnt <= 8) break;
                const int oct = q * 8 + (lane >> 3);                         // octet of this unit
                live[q] = oct < ocnt;
                const int o = obeg + min(oct, ocnt - 1);
                const int d0 = o * 8, d1 = min(d0 + 8, D) - 1;
                const float zc = __builtin_amdgcn_rcpf(fmaf((float)((lane & 4) ? d1 : d0), idstep, idmin));
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
                // [-2, W] x [-2, H] (the outermost column / row of that range is all zeros)
                bx0[q] = (int)fminf(fmaxf(floorf(umin - 0.5f) - 1.f, -2.f), (float)W);
                bx1[q] = (int)fminf(fmaxf(floorf(umax - 0.5f) + 1.f, (float)bx0[q]), (float)W);
                by0[q] = (int)fminf(fmaxf(floorf(vmin - 0.5f) - 1.f, -2.f), (float)H);
                by1[q] = (int)fminf(fmaxf(floorf(vmax - 0.5f) + 1.f, (float)by0[q]), (float)H);
                bok[q] = okc;
                bcl[q] = !(floorf(umin - 0.5f) - 1.f >= -2.f && floorf(umax - 0.5f) + 1.f <= (float)W &&
                           floorf(vmin - 0.5f) - 1.f >= -2.f && floorf(vmax - 0.5f) + 1.f <= (float)H);
                if (!live[q]) { bx0[q] = 1 << 28; by0[q] = 1 << 28; bx1[q] = -(1 << 28); by1[q] = -(1 << 28); bok[q] = 1; bcl[q] = 0; }   // neutral
            }
            int level = 0, gx0[2], gy0[2], gx1[2], gy1[2], gst[2], gcl[2];
#pragma unroll
            for (int L = 0; L < 5; ++L) {
                if (L == 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        bx0[q] = min(bx0[q], sweep_dpp<SWEEP_DPP_ROR8>(bx0[q])); by0[q] = min(by0[q], sweep_dpp<SWEEP_DPP_ROR8>(by0[q]));
                        bx1[q] = max(bx1[q], sweep_dpp<SWEEP_DPP_ROR8>(bx1[q])); by1[q] = max(by1[q], sweep_dpp<SWEEP_DPP_ROR8>(by1[q]));
                        bok[q] &= sweep_dpp<SWEEP_DPP_ROR8>(bok[q]); bcl[q] |= sweep_dpp<SWEEP_DPP_ROR8>(bcl[q]);
                    }
                } else if (L == 2 || L == 3) {
                    const int m = L == 2 ? 16 : 32;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        bx0[q] = min(bx0[q], __shfl_xor(bx0[q], m)); by0[q] = min(by0[q], __shfl_xor(by0[q], m));
                        bx1[q] = max(bx1[q], __shfl_xor(bx1[q], m)); by1[q] = max(by1[q], __shfl_xor(by1[q], m));
                        bok[q] &= __shfl_xor(bok[q], m); bcl[q] |= __shfl_xor(bcl[q], m);
                    }
                } else if (L == 4) {
                    bx0[0] = bx0[1] = min(bx0[0], bx0[1]); by0[0] = by0[1] = min(by0[0], by0[1]);
                    bx1[0] = bx1[1] = max(bx1[0], bx1[1]); by1[0] = by1[1] = max(by1[0], by1[1]);
                    bok[0] = bok[1] = bok[0] & bok[1]; bcl[0] = bcl[1] = bcl[0] | bcl[1];
                }
                bool bad = false;
                int fits[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int rw = bx1[q] - bx0[q] + 1, rh = by1[q] - by0[q] + 1;
                    // capacity in texels, and in staging items ((rw + 1) rh over four passes of SWEEP_TH x 63 lanes)
                    const int rwc = min(max(rw, 0), SWEEP_CAP + 1), rhc = min(max(rh, 0), SWEEP_CAP + 1);   // 24-bit products
                    fits[q] = bok[q] && __mul24(rwc, rhc) <= SWEEP_CAP && __mul24(rwc + 1, rhc) <= 4 * SWEEP_TH * 63;
                    const bool run_live = ((q * 8 + (lane >> 3)) & ~((1 << L) - 1)) < ocnt;
                    bad |= run_live && !fits[q];
                }
                const bool all_fit = __ballot(bad) == 0;
                if (L == 0 || all_fit) {                                          // monotone: a run that fits implies its halves fit
                    level = L;
#pragma unroll
                    for (int q = 0; q < 2; ++q) { gx0[q] = bx0[q]; gy0[q] = by0[q]; gx1[q] = bx1[q]; gy1[q] = by1[q]; gst[q] = fits[q]; gcl[q] = bcl[q]; }
                }
            }
            level = __builtin_amdgcn_readfirstlane(level);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oct = q * 8 + (lane >> 3);
                if ((lane & 7) == 0 && oct < ocnt && (oct & ((1 << level) - 1)) == 0) {
                    int* gq = grp[parity * SWEEP_MAX_OCT + (oct >> level)];
                    gq[0] = gx0[q]; gq[1] = gy0[q]; gq[2] = gx1[q] - gx0[q] + 1; gq[3] = gy1[q] - gy0[q] + 1;
                    gq[4] = gst[q] && parallax_ok; gq[5] = gcl[q];
#ifdef SWEEP_EMU
                    if (cur.tile < 4096) for (int i = 0; i < 6; ++i) sweep_emu_grp[cur.tile][oct >> level][i] = gq[i];
#endif
                }
            }
#ifdef SWEEP_EMU
            if (lane == 0 && cur.tile < 4096) sweep_emu_level[cur.tile] = level;
#endif
            if (lane == 0) { hdr[2 * parity] = level; hdr[2 * parity + 1] = ticket; }
        }
        SWEEP_TL(1);                                                        // footprints done (wave 0) / waiting (others)
        __syncthreads();   // groups parked; every wave has left the previous tile (its box is free)
        SWEEP_TL(2);
        const int level = __builtin_amdgcn_readfirstlane(hdr[2 * parity]);
        const int next_unit = __builtin_amdgcn_readfirstlane(hdr[2 * parity + 1]);
        const int ngroups = (ocnt + (1 << level) - 1) >> level;
        cur = decode(min(next_unit, nunits - 1));
        if (next_unit < nunits) tile_loads(cur, hkv, refv);

        // parallax-form constants of this pixel (only used with staged boxes, i.e. when a2 is safely non-zero)
        const float a2s = fabsf(a2) >= 0.125f ? a2 : 1.f;
        float ra = __builtin_amdgcn_rcpf(a2s);
        ra = fmaf(fmaf(-a2s, ra, 1.0f), ra, ra);
        float pu = a0 * ra, pv = a1 * ra;
        pu = fmaf(fmaf(-a2s, pu, a0), ra, pu); pv = fmaf(fmaf(-a2s, pv, a1), ra, pv);   // correctly rounded quotients but for rare ties
        const float pa = fmaf(-pu, k2e, k0), pb = fmaf(-pv, k2e, k1);
        float k2v = k2e;
        asm("" : "+v"(k2v));                                                 // VALU operands from VGPRs: an SGPR source
        asm("" : "+v"(a2));                                                  // costs the FMA its full issue rate on gfx950

        const int pix = y * W + x;
        // octets are visited in order, so the output address is a running per-lane pointer
        float* optr = a.out + (LAYOUT == 0 ? ((size_t)p * D + 8 * obeg) * HW + pix
                                           : c4_offset(p, LAYOUT == 1 ? D / 4 + 1 : D / 8 + 1, LAYOUT == 1 ? 2 * obeg : obeg, HW, pix));
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
            if (staged) {
#ifdef SWEEP_EMU
              if (!(sweep_emu_mode & 4)) {                                       // mode bit 2: the box is NOT staged (timing of a launch whose staging is free; wrong output)
#endif
                // ---- stage the box: rh rows of rw texels + one halo column, as items i = r (rw + 1) + c dealt 63 per
                // wave pass (lane 63 repeats the next pass' first item: it only feeds lane 62).  A lane loads column c of
                // image rows y and y + 1, column c + 1 comes from the next lane.  All loads of the box are issued first.
                if (g == 0) SWEEP_TL(3);                                         // first box: staging starts
                const int pitch = bx.rw + 1, n = pitch * bx.rh;
                const float inv_pitch = 1.0f / (float)pitch;
                const int origin4 = (bx.ry0 * W + bx.rx0) * 4;                // byte offset of box texel (0,0) in a channel plane
                float p0[4][3], p1[4][3];
                int dst[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k * SWEEP_TH * 63 >= n) break;                        // wave-uniform: passes the box does not need
                    const int i = (k * SWEEP_TH + wave) * 63 + lane;
                    const int r = (int)(((float)i + 0.5f) * inv_pitch);      // exact for i < 2^21 / pitch
                    int c, t, d;                                             // 24-bit multiply-adds (v_mul_lo_u32 is quarter rate)
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c) : "v"(r), "s"(-pitch), "v"(i));        // c = i - r pitch
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(r), "s"(W), "v"(c));             // texel offset from the origin
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(r), "s"(bx.rw), "v"(c));         // box texel index
                    const int xx = bx.rx0 + c, yy = bx.ry0 + r;
                    const bool in = i < n, xin = (unsigned)xx < (unsigned)W;
                    const unsigned o = (unsigned)(t * 4 + origin4);
                    const unsigned o0 = (in && xin && (unsigned)yy < (unsigned)H) ? o : 0xFFFFFFFFu;
                    const unsigned o1 = (in && xin && (unsigned)(yy + 1) < (unsigned)H) ? o + (unsigned)W * 4u : 0xFFFFFFFFu;
                    dst[k] = (in && c < bx.rw && lane < 63) ? d : -1;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        p0[k][ch] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o0, ch * chan_bytes, 0));
                        p1[k][ch] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o1, ch * chan_bytes, 0));
                    }
                }
                // the loads of this box travel while the slower waves finish the previous one (the waves of a tile end their sweeps
                // 2500 - 3000 cycles apart: -DSWEEP_TIMELINE)
                if (g > 0) __syncthreads();                                      // every wave is done with the previous box
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k * SWEEP_TH * 63 >= n) break;
                    float q0[3], q1[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) { q0[ch] = sweep_next_lane(p0[k][ch]); q1[ch] = sweep_next_lane(p1[k][ch]); }
                    const SweepTexel t = sweep_texel_pack(p0[k], q0, p1[k], q1);
                    if (dst[k] >= 0) {
                        unsigned off;
                        asm("v_mul_u32_u24 %0, %1, 48" : "=v"(off) : "v"(dst[k]));
                        float4* tb = reinterpret_cast<float4*>(reinterpret_cast<char*>(box) + off);
                        tb[0] = t.u0; tb[1] = t.u1; tb[2] = t.u2;
                    }
                }
                if (g == 0) SWEEP_TL(4);                                         // first box: texels written
#ifdef SWEEP_EMU
              }
#endif
                __syncthreads();
                if (g == 0) SWEEP_TL(5);
            } else {                                                             // whole zero-extended image as the "box"
                bx.rx0 = -2; bx.ry0 = -2; bx.rw = W + 3; bx.rh = H + 3;
            }
            const float cu = -(0.5f + (float)bx.rx0), cv = -(0.5f + (float)bx.ry0);
            float umax = (float)(bx.rw - 1), vmax = (float)(bx.rh - 1);
            const float ug = pu + cu, vg = pv + cv;
            unsigned rwv = (unsigned)bx.rw;
            asm("" : "+v"(umax)); asm("" : "+v"(vmax)); asm("" : "+v"(rwv));
            const int o_end = obeg + min((g + 1) << level, ocnt);
            for (int o = obeg + (g << level); o < o_end; ++o) {
                const int d0 = o * 8;
                float cost[8];
                if (staged) {
                    if (clamp) sweep_octet<true>(box, zsh + d0, cost, ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, rr, rg, rb);
                    else sweep_octet<false>(box, zsh + d0, cost, ug, vg, umax, vmax, rwv, pa, pb, a2, k2v, rr, rg, rb);
                } else {
#pragma unroll 1
                    for (int j = 0; j < 8; ++j) {
                        const SweepCoord cj = sweep_coords(cu, cv, umax, vmax, a0, a1, a2, k0, k1, k2e, zsh[d0 + j]);
                        const SweepTexel t = sweep_texel_global(rsrc, (int)cj.xi + bx.rx0, (int)cj.yi + bx.ry0, W, H, chan_bytes);
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
        if (LAYOUT == 1 && pvalid && obeg + ocnt == noct) *reinterpret_cast<float4*>(optr) = make_float4(rr, rg, rb, 0.f);
        if (LAYOUT == 2 && pvalid && obeg + ocnt == noct) {
            const sw_f16x8 h = {(_Float16)rr, (_Float16)rg, (_Float16)rb, 0, 0, 0, 0, 0};
            *reinterpret_cast<sw_f16x8*>(optr) = h;
        }
        SWEEP_TL(6);                                                        // tile done
#ifdef SWEEP_SPAN
        if (tid == 0 && blockIdx.x < 1024) { sweep_span[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime(); sweep_span[blockIdx.x][3] += 1;
                                             if (cur_tile_span < 4096) sweep_unit_ticks[cur_tile_span] = (unsigned)(sweep_span[blockIdx.x][2] - span_u0); }
#endif
#ifdef SWEEP_TIMELINE
        tlp += 8;
#endif
        unit = next_unit;
    }
    // every workgroup draws exactly one ticket beyond the last unit; the last one to leave rearms the counters
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

// workgroups the chip holds at once (SWEEP_MINW waves per SIMD = that many 256-thread quarters per CU); one query per device
static int sweep_resident_workgroups() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
        cached[dev] = cus * (SWEEP_MINW * 256 / SWEEP_NT);
    }
    return cached[dev];
}

static int sweep_launch(int layout, const float* ref, const float* src, const float* hmkt, float* out,
                        float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                        double idepth_min, double idepth_max, void* stream) {
    CNM_REQUIRE(ref && src && hmkt && out, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ws == nullptr || (((uintptr_t)ws & 15) == 0 && ws_floats >= 4), CNM_ERR_WORKSPACE);
    CNM_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D >= 2 && D <= CNM_MAX_PLANES, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout == 0 || D % 4 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(layout != 2 || D % 8 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((long long)H * W * 12 < (1ll << 31), CNM_ERR_BAD_ARG);
    const long long ntiles = (long long)cnm_ceil_div(W, SWEEP_TW) * cnm_ceil_div(H, SWEEP_TH) * B * S;
    CNM_REQUIRE(ntiles < (1ll << 20), CNM_ERR_BAD_ARG);                      // the kernel decodes tile ids with fp32 reciprocals
    SweepArgs a;
    a.ref = ref; a.src = src; a.hmkt = hmkt; a.out = out;
    a.queue = reinterpret_cast<unsigned int*>(ws);
    a.B = B; a.S = S; a.H = H; a.W = W; a.D = D;
    a.idmin = idepth_min;
    a.idstep = (idepth_max - idepth_min) / (D - 1.0);                        // depthNet_model.py:194
    const int slots = sweep_resident_workgroups();
    CNM_REQUIRE(slots > 0, CNM_ERR_LAUNCH);
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
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
