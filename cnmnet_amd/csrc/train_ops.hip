// Training-path kernels (reference train.py:164-310):
//   weight gradient of the convolution (fp32 MFMA, split over the pixel reduction),
//   train-mode BatchNorm2d (+ReLU) forward/backward on c4 activations (batch statistics, torch defaults),
//   adjoint of the bilinear x2 upsample.
// The plane sweep needs no backward: images and cameras carry no gradient.
#include "cnm_common.h"
#include "sync_ws.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 tr_buffer_load_f4(const float* base, unsigned bytes, unsigned voff) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ------------------------------------------------------------------ weight gradient
// dWp[co][k] = sum_pix dY[co][pix] * Xcol[k][pix],  k = (ky*ks+kx)*4*Gin + cpacked  (the forward's k order).
// GEMM with the PIXELS as the reduction dimension: tile 128 (co) x 128 (k), 16 pixels per step, split over
// pixel ranges (grid.z) into partial buffers that a second kernel sums and scatters back to OIHW.
struct WgradArgs {
    const float* x; const float* dy; float* partial;
    unsigned x_bytes, dy_bytes;
    int N, H, W, Ho, Wo, Gx_tot, gx0, Gin, Gy_tot, gy0, Cout, Cout_pad;
    int ks, stride, pad, Kflat, Kpad128, M, pix_per_split;
    int per_image_splits;           // 0: one reduction over all N*Ho*Wo pixels, split by grid.z.  s > 0: grid.z = problem * s + split -- every group of
                                    // `ipp` images is a problem of its own (the frequency points of the Winograd-domain gradients below), M = ipp*Ho*Wo
    int N_problems;                 // stream-K form: problems (frequency points) of the launch, 1 for the direct gradient
    int ipp, ksx, padx;             // images per problem; taps / padding along x (ks, pad: along y) -- the direct gradient has ksx = ks, padx = pad
};

// TCO = couts per workgroup: 128 (waves 2 x 2, 64 x 64 each) or 64 (waves 1 x 4, 64 couts x 32 k each) for the 64-cout
// layers, where a 128-row tile would spend half of its MFMAs on padding rows.
template <int TCO> struct WgradTile {
    static constexpr int LDK = 20, WK = TCO == 128 ? 2 : 4, PJ = 4 / WK;   // waves along k, 32-column blocks per wave
    static constexpr int SMEM_FLOATS = 2 * (TCO + 128) * LDK;
};

// The reduction of ONE tile (couts c0 .. c0 + TCO, flat k k0 .. k0 + 128) over the pixels r0 .. r1 of problem `zimg`, added into acc.
// The workgroup's LDS is free again when this returns (it ends on a barrier).
// LINEAR [r6]: the Winograd-domain GEMMs (1 x 1 taps, one "image" of T tiles per frequency point) -- a thread's four operand addresses advance by
// 256 bytes per step and nothing else, so they are a per-thread constant plus a scalar offset: no vector ALU in the loop (the general walk's
// coordinates, range checks and address products are ~ 50 VALU instructions per 32 MFMAs and wave, and on this chip an fp32 VALU instruction
// takes matrix-pipe time from the fp32 MFMAs of every wave on its SIMD: the counters read 1.65 VALU per MFMA at 62 % matrix-pipe busy).
#ifndef WGRAD_ABL
#define WGRAD_ABL 0      // timing experiments (wrong results; tools/wgrad_ablate.sh): 1 no global loads, 2 no LDS stores, 4 no barriers, 8 no fragment reads
#endif
#define WGRAD_SYNC() do { if constexpr ((WGRAD_ABL & 4) == 0) __syncthreads(); } while (0)
template <int TCO, int MODE>
__device__ __forceinline__ void wgrad_tile_segment(const WgradArgs& a, float* smem, int c0, int k0, int zimg, int r0, int r1,
                                                   f32x16 (&acc)[2][WgradTile<TCO>::PJ]) {
    constexpr int LDK = WgradTile<TCO>::LDK, WK = WgradTile<TCO>::WK, PJ = WgradTile<TCO>::PJ;
    constexpr bool LINEAR = MODE == 1, ROWSTEP = MODE == 2;
    float* As = smem;                 // [2][TCO co][LDK pix]
    float* Bs = smem + 2 * TCO * LDK; // [2][128 k ][LDK pix]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wc = wave / WK, wp = wave % WK;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    // loader mapping: float4 slot f = t + i*256 (i = 0,1): pixel lane pl = f % 16, quad row q = f / 16 (0..31)
    const int pl = t & 15;
    int aq[2], bgc[2], bky[2], bkx[2]; bool bok[2], aok[2];
#ifdef WGRAD_SAMEDATA   // timing experiment (wrong results): every tile of every problem reads tile (0, 0) of problem 0 -- the operand stream of a launch fits the L2
    c0 = 0; k0 = 0; zimg = 0;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (t >> 4) + 16 * i;
        aq[i] = c0 / 4 + q; aok[i] = 4 * q < TCO && (c0 + 4 * q) < a.Cout;
        const int kq = k0 / 4 + q;                                  // flat k-quad -> (tap, channel group): fixed per thread
        const int tap = kq / a.Gin;
        bgc[i] = kq - tap * a.Gin; bky[i] = tap / a.ksx; bkx[i] = tap - bky[i] * a.ksx;
        bok[i] = tap < a.ks * a.ksx;
    }
    // running output-pixel coordinate of this thread's pixel lane
    int m = r0 + pl, img = m / HoWo, rem = m - img * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
    img += zimg * a.ipp;

    // Operand prefetch TWO steps ahead [r6]: the transformed tensors stream from HBM / the memory-side cache (50-110 MB per layer), and one
    // 16-pixel step is 2048 matrix-pipe cycles = 0.85 us -- less than a loaded HBM round trip.  Two register sets, by step parity: at the end
    // of step st the set of step st + 1 goes to LDS and is re-loaded with step st + 3.
    float4 ra[2][2], rb[2][2];
    // LINEAR: byte offsets of this thread's pixel lane at step 0 of the segment (all-ones: the row is padding -- out of range whatever the scalar
    // offset adds: raw buffers check the vector offset alone); `lin_step` counts the steps loaded so far
    unsigned lva[2], lvb[2]; int lin_step = 0;
    if constexpr (LINEAR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lva[i] = aok[i] ? ((unsigned)((zimg * a.Gy_tot + a.gy0 + aq[i]) * HoWo + r0 + pl)) * 16u : 0xFFFFFFFFu;
            lvb[i] = bok[i] ? ((unsigned)((zimg * a.Gx_tot + a.gx0 + bgc[i]) * HW + r0 + pl)) * 16u : 0xFFFFFFFFu;
        }
    }
    // ROWSTEP [r6]: Wo a multiple of 16 -- the 16 pixels of a step lie in ONE output row, so the walk (image, row, column of the step's first pixel)
    // is scalar; a thread adds its own constants: the dY address needs no vector ALU at all, an X address one add per load and two compares for
    // the tap's row / column range (the general walk below: ~ 47 VALU instructions per step and wave)
    int rs_img = 0, rs_oy = 0, rs_ox = 0;                                 // of pixel r0 + 16 * (steps loaded so far)
    unsigned rva[2] = {0, 0}; int rvb[2] = {0, 0}, riy[2] = {0, 0}, rix[2] = {0, 0};
    if constexpr (ROWSTEP) {
        rs_img = __builtin_amdgcn_readfirstlane(r0 / HoWo);
        const int rem0 = r0 - rs_img * HoWo;
        rs_oy = __builtin_amdgcn_readfirstlane(rem0 / a.Wo); rs_ox = __builtin_amdgcn_readfirstlane(rem0 - rs_oy * a.Wo);
        rs_img += zimg * a.ipp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rva[i] = aok[i] ? ((unsigned)((a.gy0 + aq[i]) * HoWo + pl)) * 16u : 0xFFFFFFFFu;   // + scalar (img Gy_tot HoWo + oy Wo + ox) * 16
            riy[i] = bky[i] - a.pad; rix[i] = pl * a.stride + bkx[i] - a.padx;                  // + scalar oy * stride / ox * stride
            rvb[i] = ((a.gx0 + bgc[i]) * HW + riy[i] * a.W + rix[i]) * 16;                      // + scalar (img Gx_tot HW + oy stride W + ox stride) * 16; may be negative: added in the VALU
        }
    }
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    auto load = [&](float4 (&qa)[2], float4 (&qb)[2]) {
        if constexpr ((WGRAD_ABL & 1) != 0) { asm volatile("" : "+v"(qa[0].x), "+v"(qa[1].x), "+v"(qb[0].x), "+v"(qb[1].x)); return; }
        if constexpr (LINEAR) {
            const int soff = __builtin_amdgcn_readfirstlane(lin_step * 256);
            const bool mok = r0 + 16 * lin_step + pl < r1;               // only the problem's last, partial step has lanes beyond the last tile (branch-free on purpose: with a
            ++lin_step;                                                  // scalar branch around a masked twin hipcc put an s_waitcnt vmcnt(0) between the loads of a step)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsa, mok ? lva[i] : 0xFFFFFFFFu, soff, 0), vb = __builtin_amdgcn_raw_buffer_load_b128(rsb, mok ? lvb[i] : 0xFFFFFFFFu, soff, 0);
                qa[i] = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w));
                qb[i] = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
            }
            return;
        }
        if constexpr (ROWSTEP) {
            const int m0 = r0 + 16 * lin_step; ++lin_step;
            const bool mok = m0 + pl < r1;
            const unsigned sa = (unsigned)(((rs_img * a.Gy_tot) * HoWo + rs_oy * a.Wo + rs_ox) * 16);
            const unsigned sb = (unsigned)((rs_img * a.Gx_tot) * HW + rs_oy * a.stride * a.W + rs_ox * a.stride) * 16u;
            const int sy = rs_oy * a.stride, sx = rs_ox * a.stride;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsa, mok ? rva[i] : 0xFFFFFFFFu, sa, 0);
                const bool ok = mok && bok[i] && (unsigned)(riy[i] + sy) < (unsigned)a.H && (unsigned)(rix[i] + sx) < (unsigned)a.W;
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsb, ok ? (unsigned)rvb[i] + sb : 0xFFFFFFFFu, 0, 0);
                qa[i] = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w));
                qb[i] = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
            }
            rs_ox += 16;                                                 // scalar walk: Wo % 16 == 0, a step never straddles a row
            if (rs_ox >= a.Wo) { rs_ox = 0; if (++rs_oy >= a.Ho) { rs_oy = 0; ++rs_img; } }
            return;
        }
        const bool mok = m < r1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned offa = (mok && aok[i]) ? ((unsigned)((img * a.Gy_tot + a.gy0 + aq[i]) * HoWo + rem)) * 16u : 0xFFFFFFFFu;
            qa[i] = tr_buffer_load_f4(a.dy, a.dy_bytes, offa);
            const int iy = oy * a.stride - a.pad + bky[i], ix = ox * a.stride - a.padx + bkx[i];
            const bool ok = mok && bok[i] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned offb = ok ? ((unsigned)((img * a.Gx_tot + a.gx0 + bgc[i]) * HW + iy * a.W + ix)) * 16u : 0xFFFFFFFFu;
            qb[i] = tr_buffer_load_f4(a.x, a.x_bytes, offb);
        }
        m += 16; ox += 16; rem += 16;
        while (ox >= a.Wo) { ox -= a.Wo; ++oy; }
        while (oy >= a.Ho) { oy -= a.Ho; ++img; rem -= HoWo; }
    };
    auto store = [&](int buf, const float4 (&qa)[2], const float4 (&qb)[2]) {
        if constexpr ((WGRAD_ABL & 2) != 0) { asm volatile("" :: "v"(qa[0].x), "v"(qa[1].w), "v"(qb[0].x), "v"(qb[1].w)); return; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * ((t >> 4) + 16 * i);
            if (row < TCO) {
                float* pa = As + ((size_t)buf * TCO + row) * LDK + pl;
                pa[0] = qa[i].x; pa[LDK] = qa[i].y; pa[2 * LDK] = qa[i].z; pa[3 * LDK] = qa[i].w;
            }
            float* pb = Bs + ((size_t)buf * 128 + row) * LDK + pl;
            pb[0] = qb[i].x; pb[LDK] = qb[i].y; pb[2 * LDK] = qb[i].z; pb[3 * LDK] = qb[i].w;
        }
    };
    const int nsteps = (r1 - r0 + 15) / 16;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    auto compute = [&](int buf) {
        const float* Ab = As + ((size_t)buf * TCO + wc * 64 + frow) * LDK + fk;
        const float* Bb = Bs + ((size_t)buf * 128 + wp * 32 * PJ + frow) * LDK + fk;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            float4 af[2], bf[PJ];
            if constexpr ((WGRAD_ABL & 8) != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { af[i] = make_float4(1.f, 2.f, 3.f, 4.f); asm volatile("" : "+v"(af[i].x), "+v"(af[i].y), "+v"(af[i].z), "+v"(af[i].w)); }
#pragma unroll
                for (int j = 0; j < PJ; ++j) { bf[j] = make_float4(1.f, 2.f, 3.f, 4.f); asm volatile("" : "+v"(bf[j].x), "+v"(bf[j].y), "+v"(bf[j].z), "+v"(bf[j].w)); }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + kg * 8);
#pragma unroll
            for (int j = 0; j < PJ; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + kg * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    if (nsteps > 0) {
        // every load below is issued UNCONDITIONALLY (steps beyond the segment have all lanes out of range: zeros, no memory access): with
        // `if (st + 3 < nsteps) load(...)` the number of loads in flight differed between paths and hipcc's s_waitcnt in front of the LDS stores
        // assumed the smaller one -- vmcnt(3) .. (0) instead of (7) .. (4), i.e. it also waited for the set loaded one step ago
        load(ra[0], rb[0]); store(0, ra[0], rb[0]);                      // step 0 -> LDS
        load(ra[1], rb[1]);                                              // step 1 (odd set)
        load(ra[0], rb[0]);                                              // step 2 (even set)
        WGRAD_SYNC();
        for (int st = 0; st < nsteps; st += 2) {
            compute(0);                                                  // step st (even)
            store(1, ra[1], rb[1]);
            load(ra[1], rb[1]);                                          // step st + 3
            WGRAD_SYNC();
            if (st + 1 < nsteps) {
                compute(1);                                              // step st + 1 (odd)
                store(0, ra[0], rb[0]);
                load(ra[0], rb[0]);                                      // step st + 4
                WGRAD_SYNC();
            }
        }
    }
}

// out[co][k] of one tile: acc row = co, col = k
template <int TCO>
__device__ __forceinline__ void wgrad_tile_store(const WgradArgs& a, float* P, int c0, int k0, const f32x16 (&acc)[2][WgradTile<TCO>::PJ]) {
    constexpr int WK = WgradTile<TCO>::WK, PJ = WgradTile<TCO>::PJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wc = wave / WK, wp = wave % WK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int k = k0 + (wp * PJ + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c0 + (wc * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                P[(size_t)co * a.Kpad128 + k] = acc[i][j][r];
            }
        }
}

// Split form: grid.z = pixel ranges (x problems); every workgroup writes its partial tile, a second kernel sums the splits.  Kept for A/B
// (cnm_tune_wgrad_streamk(0)) and for devices whose sync workspace cannot be had.
template <int TCO, int MODE>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int PJ = WgradTile<TCO>::PJ;
    __shared__ __attribute__((aligned(16))) float smem[WgradTile<TCO>::SMEM_FLOATS];
    const int c0 = blockIdx.x * TCO, k0 = blockIdx.y * 128;
    const int zsplit = a.per_image_splits ? (int)blockIdx.z % a.per_image_splits : (int)blockIdx.z;
    const int zimg = a.per_image_splits ? (int)blockIdx.z / a.per_image_splits : 0;
    const int r0 = zsplit * a.pix_per_split, r1 = min(r0 + a.pix_per_split, a.M);
    f32x16 acc[2][PJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    wgrad_tile_segment<TCO, MODE>(a, smem, c0, k0, zimg, r0, r1, acc);
    // partial[split][co][k]
    wgrad_tile_store<TCO>(a, a.partial + (size_t)blockIdx.z * a.Cout_pad * a.Kpad128, c0, k0, acc);
}

// [r6] Stream-K form: a PERSISTENT grid of G workgroups (three per CU: what the kernel's registers and LDS allow) shares the flattened
// (tile, 16-pixel step) space in equal contiguous ranges, so a tile's reduction is cut only where a range boundary falls into it -- at
// most one partial tile per workgroup travels through memory (G x 64 KB) instead of `splits` partial copies of EVERY tile (134 MB written
// and read back per layer: 8.6 GB per training step), and there is no second kernel.  A workgroup walks its range from the LAST tile to the
// first: the part that does not reach its tile's end (the head of a tile whose tail belongs to the next workgroup) comes first and is
// PUBLISHED at once (sc1 stores into this workgroup's slot, then its flag = the launch's generation, sync_ws.h); the part that holds a
// tile's end comes last and FINISHES the tile: it waits for the lower-numbered workgroups that hold the tile's earlier parts -- they published
// at the start of their walk, long ago -- adds their slots in descending workgroup order and stores the tile.  Waits point at lower workgroup
// numbers only (dispatched earlier), so progress does not depend on all G workgroups being resident; the summation order is a function of
// (shape, G) alone: bit-reproducible on a device.  A time-out is reported as for the convolution kernels (cnm_engine_status).
template <int TCO, int MODE>
__global__ __launch_bounds__(256, 3) void conv_wgrad_sk_kernel(const WgradArgs a, int tilesC, int tilesK, int S, unsigned* __restrict__ flags, float* __restrict__ slots) {
    constexpr int PJ = WgradTile<TCO>::PJ, SLOT_FLOATS = TCO * 128;
    __shared__ __attribute__((aligned(16))) float smem[WgradTile<TCO>::SMEM_FLOATS];
    const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x;
    const long long Wtot = (long long)a.N_problems * tilesC * tilesK * S;
    const auto range_begin = [&](int r) { return Wtot * r / G; };
    const long long fb = range_begin(g);
    long long fe = range_begin(g + 1);
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc(slots, 0, (unsigned)G * (unsigned)(SLOT_FLOATS * 4), 0x00020000);
    while (fe > fb) {
        const long long tile = (fe - 1) / S, t0 = tile * S;
        const int s1 = (int)(fe - t0), s0 = (int)((fb > t0 ? fb : t0) - t0);
        const int bx = (int)(tile % tilesC), rest = (int)(tile / tilesC), by = rest % tilesK, z = rest / tilesK;
        const int c0 = bx * TCO, k0 = by * 128;
        f32x16 acc[2][PJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        wgrad_tile_segment<TCO, MODE>(a, smem, c0, k0, z, s0 * 16, min(s1 * 16, a.M), acc);
        if (s1 < S) {
            // ---- publish: [workgroup][(i, j, quad)][thread] float4 -- every lane of the finisher re-reads exactly what its twin wrote
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), srsrc,
                                                               ((unsigned)g * (unsigned)SLOT_FLOATS + (unsigned)(((i * PJ + j) * 4 + q) * 256 + t) * 4u) * 4u, 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) sync_publish(flags, g, sync_generation());
        } else {
            if (s0 > 0) {
                // ---- finish a tile whose earlier parts lie in the ranges g - 1, g - 2, ... down to the one that holds the tile's first step
                int nsrc = 1;
                while (range_begin(g - nsrc) > t0) ++nsrc;
                if (t == 0) {
                    const unsigned gen = sync_generation();
                    for (int k = 1; k <= nsrc; ++k) sync_wait(flags, g - k, gen);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int k = 1; k <= nsrc; ++k) {                        // fixed order: own part, then the ranges below in descending order; sc1 loads match the sc1 stores
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < PJ; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(srsrc, ((unsigned)(g - k) * (unsigned)SLOT_FLOATS + (unsigned)(((i * PJ + j) * 4 + q) * 256 + t) * 4u) * 4u, 0, 16);
                                const f32x4v f = *reinterpret_cast<const f32x4v*>(&pv);
                                acc[i][j][4 * q] += f[0]; acc[i][j][4 * q + 1] += f[1]; acc[i][j][4 * q + 2] += f[2]; acc[i][j][4 * q + 3] += f[3];
                            }
                }
            }
            wgrad_tile_store<TCO>(a, a.partial + (size_t)z * a.Cout_pad * a.Kpad128, c0, k0, acc);
        }
        fe = t0 + s0;
    }
}

// One thread per (cout, flat k) of the PADDED partial layout, so the `splits` reads of a wave are contiguous rows (the
// reads outnumber the one scattered OIHW write per element by the split count); fp64 sum.
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int Cout, int Cout_pad, int Cin, int ks,
                                    int rot, int Kpad128, float* __restrict__ dw) {
    const int Cp = 4 * ((Cin + 3) / 4), Kflat = ks * ks * Cp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Kflat) return;
    const int co = (int)(idx / Kflat), k = (int)(idx - (long long)co * Kflat);
    const int tap = k / Cp, cp = k - tap * Cp;
    double s = 0.0;
    const float* q = partial + (size_t)co * Kpad128 + k;
    const size_t zstride = (size_t)Cout_pad * Kpad128;
    for (int z = 0; z < splits; ++z) s += (double)q[(size_t)z * zstride];
    if (cp < Cin) dw[((size_t)co * Cin + (cp + rot) % Cin) * (ks * ks) + tap] = (float)s;
}

#ifndef CNM_WGRAD_WORKGROUPS
#define CNM_WGRAD_WORKGROUPS 2048
#endif
static inline int wg_round(int v, int m) { return (v + m - 1) / m * m; }

static inline int wgrad_tco(int Cout) { return wg_round(Cout, 64) % 128 ? 64 : 128; }   // couts per workgroup

// ---- stream-K form (conv_wgrad_sk_kernel): host side
static int g_wgrad_linear = 1;                                           // 1: the Winograd-domain GEMMs use the scalar-offset loader (default); 0: the general walk (A/B)
extern "C" int cnm_tune_wgrad_linear(int n) { const int old = g_wgrad_linear; if (n == 0 || n == 1) g_wgrad_linear = n; return old; }
static int g_wgrad_streamk = 1;                                          // 1: persistent stream-K launch, no split partials (default); 0: the split form + reduction kernels (A/B)
extern "C" int cnm_tune_wgrad_streamk(int n) { const int old = g_wgrad_streamk; if (n == 0 || n == 1) g_wgrad_streamk = n; return old; }
static int wgrad_cus() {                                                 // compute units of the current device, queried once per device
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (!cus[dev]) { int n = 0; cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    return cus[dev];
}
// Which launches take the stream-K form: those whose tiles are shared by at most `g_wgrad_sk_share` ranges on average -- the range that
// finishes a tile adds the other ranges' partial tiles one after the other (sc1 loads from the memory side, a round trip each), so a launch
// of a few dozen tiles cut into 768 ranges ends on 10-20 serial round trips (tools/wgrad_sk_trace.sh).  0 = no limit.
static int g_wgrad_sk_share = 8;
extern "C" int cnm_tune_wgrad_streamk_share(int n) { const int old = g_wgrad_sk_share; if (n >= 0) g_wgrad_sk_share = n; return old; }
static inline int wgrad_sk_max_ranges();
static bool wgrad_sk_wanted(const WgradArgs& a, int NP) {
    if (!g_wgrad_streamk) return false;
    const int tco = a.Cout_pad % 128 ? 64 : 128;
    const long long tiles = (long long)NP * (a.Cout_pad / tco) * (a.Kpad128 / 128), S = (a.M + 15) / 16;
    long long G = wgrad_sk_max_ranges();
    if (G > tiles * S / 16) G = tiles * S / 16 > 0 ? tiles * S / 16 : 1;
    return g_wgrad_sk_share == 0 || G <= (long long)g_wgrad_sk_share * tiles;
}
static inline int wgrad_sk_max_ranges() { const int g = 3 * wgrad_cus(); return g < kSyncMaxRanges ? g : kSyncMaxRanges; }   // three workgroups per CU: 132 registers, 40 KB of LDS
// floats of the sync area appended to every weight-gradient workspace: flag words + one 64 KB (TCO = 128) partial-tile slot per range
static inline size_t wgrad_sk_sync_floats() { return kSyncFlagBytes / 4 + (size_t)wgrad_sk_max_ranges() * 128 * 128; }
// Launch the persistent form over `NP` problems; out = [NP][Cout_pad][Kpad128] (the split form's layout with one split).  `sync` = the
// workspace's sync area; its flag words must be zero (zero_flags: a memset node here -- the Winograd paths let their transform kernel do it).
// the Winograd-domain GEMMs: 1 x 1 taps over one row of T tiles per problem -- the loader's addresses are linear in the step
static inline bool wgrad_linear(const WgradArgs& a) { return a.ks == 1 && a.ksx == 1 && a.H == 1 && a.Ho == 1 && a.W == a.Wo && a.ipp == 1 && a.stride == 1 && a.pad == 0 && a.padx == 0 && g_wgrad_linear; }
// loader of a launch: 1 = linear (above); 2 = row steps (Wo a multiple of 16 and the split boundaries on steps: the walk is scalar); 0 = the general walk
static inline int wgrad_loader_mode(const WgradArgs& a) {
    if (wgrad_linear(a)) return 1;
    return (g_wgrad_linear && a.Wo % 16 == 0 && a.pix_per_split % 16 == 0) ? 2 : 0;
}
static void wgrad_split_launch(const WgradArgs& a, int tco, dim3 grid, hipStream_t s) {
    const int mode = wgrad_loader_mode(a);
#define WGRAD_SPLIT(T_, M_) conv_wgrad_kernel<T_, M_><<<grid, 256, 0, s>>>(a)
    if (tco == 128) { if (mode == 1) WGRAD_SPLIT(128, 1); else if (mode == 2) WGRAD_SPLIT(128, 2); else WGRAD_SPLIT(128, 0); }
    else { if (mode == 1) WGRAD_SPLIT(64, 1); else if (mode == 2) WGRAD_SPLIT(64, 2); else WGRAD_SPLIT(64, 0); }
#undef WGRAD_SPLIT
}
static int wgrad_sk_launch(WgradArgs a, int NP, float* sync, bool zero_flags, hipStream_t s) {
    const int tco = wgrad_tco(a.Cout), tilesC = a.Cout_pad / tco, tilesK = a.Kpad128 / 128, S = (a.M + 15) / 16;
    const long long Wtot = (long long)NP * tilesC * tilesK * S;
    long long G = wgrad_sk_max_ranges();
    if (G > Wtot / 16) G = Wtot / 16 > 0 ? Wtot / 16 : 1;                // at least 16 steps (256 pixels) per range
    a.N_problems = NP;
    sync_ctl_upload(s);
    if (cnm_sync_failed()) return CNM_ERR_LAUNCH;                        // an earlier hand-off timed out: refuse until cnm_engine_status(1) has acknowledged it
    unsigned* flags = reinterpret_cast<unsigned*>(sync);
    float* slots = sync + kSyncFlagBytes / 4;
    if (zero_flags && hipMemsetAsync(flags, 0, kSyncFlagBytes, s) != hipSuccess) { (void)hipGetLastError(); return CNM_ERR_LAUNCH; }
    const int mode = wgrad_loader_mode(a);
#define WGRAD_SK(T_, M_) conv_wgrad_sk_kernel<T_, M_><<<(unsigned)G, 256, 0, s>>>(a, tilesC, tilesK, S, flags, slots)
    if (tco == 128) { if (mode == 1) WGRAD_SK(128, 1); else if (mode == 2) WGRAD_SK(128, 2); else WGRAD_SK(128, 0); }
    else { if (mode == 1) WGRAD_SK(64, 1); else if (mode == 2) WGRAD_SK(64, 2); else WGRAD_SK(64, 0); }
#undef WGRAD_SK
    return CNM_OK;
}
static void wgrad_plan(int Cout, int Cin, int ksize, int M, int* Cout_pad, int* Kpad128, int* splits, int* pps) {
    const int tco = wgrad_tco(Cout);
    *Cout_pad = wg_round(Cout, tco);
    *Kpad128 = wg_round(ksize * ksize * 4 * ((Cin + 3) / 4), 128);
    const int tiles = (*Cout_pad / tco) * (*Kpad128 / 128);
    int s = (CNM_WGRAD_WORKGROUPS + tiles - 1) / tiles;       // aim at that many workgroups (1024 and 512 measured slower)
    const int maxs = (M + 255) / 256;                         // at least 256 pixels per split
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    *pps = wg_round((M + s - 1) / s, 16);
    *splits = (M + *pps - 1) / *pps;
}

extern "C" size_t cnm_conv2d_wgrad_workspace_floats(int Cout, int Cin, int ksize, int N, int Ho, int Wo) {
    if (Cout <= 0 || Cin <= 0 || ksize <= 0 || N <= 0 || Ho <= 0 || Wo <= 0) return 0;
    int cp, kp, sp, pps;
    wgrad_plan(Cout, Cin, ksize, N * Ho * Wo, &cp, &kp, &sp, &pps);
    return (size_t)sp * cp * kp + wgrad_sk_sync_floats();
}

extern "C" int cnm_conv2d_wgrad_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                       const float* dy, int Gy_total, int gy0, int Cout,
                                       float* dw_oihw, float* ws, size_t ws_floats,
                                       int N, int H, int W, int ksize, int stride, int rot, void* stream) {
    CNM_REQUIRE(x && dy && dw_oihw && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((ksize == 3 || ksize == 5 || ksize == 7) && (stride == 1 || stride == 2) && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = ws;
    a.N = N; a.H = H; a.W = W; a.ks = ksize; a.stride = stride; a.pad = (ksize - 1) / 2;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1; a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    a.Gx_tot = Gx_total; a.gx0 = gx0; a.Gin = (Cin + 3) / 4; a.Gy_tot = Gy_total; a.gy0 = gy0; a.Cout = Cout;
    a.Kflat = ksize * ksize * 4 * a.Gin; a.M = N * a.Ho * a.Wo;
    int splits;
    wgrad_plan(Cout, Cin, ksize, a.M, &a.Cout_pad, &a.Kpad128, &splits, &a.pix_per_split);
    a.per_image_splits = 0; a.ipp = 1; a.ksx = a.ks; a.padx = a.pad; a.N_problems = 1;
    CNM_REQUIRE((size_t)splits * a.Cout_pad * a.Kpad128 + wgrad_sk_sync_floats() <= ws_floats, CNM_ERR_WORKSPACE);
    const unsigned long long xb = (unsigned long long)N * Gx_total * H * W * 16ull, yb = (unsigned long long)N * Gy_total * a.Ho * a.Wo * 16ull;
    CNM_REQUIRE(xb < 0xFFFFFFFFull && yb < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    if (wgrad_sk_wanted(a, 1)) {
        const int rc = wgrad_sk_launch(a, 1, ws + (size_t)splits * a.Cout_pad * a.Kpad128, true, cnm_stream(stream));
        if (rc != CNM_OK) return rc;
        splits = 1;
    } else wgrad_split_launch(a, wgrad_tco(Cout), dim3(a.Cout_pad / wgrad_tco(Cout), a.Kpad128 / 128, splits), cnm_stream(stream));
    const long long total = (long long)Cout * a.Kflat;
    wgrad_reduce_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(
        ws, splits, Cout, a.Cout_pad, Cin, ksize, rot, a.Kpad128, dw_oihw);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ weight gradient of the 3x3 stride-1 layers in the Winograd domain
// y = A^T [ (G g G^T) (.) (B^T d B) ] A per 4x4 output tile (F(4x4,3x3), the forward's algorithm)  =>
//     dg = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G :
// for each of the 36 frequency points one GEMM  dU[xi][co][ci] = sum_t Yh[xi][co][t] Xh[xi][ci][t]  over the T = N*ceil(H/4)*ceil(W/4)
// tiles -- a quarter of the direct gradient's multiplies (36 per 16 pixels and tap-free, instead of 9 per pixel).  Unfused on
// purpose: the transformed tensors (2.25x the size of X and dY) are written once and read once, 0.05-0.1 ms per layer at HBM
// rates, against 0.3 ms for the direct kernel; the 36 GEMMs are ONE launch of conv_wgrad_kernel (the transformed tensors are
// laid out as c4 "images" [36][G][1][T][4]: per_image_splits mode, 1x1 taps); a reduction kernel sums the splits (fp64) and the
// finishing kernel applies G^T . G and scatters to OIHW.
typedef float wg_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wg_bt6(wg_f4& x0, wg_f4& x1, wg_f4& x2, wg_f4& x3, wg_f4& x4, wg_f4& x5) {   // B^T of F(4,3), points {0, 1, -1, 2, -2, inf}
    const wg_f4 t0 = 4.f * x0 - 5.f * x2 + x4, t5 = 4.f * x1 - 5.f * x3 + x5;
    const wg_f4 e1 = x4 - 4.f * x2, o1 = x3 - 4.f * x1, e2 = x4 - x2, o2 = x3 - x1;
    x0 = t0; x1 = e1 + o1; x2 = e1 - o1; x3 = e2 + 2.f * o2; x4 = e2 - 2.f * o2; x5 = t5;
}

// Xh[xi][g][t] = (B^T d B)[xi] of the 6x6 window of tile t (origin M ty - PADW, M tx - PADW, zero padding), channel group g.
// One thread per (tile, group), tiles fastest: the 36 stores of a wave are 36 contiguous 1 KB rows.
// S2 (the stride-2 layers, conv_winograd4s.hip): the "image" is one of the four pixel phases of x -- group gg = phase * Gin + g reads
// x(2 iy + py, 2 ix + px) on the H x W phase grid (= the output size); M = 3 with PADW = 2 for the 4x4 phase filters of a 7x7.
template <int M, bool S2>
__global__ __launch_bounds__(256) void wino_wgrad_xform_x_kernel(const float* __restrict__ x, int Gx_tot, int gx0, int Gin, int N, int H, int W,
                                                                 int TH, int TW, float* __restrict__ xh) {
    constexpr int PADW = (S2 && M == 3) ? 2 : 1;
    const int T = N * TH * TW, Geff = S2 ? 4 * Gin : Gin;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * Geff) return;
    const int t = (int)(idx % T), gg = (int)(idx / T);
    const int ph = S2 ? gg / Gin : 0, g = S2 ? gg - ph * Gin : gg, py = ph >> 1, px = ph & 1;
    const int n = t / (TH * TW), r = t - n * TH * TW, ty = r / TW, tx = r - ty * TW;
    const int Wi = S2 ? 2 * W : W, HWi = S2 ? 4 * H * W : H * W;         // the stored image
    const wg_f4* base = reinterpret_cast<const wg_f4*>(x + c4_offset(n, Gx_tot, gx0 + g, HWi, 0));
    wg_f4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int iy = M * ty - PADW + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = M * tx - PADW + j;
            const int off = S2 ? (2 * iy + py) * Wi + 2 * ix + px : iy * W + ix;
            d[i][j] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? base[off] : wg_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wg_bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) wg_bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    wg_f4* out = reinterpret_cast<wg_f4*>(xh) + (size_t)gg * T + t;
    const size_t plane = (size_t)Geff * T;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) out[(size_t)(i * 6 + j) * plane] = d[i][j];
}

// Yh[xi][g][t] = (A dY A^T)[xi] of the M x M output tile t (zero beyond the image), A = (A^T of F(M, 7 - M))^T:
// M = 4: rows [1 0 0 0], [1 1 1 1], [1 -1 1 -1], [1 2 4 8], [1 -2 4 -8], [0 0 0 1];  M = 3: [1 0 0], [1 1 1], [1 -1 1], [1 2 4], [1 -2 4], [0 0 1].
template <int M>
__device__ __forceinline__ void wg_a6(const wg_f4 (&v)[M], wg_f4 (&o)[6]) {
    if constexpr (M == 4) {
        const wg_f4 s02 = v[0] + v[2], s13 = v[1] + v[3], p = v[0] + 4.f * v[2], q = 2.f * v[1] + 8.f * v[3];
        o[0] = v[0]; o[1] = s02 + s13; o[2] = s02 - s13; o[3] = p + q; o[4] = p - q; o[5] = v[3];
    } else {
        const wg_f4 s02 = v[0] + v[2], p = v[0] + 4.f * v[2], q = 2.f * v[1];
        o[0] = v[0]; o[1] = s02 + v[1]; o[2] = s02 - v[1]; o[3] = p + q; o[4] = p - q; o[5] = v[2];
    }
}
template <int M>
__global__ __launch_bounds__(256) void wino_wgrad_xform_dy_kernel(const float* __restrict__ dy, int Gy_tot, int gy0, int Gout, int N, int H, int W,
                                                                  int TH, int TW, float* __restrict__ yh, unsigned* __restrict__ zero_flags) {
    const int T = N * TH * TW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_flags)                                                      // the flag words of the stream-K GEMM that follows on this stream
        for (long long i = idx; i < (long long)(kSyncFlagBytes / 4); i += (long long)gridDim.x * blockDim.x) zero_flags[i] = 0u;
    if (idx >= (long long)T * Gout) return;
    const int t = (int)(idx % T), g = (int)(idx / T);
    const int n = t / (TH * TW), r = t - n * TH * TW, ty = r / TW, tx = r - ty * TW;
    const wg_f4* base = reinterpret_cast<const wg_f4*>(dy + c4_offset(n, Gy_tot, gy0 + g, H * W, 0));
    wg_f4 c[M][6];                                                       // along x first: c[i][*] = A applied to row i of the tile
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int iy = M * ty + i;
        wg_f4 v[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int ix = M * tx + j;
            v[j] = (iy < H && ix < W) ? base[iy * W + ix] : wg_f4{0.f, 0.f, 0.f, 0.f};
        }
        wg_a6<M>(v, c[i]);
    }
    wg_f4* out = reinterpret_cast<wg_f4*>(yh) + (size_t)g * T + t;
    const size_t plane = (size_t)Gout * T;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        wg_f4 col[M], o[6];
#pragma unroll
        for (int i = 0; i < M; ++i) col[i] = c[i][j];
        wg_a6<M>(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) out[(size_t)(i * 6 + j) * plane] = o[i];
    }
}

// dU[xi][co][cp] = the sum of the `splits` partial tiles of frequency point xi (fp64 sum, one thread per element: 36 x Cout x Cp
// threads -- a thread per (cout, channel) that walked all 36 x splits partials itself ran 4352 threads for 0.5 ms on the 64-channel
// full-resolution layers), then dW[co][ci] = G^T dU[.][co][ci] G in fp64, scattered to OIHW.
// partial: [36 * splits][Cout_pad][Kpad128]; u: [36][Cout][Cp] (Cp = packed input channels, x 4 phases for the stride-2 form).
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int Cout, int Cout_pad, int Cp, int Kpad128,
                                                                float* __restrict__ u, int npts = 36) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)Cout * Cp;
    if (idx >= npts * per) return;
    const int xi = (int)(idx / per); const long long r = idx - xi * per;
    const int co = (int)(r / Cp), cp = (int)(r - (long long)co * Cp);
    const float* q = partial + (size_t)xi * splits * Cout_pad * Kpad128 + (size_t)co * Kpad128 + cp;
    const size_t zstride = (size_t)Cout_pad * Kpad128;
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += (double)q[(size_t)z * zstride];
    u[idx] = (float)s;
}

// FUSED (few splits: the low-resolution layers): u = the partial tiles themselves, every thread sums its 36 x splits values -- one
// launch and no 36 x Cout x Cp intermediate.  R = taps per axis of the (phase) filter: 3 (G of F(4,3)) or 4 (G of F(3,4)).
// ks = 0: a 3x3 stride-1 filter.  ks = 5 / 7: the stride-2 form -- packed channel cpe = phase * Cp + cp, tap (jy, jx) of phase (py, px)
// is w[2 jy + py - o][2 jx + px - o] (o = 0 for 5x5, 1 for 7x7), taps that leave the filter are dropped.
template <bool FUSED, int R>
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ u, int splits, int Cout, int Cout_pad, int Kpad128,
                                                                int Cin, int rot, int ks, float* __restrict__ dw) {
    const int Cp = 4 * ((Cin + 3) / 4), Ce = ks ? 4 * Cp : Cp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)Cout * Ce;
    if (idx >= per) return;
    const int co = (int)(idx / Ce), cpe = (int)(idx - (long long)co * Ce);
    const int ph = cpe / Cp, cp = cpe - ph * Cp;
    if (cp >= Cin) return;
    double uu[6][6];
    if constexpr (FUSED) {
        const float* q = u + (size_t)co * Kpad128 + cpe;
        const size_t zstride = (size_t)Cout_pad * Kpad128;
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) {
            double s = 0.0;
            for (int z = 0; z < splits; ++z) s += (double)q[(size_t)(xi * splits + z) * zstride];
            uu[xi / 6][xi % 6] = s;
        }
    } else {
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) uu[xi / 6][xi % 6] = (double)u[(size_t)xi * per + idx];
    }
    constexpr double G3[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
    constexpr double G4[6][4] = {{1. / 4, 0, 0, 0}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6},
                                 {1. / 24, 1. / 12, 1. / 6, 1. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3}, {0, 0, 0, 1}};
    const int K = ks ? ks : 3, py = ph >> 1, px = ph & 1, o_ = ks == 7 ? 1 : 0;
    float* o = dw + ((size_t)co * Cin + (cp + rot) % Cin) * (K * K);
#pragma unroll
    for (int p = 0; p < R; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) s += (R == 3 ? G3[a][p] * G3[b][r] : G4[a][p] * G4[b][r]) * uu[a][b];
            const int ky = ks ? 2 * p + py - o_ : p, kx = ks ? 2 * r + px - o_ : r;
            if (ky >= 0 && ky < K && kx >= 0 && kx < K) o[ky * K + kx] = (float)s;
        }
}

static void wino_wgrad_plan(int Cout, int Ceff, int T, int* Cout_pad, int* Kpad128, int* splits, int* pps) {   // Ceff = packed input channels (x 4 phases)
    const int tco = wgrad_tco(Cout);
    *Cout_pad = wg_round(Cout, tco);
    *Kpad128 = wg_round(Ceff, 128);
    const int tiles = 36 * (*Cout_pad / tco) * (*Kpad128 / 128);
    int s = (CNM_WGRAD_WORKGROUPS + tiles - 1) / tiles;
    const int maxs = (T + 255) / 256;                                    // at least 256 tiles per split
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    *pps = wg_round((T + s - 1) / s, 16);
    *splits = (T + *pps - 1) / *pps;
}

// ks = 3: stride 1 (H, W = image size).  ks = 5 / 7: stride 2 on the pixel phases (H, W = INPUT size, even).
static size_t wino_wgrad_ws(int Cout, int Cin, int N, int H, int W, int ks) {
    if (Cout <= 0 || Cin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
    const int M = ks == 7 ? 3 : 4, Ho = ks == 3 ? H : H / 2, Wo = ks == 3 ? W : W / 2, ph = ks == 3 ? 1 : 4;
    const size_t T = (size_t)N * ((Ho + M - 1) / M) * ((Wo + M - 1) / M);
    const int Ceff = ph * 4 * ((Cin + 3) / 4);
    int cp, kp, sp, pps;
    wino_wgrad_plan(Cout, Ceff, (int)T, &cp, &kp, &sp, &pps);
    return 36 * T * (size_t)Ceff + 36 * T * 4 * (size_t)((Cout + 3) / 4) + (size_t)36 * sp * cp * kp + (size_t)36 * Cout * Ceff + wgrad_sk_sync_floats();
}

static int wino_wgrad(const float* x, int Gx_total, int gx0, int Cin, const float* dy, int Gy_total, int gy0, int Cout,
                      float* dw_oihw, float* ws, size_t ws_floats, int N, int H, int W, int ks, int rot, void* stream) {
    CNM_REQUIRE(x && dy && dw_oihw && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % 4 == 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ks == 3 || ((ks == 5 || ks == 7) && H % 2 == 0 && W % 2 == 0), CNM_ERR_BAD_ARG);
    const bool s2 = ks != 3;
    const int M = ks == 7 ? 3 : 4, Ho = s2 ? H / 2 : H, Wo = s2 ? W / 2 : W;
    const int Gin = (Cin + 3) / 4, Geff = (s2 ? 4 : 1) * Gin, Gout = Cout / 4, TH = (Ho + M - 1) / M, TW = (Wo + M - 1) / M;
    const long long Tll = (long long)N * TH * TW;
    CNM_REQUIRE(gx0 >= 0 && gx0 + Gin <= Gx_total && gy0 >= 0 && gy0 + Gout <= Gy_total && Tll < (1ll << 24), CNM_ERR_BAD_ARG);
    const int T = (int)Tll;
    CNM_REQUIRE(wino_wgrad_ws(Cout, Cin, N, H, W, ks) <= ws_floats, CNM_ERR_WORKSPACE);
    float* xh = ws; float* yh = xh + (size_t)36 * T * 4 * Geff; float* partial = yh + (size_t)36 * T * 4 * Gout;
    const unsigned long long xb = 36ull * T * Geff * 16ull, yb = 36ull * T * Gout * 16ull;
    CNM_REQUIRE(xb < 0xFFFFFFFFull && yb < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    hipStream_t s = cnm_stream(stream);
    const unsigned nbx = (unsigned)cnm_ceil_div_ll((long long)T * Geff, 256), nby = (unsigned)cnm_ceil_div_ll((long long)T * Gout, 256);
    if (ks == 3) wino_wgrad_xform_x_kernel<4, false><<<nbx, 256, 0, s>>>(x, Gx_total, gx0, Gin, N, Ho, Wo, TH, TW, xh);
    else if (ks == 5) wino_wgrad_xform_x_kernel<4, true><<<nbx, 256, 0, s>>>(x, Gx_total, gx0, Gin, N, Ho, Wo, TH, TW, xh);
    else wino_wgrad_xform_x_kernel<3, true><<<nbx, 256, 0, s>>>(x, Gx_total, gx0, Gin, N, Ho, Wo, TH, TW, xh);
    WgradArgs a;
    int splits;
    wino_wgrad_plan(Cout, 4 * Geff, T, &a.Cout_pad, &a.Kpad128, &splits, &a.pix_per_split);
    float* sync = partial + (size_t)36 * splits * a.Cout_pad * a.Kpad128 + (size_t)36 * Cout * 4 * Geff;   // the sync area of the stream-K form: behind everything else
    a.Cout = Cout; a.M = T;
    const bool sk = wgrad_sk_wanted(a, 36);
    unsigned* zf = sk ? reinterpret_cast<unsigned*>(sync) : nullptr;                                        // its flag words are zeroed by the dY transform
    if (M == 4) wino_wgrad_xform_dy_kernel<4><<<nby, 256, 0, s>>>(dy, Gy_total, gy0, Gout, N, Ho, Wo, TH, TW, yh, zf);
    else wino_wgrad_xform_dy_kernel<3><<<nby, 256, 0, s>>>(dy, Gy_total, gy0, Gout, N, Ho, Wo, TH, TW, yh, zf);
    a.x = xh; a.dy = yh; a.partial = partial; a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    a.N = 36; a.H = 1; a.W = T; a.Ho = 1; a.Wo = T; a.ks = 1; a.stride = 1; a.pad = 0;
    a.Gx_tot = Geff; a.gx0 = 0; a.Gin = Geff; a.Gy_tot = Gout; a.gy0 = 0; a.Cout = Cout;
    a.Kflat = 4 * Geff; a.M = T;
    a.per_image_splits = splits; a.ipp = 1; a.ksx = 1; a.padx = 0; a.N_problems = 36;
    if (sk) {
        const int rc = wgrad_sk_launch(a, 36, sync, false, s);
        if (rc != CNM_OK) return rc;
        splits = 1;                                                       // the tiles are whole: the fused finishing kernel reads them as one split
    } else wgrad_split_launch(a, wgrad_tco(Cout), dim3(a.Cout_pad / wgrad_tco(Cout), a.Kpad128 / 128, 36 * splits), s);
    const unsigned nfin = (unsigned)cnm_ceil_div_ll((long long)Cout * 4 * Geff, 256);
    const int fks = s2 ? ks : 0;
    const float* src = partial;
    if (splits > 4) {
        float* u = partial + (size_t)36 * splits * a.Cout_pad * a.Kpad128;
        wino_wgrad_reduce_kernel<<<(unsigned)cnm_ceil_div_ll(36ll * Cout * 4 * Geff, 256), 256, 0, s>>>(partial, splits, Cout, a.Cout_pad, 4 * Geff, a.Kpad128, u);
        src = u;
    }
    if (splits > 4) {
        if (M == 4) wino_wgrad_finish_kernel<false, 3><<<nfin, 256, 0, s>>>(src, splits, Cout, a.Cout_pad, a.Kpad128, Cin, rot, fks, dw_oihw);
        else wino_wgrad_finish_kernel<false, 4><<<nfin, 256, 0, s>>>(src, splits, Cout, a.Cout_pad, a.Kpad128, Cin, rot, fks, dw_oihw);
    } else {
        if (M == 4) wino_wgrad_finish_kernel<true, 3><<<nfin, 256, 0, s>>>(src, splits, Cout, a.Cout_pad, a.Kpad128, Cin, rot, fks, dw_oihw);
        else wino_wgrad_finish_kernel<true, 4><<<nfin, 256, 0, s>>>(src, splits, Cout, a.Cout_pad, a.Kpad128, Cin, rot, fks, dw_oihw);
    }
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" size_t cnm_conv3x3_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W) { return wino_wgrad_ws(Cout, Cin, N, H, W, 3); }

extern "C" int cnm_conv3x3_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                                 const float* dy, int Gy_total, int gy0, int Cout,
                                                 float* dw_oihw, float* ws, size_t ws_floats,
                                                 int N, int H, int W, int rot, void* stream) {
    return wino_wgrad(x, Gx_total, gx0, Cin, dy, Gy_total, gy0, Cout, dw_oihw, ws, ws_floats, N, H, W, 3, rot, stream);
}

// The stride-2 5x5 / 7x7 layers: the same on the four pixel phases of x (F(4x4,3x3) / F(3x3,4x4) of the 3x3 / 4x4 phase filters, as the
// forward of conv_winograd4s.hip): dY is transformed once, X per phase as 4*Cin channels; H, W = INPUT size (even).
extern "C" size_t cnm_conv_s2_wgrad_winograd_workspace_floats(int Cout, int Cin, int ksize, int N, int H, int W) {
    return (ksize == 5 || ksize == 7) && H % 2 == 0 && W % 2 == 0 ? wino_wgrad_ws(Cout, Cin, N, H, W, ksize) : 0;
}

extern "C" int cnm_conv_s2_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                                 const float* dy, int Gy_total, int gy0, int Cout,
                                                 float* dw_oihw, float* ws, size_t ws_floats,
                                                 int N, int H, int W, int ksize, int rot, void* stream) {
    CNM_REQUIRE(ksize == 5 || ksize == 7, CNM_ERR_BAD_ARG);
    return wino_wgrad(x, Gx_total, gx0, Cin, dy, Gy_total, gy0, Cout, dw_oihw, ws, ws_floats, N, H, W, ksize, rot, stream);
}

// ------------------------------------------------------------------ ... and of the 7x7 stride-1 layer (conv1.0), row-wise F(4,7)
// The forward's own algorithm (conv_winograd_rows.hip: Winograd along image rows, the seven kernel rows stay in the reduction):
//     dW[co][ci][ky][.] = G^T sum_{n, y, t} (A dY[co][n][y][4t .. 4t+3]) (.) (B^T X[ci][n][y + ky - 3][4t - 3 .. 4t + 6])
// -- ten frequency points, each a weight gradient with 7 x 1 taps on [N][H][W/4] "images": 17.5 multiplies per pixel instead of 49.
// Tables: RowWino<74> of conv_winograd_rows.hip (tools/wino1d_matrices.py, points 0, +-1, +-2, +-1/2, +-3/2, inf).
__device__ __constant__ float kWg47BT[10][10] = {
    {9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1, 0},
    {0, -9. / 4, -9. / 4, 169. / 16, 169. / 16, -13. / 2, -13. / 2, 1, 1, 0}, {0, 9. / 4, -9. / 4, -169. / 16, 169. / 16, 13. / 2, -13. / 2, -1, 1, 0},
    {0, -9. / 8, -9. / 16, 49. / 8, 49. / 16, -7, -7. / 2, 2, 1, 0}, {0, 9. / 8, -9. / 16, -49. / 8, 49. / 16, 7, -7. / 2, -2, 1, 0},
    {0, -9. / 2, -9, 61. / 8, 61. / 4, -29. / 8, -29. / 4, 1. / 2, 1, 0}, {0, 9. / 2, -9, -61. / 8, 61. / 4, 29. / 8, -29. / 4, -1. / 2, 1, 0},
    {0, -3. / 2, -1, 63. / 8, 21. / 4, -63. / 8, -21. / 4, 3. / 2, 1, 0}, {0, 3. / 2, -1, -63. / 8, 21. / 4, 63. / 8, -21. / 4, -3. / 2, 1, 0},
    {0, 9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1}};
__device__ __constant__ float kWg47A[10][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {1, 1. / 2, 1. / 4, 1. / 8},
                                               {1, -1. / 2, 1. / 4, -1. / 8}, {1, 3. / 2, 9. / 4, 27. / 8}, {1, -3. / 2, 9. / 4, -27. / 8}, {0, 0, 0, 1}};
__device__ __constant__ double kWg47G[10][7] = {
    {4. / 9, 0, 0, 0, 0, 0, 0},
    {8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45}, {8. / 45, -8. / 45, 8. / 45, -8. / 45, 8. / 45, -8. / 45, 8. / 45},
    {2. / 315, 4. / 315, 8. / 315, 16. / 315, 32. / 315, 64. / 315, 128. / 315}, {2. / 315, -4. / 315, 8. / 315, -16. / 315, 32. / 315, -64. / 315, 128. / 315},
    {-16. / 45, -8. / 45, -4. / 45, -2. / 45, -1. / 45, -1. / 90, -1. / 180}, {-16. / 45, 8. / 45, -4. / 45, 2. / 45, -1. / 45, 1. / 90, -1. / 180},
    {-16. / 315, -8. / 105, -4. / 35, -6. / 35, -9. / 35, -27. / 70, -81. / 140}, {-16. / 315, 8. / 105, -4. / 35, 6. / 35, -9. / 35, 27. / 70, -81. / 140},
    {0, 0, 0, 0, 0, 0, 1}};

// F(4,5) (conv2.0, 5x5 stride 1): eight points 0, +-1, +-2, +-1/2, inf -- 10 multiplies per pixel instead of 25.
__device__ __constant__ float kWg45BT[8][8] = {{-1, 0, 21. / 4, 0, -21. / 4, 0, 1, 0}, {0, 1, 1, -17. / 4, -17. / 4, 1, 1, 0}, {0, -1, 1, 17. / 4, -17. / 4, -1, 1, 0},
                                               {0, 1. / 2, 1. / 4, -5. / 2, -5. / 4, 2, 1, 0}, {0, -1. / 2, 1. / 4, 5. / 2, -5. / 4, -2, 1, 0},
                                               {0, 2, 4, -5. / 2, -5, 1. / 2, 1, 0}, {0, -2, 4, 5. / 2, -5, -1. / 2, 1, 0}, {0, -1, 0, 21. / 4, 0, -21. / 4, 0, 1}};
__device__ __constant__ float kWg45A[8][4] = {{1, 0, 0, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {1, 2, 4, 8}, {1, -2, 4, -8}, {1, 1. / 2, 1. / 4, 1. / 8}, {1, -1. / 2, 1. / 4, -1. / 8}, {0, 0, 0, 1}};
__device__ __constant__ double kWg45G[8][5] = {{-1, 0, 0, 0, 0}, {-2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9}, {-2. / 9, 2. / 9, -2. / 9, 2. / 9, -2. / 9},
                                               {1. / 90, 1. / 45, 2. / 45, 4. / 45, 8. / 45}, {1. / 90, -1. / 45, 2. / 45, -4. / 45, 8. / 45},
                                               {32. / 45, 16. / 45, 8. / 45, 4. / 45, 2. / 45}, {32. / 45, -16. / 45, 8. / 45, -4. / 45, 2. / 45}, {0, 0, 0, 0, 1}};

// out[(xi * N + n)][g][y][t] = sum_j Mx[xi][j] in[n][g][y][4 t - off + j]: DY: Mx = A (4 pixels of the tile), else Mx = B^T (the (R + 3)-pixel
// window, off = R / 2).  One thread per (tile, row, image, group), tiles fastest.  R = 7: F(4,7), 10 points; R = 5: F(4,5), 8 points.
template <bool DY, int R>
__global__ __launch_bounds__(256) void wino_wgrad_rows_xform_kernel(const float* __restrict__ in, int G_tot, int g0, int G, int N, int H, int W, int TW,
                                                                    float* __restrict__ out) {
    constexpr int NP = R + 3, NL = DY ? 4 : NP, OFF = DY ? 0 : R / 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)TW * H * N * G;
    if (idx >= total) return;
    const int t = (int)(idx % TW); long long r = idx / TW;
    const int y = (int)(r % H); r /= H;
    const int n = (int)(r % N), g = (int)(r / N);
    const wg_f4* row = reinterpret_cast<const wg_f4*>(in + c4_offset(n, G_tot, g0 + g, H * W, y * W));
    wg_f4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int ix = 4 * t - OFF + j;
        v[j] = (unsigned)ix < (unsigned)W ? row[ix] : wg_f4{0.f, 0.f, 0.f, 0.f};
    }
    const size_t plane = (size_t)N * G * H * TW;                          // float4 elements per frequency point
    wg_f4* o = reinterpret_cast<wg_f4*>(out) + ((size_t)(n * G + g) * H + y) * TW + t;
#pragma unroll
    for (int xi = 0; xi < NP; ++xi) {
        wg_f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const float c = R == 7 ? (DY ? kWg47A[xi][j] : kWg47BT[xi][j]) : (DY ? kWg45A[xi][j] : kWg45BT[xi][j]);
            if (c != 0.f) s += c * v[j];
        }
        o[(size_t)xi * plane] = s;
    }
}

// dW[co][ci][ky][kx] = sum_xi G[xi][kx] dU[xi][co][ky * Cp + cp]   (u: [R + 3][Cout][R * Cp], split sums done by wino_wgrad_reduce_kernel)
template <int R>
__global__ __launch_bounds__(256) void wino_wgrad_rows_finish_kernel(const float* __restrict__ u, int Cout, int Cin, int rot, float* __restrict__ dw) {
    constexpr int NP = R + 3;
    const int Cp = 4 * ((Cin + 3) / 4), K7 = R * Cp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)Cout * K7;
    if (idx >= per) return;
    const int co = (int)(idx / K7), k = (int)(idx - (long long)co * K7), ky = k / Cp, cp = k - ky * Cp;
    if (cp >= Cin) return;
    double uu[NP];
#pragma unroll
    for (int xi = 0; xi < NP; ++xi) uu[xi] = (double)u[(size_t)xi * per + idx];
    float* o = dw + (((size_t)co * Cin + (cp + rot) % Cin) * R + ky) * R;
#pragma unroll
    for (int kx = 0; kx < R; ++kx) {
        double s = 0.0;
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) s += (R == 7 ? kWg47G[xi][kx] : kWg45G[xi][kx]) * uu[xi];
        o[kx] = (float)s;
    }
}

static void wino_rows_plan(int Cout, int Cin, int R, long long Mpix, int* Cout_pad, int* Kpad128, int* splits, int* pps) {
    const int tco = wgrad_tco(Cout);
    *Cout_pad = wg_round(Cout, tco);
    *Kpad128 = wg_round(R * 4 * ((Cin + 3) / 4), 128);
    const int tiles = (R + 3) * (*Cout_pad / tco) * (*Kpad128 / 128);
    int s = (CNM_WGRAD_WORKGROUPS + tiles - 1) / tiles;
    const int maxs = (int)((Mpix + 255) / 256);
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    *pps = wg_round((int)((Mpix + s - 1) / s), 16);
    *splits = (int)((Mpix + *pps - 1) / *pps);
}

static size_t wino_rows_ws(int Cout, int Cin, int R, int N, int H, int W) {
    if (Cout <= 0 || Cin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
    const size_t TW = (size_t)(W + 3) / 4, P = (size_t)N * H * TW, NP = R + 3;
    int cp, kp, sp, pps;
    wino_rows_plan(Cout, Cin, R, (long long)P, &cp, &kp, &sp, &pps);
    return NP * P * 4 * (size_t)((Cin + 3) / 4) + NP * P * 4 * (size_t)((Cout + 3) / 4) + NP * sp * (size_t)cp * kp + NP * Cout * (size_t)R * 4 * ((Cin + 3) / 4) + wgrad_sk_sync_floats();
}

template <int R>
static int wino_rows_wgrad(const float* x, int Gx_total, int gx0, int Cin, const float* dy, int Gy_total, int gy0, int Cout,
                           float* dw_oihw, float* ws, size_t ws_floats, int N, int H, int W, int rot, void* stream) {
    constexpr int NP = R + 3;
    CNM_REQUIRE(x && dy && dw_oihw && ws && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % 4 == 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    const int Gin = (Cin + 3) / 4, Gout = Cout / 4, TW = (W + 3) / 4;
    const long long P = (long long)N * H * TW;
    CNM_REQUIRE(gx0 >= 0 && gx0 + Gin <= Gx_total && gy0 >= 0 && gy0 + Gout <= Gy_total && P < (1ll << 24), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(wino_rows_ws(Cout, Cin, R, N, H, W) <= ws_floats, CNM_ERR_WORKSPACE);
    float* xh = ws; float* yh = xh + (size_t)NP * P * 4 * Gin; float* partial = yh + (size_t)NP * P * 4 * Gout;
    const unsigned long long xb = (unsigned long long)NP * P * Gin * 16ull, yb = (unsigned long long)NP * P * Gout * 16ull;
    CNM_REQUIRE(xb < 0xFFFFFFFFull && yb < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    hipStream_t s = cnm_stream(stream);
    wino_wgrad_rows_xform_kernel<false, R><<<(unsigned)cnm_ceil_div_ll(P * Gin, 256), 256, 0, s>>>(x, Gx_total, gx0, Gin, N, H, W, TW, xh);
    wino_wgrad_rows_xform_kernel<true, R><<<(unsigned)cnm_ceil_div_ll(P * Gout, 256), 256, 0, s>>>(dy, Gy_total, gy0, Gout, N, H, W, TW, yh);
    WgradArgs a;
    a.x = xh; a.dy = yh; a.partial = partial; a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    a.N = NP * N; a.H = H; a.W = TW; a.Ho = H; a.Wo = TW; a.ks = R; a.stride = 1; a.pad = R / 2; a.ksx = 1; a.padx = 0; a.ipp = N;
    a.Gx_tot = Gin; a.gx0 = 0; a.Gin = Gin; a.Gy_tot = Gout; a.gy0 = 0; a.Cout = Cout;
    a.Kflat = R * 4 * Gin; a.M = (int)P;
    int splits;
    wino_rows_plan(Cout, Cin, R, P, &a.Cout_pad, &a.Kpad128, &splits, &a.pix_per_split);
    a.per_image_splits = splits; a.N_problems = NP;
    float* u = partial + (size_t)NP * splits * a.Cout_pad * a.Kpad128;
    const int KR = R * 4 * Gin;
    if (wgrad_sk_wanted(a, NP)) {
        const int rc = wgrad_sk_launch(a, NP, u + (size_t)NP * Cout * KR, true, s);
        if (rc != CNM_OK) return rc;
        splits = 1;                                                       // whole tiles: the reduction kernel below only repacks them
    } else wgrad_split_launch(a, wgrad_tco(Cout), dim3(a.Cout_pad / wgrad_tco(Cout), a.Kpad128 / 128, NP * splits), s);
    wino_wgrad_reduce_kernel<<<(unsigned)cnm_ceil_div_ll((long long)NP * Cout * KR, 256), 256, 0, s>>>(partial, splits, Cout, a.Cout_pad, KR, a.Kpad128, u, NP);
    wino_wgrad_rows_finish_kernel<R><<<(unsigned)cnm_ceil_div_ll((long long)Cout * KR, 256), 256, 0, s>>>(u, Cout, Cin, rot, dw_oihw);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" size_t cnm_conv7x7_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W) { return wino_rows_ws(Cout, Cin, 7, N, H, W); }
extern "C" int cnm_conv7x7_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                                 const float* dy, int Gy_total, int gy0, int Cout,
                                                 float* dw_oihw, float* ws, size_t ws_floats,
                                                 int N, int H, int W, int rot, void* stream) {
    return wino_rows_wgrad<7>(x, Gx_total, gx0, Cin, dy, Gy_total, gy0, Cout, dw_oihw, ws, ws_floats, N, H, W, rot, stream);
}
// ... and of the 5x5 stride-1 layer (conv2.0) on F(4,5): eight gradients with 5 x 1 taps, 10 multiplies per pixel instead of 25
extern "C" size_t cnm_conv5x5_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W) { return wino_rows_ws(Cout, Cin, 5, N, H, W); }
extern "C" int cnm_conv5x5_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                                 const float* dy, int Gy_total, int gy0, int Cout,
                                                 float* dw_oihw, float* ws, size_t ws_floats,
                                                 int N, int H, int W, int rot, void* stream) {
    return wino_rows_wgrad<5>(x, Gx_total, gx0, Cin, dy, Gy_total, gy0, Cout, dw_oihw, ws, ws_floats, N, H, W, rot, stream);
}

// ------------------------------------------------------------------ BatchNorm2d (train mode) + ReLU on c4
// stats[c] = (sum, sum of squares) in fp64; mean/invstd derived from them (biased variance for the
// normalisation, unbiased for the running update -- torch.nn.BatchNorm2d defaults).
// S statistics groups (blockIdx.z): sample n belongs to group n % S -- the S sources of a frame processed as ONE batch keep the
// batch statistics of S separate forward calls (reference train.py:164-167 calls depthNet once per source).  S = 1: plain BatchNorm.
// Grids [r5]: the reducing kernels run (group g, image x pixel-chunk, statistics group), the elementwise ones (pixel-chunk, plane n G + g):
// a workgroup stays inside ONE c4 plane, its four channels' parameters are workgroup constants and the loops are free of the 64-bit
// divisions (flat index -> plane, pixel) and per-element fp64 divisions the round-4 kernels paid per float4 -- those, not memory, bound them
// (45 % of the HBM roof over a training step's 76 layers).
struct BnGridY { int ypi, ni; };                                          // pixel chunks per image, images side by side in grid.y
static BnGridY bn_grid_y(int Ng, int HW) {
    BnGridY r;
    r.ypi = (HW + 1023) / 1024; r.ypi = r.ypi < 1 ? 1 : (r.ypi > 16 ? 16 : r.ypi);
    r.ni = 64 / r.ypi; r.ni = r.ni > Ng ? Ng : r.ni; if (r.ni < 1) r.ni = 1;
    return r;
}
__device__ __forceinline__ void bn_block_sums(double (&s)[4], double (&q)[4], double* __restrict__ out, int g) {
    __shared__ double red[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s[j] += __shfl_down(s[j], o); q[j] += __shfl_down(q[j], o); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int j = 0; j < 4; ++j) { red[wave][j] = s[j]; red[wave + 4][j] = q[j]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x;
        atomicAdd(&out[(4 * g + j) * 2 + 0], red[0][j] + red[1][j] + red[2][j] + red[3][j]);
        atomicAdd(&out[(4 * g + j) * 2 + 1], red[4][j] + red[5][j] + red[6][j] + red[7][j]);
    }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int G, int HW, double* __restrict__ stats, int S, int ypi, int ni) {
    const int g = blockIdx.x, grp = blockIdx.z;
    const int i0 = blockIdx.y / ypi, chunk = blockIdx.y - i0 * ypi, step = ypi * 256;
    stats += (size_t)grp * 8 * G;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int n = grp + i0 * S; n < N; n += ni * S) {                       // images of this statistics group, ni of them side by side
        const float4* __restrict__ p = reinterpret_cast<const float4*>(x + c4_offset(n, G, g, HW, 0));
        int pix = chunk * 256 + threadIdx.x;
        for (; pix + step < HW; pix += 2 * step) {                         // two loads in flight
            const float4 v = p[pix], w = p[pix + step];
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
            s[0] += w.x; s[1] += w.y; s[2] += w.z; s[3] += w.w;
            q[0] += (double)w.x * w.x; q[1] += (double)w.y * w.y; q[2] += (double)w.z * w.z; q[3] += (double)w.w * w.w;
        }
        if (pix < HW) {
            const float4 v = p[pix];
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    }
    bn_block_sums(s, q, stats, g);
}

// rezero: the sums are cleared again once read (the "zero on entry, left zero" workspace of the *_z entry points: no clearing
// launch per layer); tracked: nn.BatchNorm2d.num_batches_tracked, incremented here instead of by a launch of its own
__global__ void bn_finalize_kernel(double* __restrict__ stats, int C, double count, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int rezero, long long* __restrict__ tracked,
                                   int S, int N, int HW) {
    // count: unused with groups (kept for the signature); group grp holds ceil((N - grp) / S) samples.  The running statistics take the
    // groups' updates one after the other, in source order, as S forward calls would
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && tracked) *tracked += S;
    const int Cp = 4 * ((C + 3) / 4);
    for (int grp = 0; grp < S; ++grp) {
        double* st = stats + (size_t)grp * 2 * Cp;
        if (c >= C) {
            if (rezero && c < Cp) { st[2 * c] = 0.0; st[2 * c + 1] = 0.0; }   // the padding channels of the last group
            continue;
        }
        const double cnt = S == 1 ? count : (double)((N - grp + S - 1) / S) * HW;
        const double mu = st[2 * c] / cnt;
        double var = st[2 * c + 1] / cnt - mu * mu;
        if (rezero) { st[2 * c] = 0.0; st[2 * c + 1] = 0.0; }
        if (var < 0) var = 0;
        mean[grp * C + c] = (float)mu; invstd[grp * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = cnt > 1 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

// y before the ReLU, in ONE pinned operation order: the backward kernels recompute the ReLU mask from x with the same function
// (r > 0 <=> y > 0 bit for bit), so they need not read y -- two of the seven tensor passes of a BatchNorm backward
__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
#pragma clang fp contract(off)
    const float xh = (x - mean) * invstd;
    return __builtin_fmaf(xh, gamma, beta);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int C, int relu,
                                                       float* __restrict__ y, int N, int G, int HW, int S) {
    const int plane = blockIdx.y, n = plane / G, g = plane - n * G, so = S == 1 ? 0 : (n % S) * C;   // statistics of the sample's group
    float mu[4], is[4], ga[4], be[4]; bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; live[j] = c < C;
        mu[j] = live[j] ? mean[so + c] : 0.f; is[j] = live[j] ? invstd[so + c] : 0.f; ga[j] = live[j] ? gamma[c] : 0.f; be[j] = live[j] ? beta[c] : 0.f;
    }
    const float4* __restrict__ p = reinterpret_cast<const float4*>(x) + (size_t)plane * HW;
    float4* __restrict__ o = reinterpret_cast<float4*>(y) + (size_t)plane * HW;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const float4 v = p[pix];
        const float in[4] = {v.x, v.y, v.z, v.w}; float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = live[j] ? bn_affine(in[j], mu[j], is[j], ga[j], be[j]) : 0.f;
            if (relu && live[j]) r[j] = fmaxf(r[j], 0.f);
        }
        o[pix] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// backward pass 1: sum_dy[c], sum_dy_xhat[c] (dy masked by the ReLU of the forward output y)
// RECOMP: the ReLU mask from x (bn_affine(x) > 0) instead of from the saved output y (y == nullptr then)
template <bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int C, int relu,
                                                            int N, int G, int HW, double* __restrict__ sums, int S, int ypi, int ni) {
    const int g = blockIdx.x, grp = blockIdx.z;
    const int i0 = blockIdx.y / ypi, chunk = blockIdx.y - i0 * ypi, step = ypi * 256;
    sums += (size_t)grp * 8 * G;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; mu[j] = c < C ? mean[grp * C + c] : 0.f; is[j] = c < C ? invstd[grp * C + c] : 0.f;
        ga[j] = (RECOMP && c < C) ? gamma[c] : 0.f; be[j] = (RECOMP && c < C) ? beta[c] : 0.f;
    }
    for (int n = grp + i0 * S; n < N; n += ni * S) {
        const size_t base = c4_offset(n, G, g, HW, 0);
        const float4* __restrict__ px = reinterpret_cast<const float4*>(x + base);
        const float4* __restrict__ py = RECOMP ? nullptr : reinterpret_cast<const float4*>(y + base);
        const float4* __restrict__ pd = reinterpret_cast<const float4*>(dy + base);
        for (int pix = chunk * 256 + threadIdx.x; pix < HW; pix += step) {
            const float4 xv = px[pix];
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (!RECOMP) yv = py[pix];
            const float4 dv = pd[pix];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pre = RECOMP ? bn_affine(xs[j], mu[j], is[j], ga[j], be[j]) : ys[j];
                const float d = (relu && !(pre > 0.f)) ? 0.f : ds[j];
                s[j] += d; q[j] += (double)d * ((xs[j] - mu[j]) * is[j]);
            }
        }
    }
    bn_block_sums(s, q, sums, g);
}

// backward pass 2: dx = gamma*invstd*(dy - sum_dy/m - xhat*sum_dy_xhat/m); dgamma = sum_dy_xhat; dbeta = sum_dy
template <bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const double* __restrict__ sums, double count, int C, int relu,
                                                           float* __restrict__ dx, int N, int G, int HW, int S) {
    const int plane = blockIdx.y, n = plane / G, g = plane - n * G, grp = S == 1 ? 0 : n % S, so = grp * C;
    const double* sg = sums + (size_t)grp * 8 * G;
    const double cnt = S == 1 ? count : (double)((N - grp + S - 1) / S) * HW;
    float mu[4], is[4], ga[4], be[4], sd[4], sq[4]; bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; live[j] = c < C;
        mu[j] = live[j] ? mean[so + c] : 0.f; is[j] = live[j] ? invstd[so + c] : 0.f; ga[j] = live[j] ? gamma[c] : 0.f; be[j] = (RECOMP && live[j]) ? beta[c] : 0.f;
        sd[j] = live[j] ? (float)(sg[2 * c] / cnt) : 0.f; sq[j] = live[j] ? (float)(sg[2 * c + 1] / cnt) : 0.f;
    }
    const size_t base = (size_t)plane * HW;
    const float4* __restrict__ px = reinterpret_cast<const float4*>(x) + base;
    const float4* __restrict__ py = RECOMP ? nullptr : reinterpret_cast<const float4*>(y) + base;
    const float4* __restrict__ pd = reinterpret_cast<const float4*>(dy) + base;
    float4* __restrict__ po = reinterpret_cast<float4*>(dx) + base;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const float4 xv = px[pix];
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (!RECOMP) yv = py[pix];
        const float4 dv = pd[pix];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pre = RECOMP ? bn_affine(xs[j], mu[j], is[j], ga[j], be[j]) : ys[j];
            const float d = (relu && !(pre > 0.f)) ? 0.f : ds[j];
            const float xh = (xs[j] - mu[j]) * is[j];
            o[j] = live[j] ? ga[j] * is[j] * (d - sd[j] - xh * sq[j]) : 0.f;
        }
        po[pix] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void bn_param_grad_kernel(double* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta, int rezero, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, Cp = 4 * ((C + 3) / 4);
    if (c >= Cp) return;
    float db = 0.f, dg = 0.f;                                               // the groups' parameter gradients add in source order, in fp32, as S backward passes accumulate them
    for (int grp = 0; grp < S; ++grp) {
        double* sg = sums + (size_t)grp * 2 * Cp;
        db += (float)sg[2 * c]; dg += (float)sg[2 * c + 1];
        if (rezero) { sg[2 * c] = 0.0; sg[2 * c + 1] = 0.0; }
    }
    if (c < C) { dbeta[c] = db; dgamma[c] = dg; }
}

// The fp64 sum buffers are cleared by a kernel, not hipMemsetAsync: captured into a HIP graph (TrainStepWoNormal(graph=True))
// the memset node took effect on the first replay only -- later replays accumulated onto the previous sums.
__global__ void bn_zero_kernel(double* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

static dim3 bn_plane_grid(int N, int G, int HW) {                         // elementwise kernels: (pixel chunks of 1024, planes)
    int cx = (HW + 1023) / 1024; cx = cx < 1 ? 1 : (cx > 48 ? 48 : cx);
    return dim3(cx, N * G);
}

static int bn_forward(const float* x, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, int relu,
                      float* y, float* save_mean, float* save_invstd, double* stats_ws, int zeroed, long long* tracked,
                      int N, int C, int H, int W, void* stream, int S = 1) {
    CNM_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && stats_ws && N > 0 && C > 0 && H > 0 && W > 0 && S >= 1 && S <= N, CNM_ERR_BAD_ARG);
    const int G = (C + 3) / 4, HW = H * W;
    CNM_REQUIRE((long long)N * G <= 65535, CNM_ERR_BAD_ARG);                // planes in grid.y
    hipStream_t s = cnm_stream(stream);
    if (!zeroed) bn_zero_kernel<<<cnm_ceil_div(8 * G * S, 256), 256, 0, s>>>(stats_ws, 8 * G * S);
    const BnGridY gy = bn_grid_y((N + S - 1) / S, HW);
    bn_stats_kernel<<<dim3(G, gy.ypi * gy.ni, S), 256, 0, s>>>(x, N, G, HW, stats_ws, S, gy.ypi, gy.ni);
    bn_finalize_kernel<<<cnm_ceil_div(4 * G, 256), 256, 0, s>>>(stats_ws, C, (double)N * HW, eps, momentum, save_mean, save_invstd, running_mean, running_var, zeroed, tracked, S, N, HW);
    bn_apply_kernel<<<bn_plane_grid(N, G, HW), 256, 0, s>>>(x, save_mean, save_invstd, gamma, beta, C, relu, y, N, G, HW, S);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_bn_train_forward_c4_f32(const float* x, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps, int relu,
                                           float* y, float* save_mean, float* save_invstd, double* stats_ws,
                                           int N, int C, int H, int W, void* stream) {
    return bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, relu, y, save_mean, save_invstd, stats_ws, 0, nullptr, N, C, H, W, stream);
}
// The same with a workspace that is ZERO when the call starts and left zero by it (one per stream, any number of layers), and
// with num_batches_tracked (int64 scalar on the device, may be NULL) incremented by the call: two launches less per layer.
extern "C" int cnm_bn_train_forward_z_c4_f32(const float* x, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, float momentum, float eps, int relu,
                                             float* y, float* save_mean, float* save_invstd, double* zero_ws, long long* num_batches_tracked,
                                             int N, int C, int H, int W, void* stream) {
    return bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, relu, y, save_mean, save_invstd, zero_ws, 1, num_batches_tracked, N, C, H, W, stream);
}

// y == nullptr (with beta): the ReLU mask is recomputed from x -- y is then neither read nor needed by the caller's tape
static int bn_backward(const float* x, const float* y, const float* dy, const float* gamma,
                       const float* save_mean, const float* save_invstd, int relu,
                       float* dx, float* dgamma, float* dbeta, double* sums_ws, int zeroed,
                       int N, int C, int H, int W, void* stream, int S = 1, const float* beta = nullptr) {
    CNM_REQUIRE(x && (y || beta || !relu) && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && sums_ws && N > 0 && C > 0 && S >= 1 && S <= N, CNM_ERR_BAD_ARG);
    const int G = (C + 3) / 4, HW = H * W;
    CNM_REQUIRE((long long)N * G <= 65535, CNM_ERR_BAD_ARG);                // planes in grid.y
    hipStream_t s = cnm_stream(stream);
    if (!zeroed) bn_zero_kernel<<<cnm_ceil_div(8 * G * S, 256), 256, 0, s>>>(sums_ws, 8 * G * S);
    const bool recomp = !y && relu;
    const BnGridY gy = bn_grid_y((N + S - 1) / S, HW);
    const dim3 rg(G, gy.ypi * gy.ni, S);
    if (recomp) bn_bwd_reduce_kernel<true><<<rg, 256, 0, s>>>(x, nullptr, dy, save_mean, save_invstd, gamma, beta, C, relu, N, G, HW, sums_ws, S, gy.ypi, gy.ni);
    else bn_bwd_reduce_kernel<false><<<rg, 256, 0, s>>>(x, y ? y : x, dy, save_mean, save_invstd, gamma, beta, C, relu, N, G, HW, sums_ws, S, gy.ypi, gy.ni);
    const dim3 ag = bn_plane_grid(N, G, HW);
    if (recomp) bn_bwd_apply_kernel<true><<<ag, 256, 0, s>>>(x, nullptr, dy, save_mean, save_invstd, gamma, beta, sums_ws, (double)N * HW, C, relu, dx, N, G, HW, S);
    else bn_bwd_apply_kernel<false><<<ag, 256, 0, s>>>(x, y ? y : x, dy, save_mean, save_invstd, gamma, beta, sums_ws, (double)N * HW, C, relu, dx, N, G, HW, S);
    bn_param_grad_kernel<<<cnm_ceil_div(4 * G, 256), 256, 0, s>>>(sums_ws, C, dgamma, dbeta, zeroed, S);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_bn_train_backward_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                            const float* save_mean, const float* save_invstd, int relu,
                                            float* dx, float* dgamma, float* dbeta, double* sums_ws,
                                            int N, int C, int H, int W, void* stream) {
    return bn_backward(x, y, dy, gamma, save_mean, save_invstd, relu, dx, dgamma, dbeta, sums_ws, 0, N, C, H, W, stream);
}
extern "C" int cnm_bn_train_backward_z_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                              const float* save_mean, const float* save_invstd, int relu,
                                              float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                              int N, int C, int H, int W, void* stream) {
    return bn_backward(x, y, dy, gamma, save_mean, save_invstd, relu, dx, dgamma, dbeta, zero_ws, 1, N, C, H, W, stream);
}

// The same with `groups` statistics groups: sample n is normalised with the batch statistics of the samples n' = n (mod groups) -- a
// batch that interleaves the `groups` sources of every frame (pair p = b * groups + s) then computes exactly what `groups` separate
// calls, one per source, compute (reference train.py:164-167), running statistics and num_batches_tracked updated `groups` times in
// source order.  save_mean / save_invstd: groups * C floats; zero_ws: 8 * ceil(C / 4) * groups doubles.
extern "C" int cnm_bn_train_forward_zg_c4_f32(const float* x, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, float momentum, float eps, int relu,
                                              float* y, float* save_mean, float* save_invstd, double* zero_ws, long long* num_batches_tracked,
                                              int N, int C, int H, int W, int groups, void* stream) {
    return bn_forward(x, gamma, beta, running_mean, running_var, momentum, eps, relu, y, save_mean, save_invstd, zero_ws, 1, num_batches_tracked, N, C, H, W, stream, groups);
}
extern "C" int cnm_bn_train_backward_zg_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                               const float* save_mean, const float* save_invstd, int relu,
                                               float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                               int N, int C, int H, int W, int groups, void* stream) {
    return bn_backward(x, y, dy, gamma, save_mean, save_invstd, relu, dx, dgamma, dbeta, zero_ws, 1, N, C, H, W, stream, groups);
}

// The grouped backward WITHOUT the saved output: the ReLU mask is recomputed from x, gamma, beta and the saved statistics with
// the forward's own operation order (bit-identical to the y-reading form); the tape then keeps x only.
extern "C" int cnm_bn_train_backward_zgb_c4_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                                const float* save_mean, const float* save_invstd, int relu,
                                                float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                                int N, int C, int H, int W, int groups, void* stream) {
    CNM_REQUIRE(beta, CNM_ERR_BAD_ARG);
    return bn_backward(x, nullptr, dy, gamma, save_mean, save_invstd, relu, dx, dgamma, dbeta, zero_ws, 1, N, C, H, W, stream, groups, beta);
}

// ------------------------------------------------------------------ BatchNorm without the finalising launches [r6]
// The *_p entry points: the reductions write one PARTIAL per workgroup into fixed slots ([group][chunk][channel][2] fp64: no atomics, nothing
// to clear, a fixed summation order), and the elementwise pass that follows sums the <= 64 partials of its own four channels in its prologue
// (one wave, four loads per lane, a shuffle tree: the same tree in every workgroup, so every workgroup -- and the one that records
// save_mean / save_invstd / the running statistics / dgamma / dbeta -- holds bit-identical statistics).  Two launches per direction instead
// of three: 76 launches of ~5 us less per training step; and the sums no longer depend on the order atomics happened to land in.
// (Round 5 folded the finalisation with last-workgroup TICKETS and lost: a ticket per workgroup is dearer than a launch.  Here nobody
// waits for anybody.)
__device__ __forceinline__ void bn_block_partials(double (&s)[4], double (&q)[4], double* __restrict__ slot) {   // slot: [4 channels of this group][2]
    __shared__ double red[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s[j] += __shfl_down(s[j], o); q[j] += __shfl_down(q[j], o); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int j = 0; j < 4; ++j) { red[wave][j] = s[j]; red[wave + 4][j] = q[j]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x;
        slot[2 * j + 0] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
        slot[2 * j + 1] = red[4][j] + red[5][j] + red[6][j] + red[7][j];
    }
}
// sum of the Y partials of channel group g, statistics group grp: every lane of wave 0 returns (sum0, sum1) of channel 4 g + (lane & 3)
__device__ __forceinline__ void bn_sum_partials(const double* __restrict__ part, int Y, int Cp, int grp, int g, double& a, double& b) {
    const int lane = threadIdx.x & 63, j = lane & 3;
    a = 0.0; b = 0.0;
    for (int y = lane >> 2; y < Y; y += 16) {
        const double* p = part + (((size_t)grp * Y + y) * Cp + 4 * g + j) * 2;
        a += p[0]; b += p[1];
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
}

__global__ __launch_bounds__(256) void bn_stats_p_kernel(const float* __restrict__ x, int N, int G, int HW, double* __restrict__ part, int S, int ypi, int ni) {
    const int g = blockIdx.x, grp = blockIdx.z;
    const int i0 = blockIdx.y / ypi, chunk = blockIdx.y - i0 * ypi, step = ypi * 256;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int n = grp + i0 * S; n < N; n += ni * S) {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(x + c4_offset(n, G, g, HW, 0));
        int pix = chunk * 256 + threadIdx.x;
        for (; pix + step < HW; pix += 2 * step) {
            const float4 v = p[pix], w = p[pix + step];
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
            s[0] += w.x; s[1] += w.y; s[2] += w.z; s[3] += w.w;
            q[0] += (double)w.x * w.x; q[1] += (double)w.y * w.y; q[2] += (double)w.z * w.z; q[3] += (double)w.w * w.w;
        }
        if (pix < HW) {
            const float4 v = p[pix];
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    }
    bn_block_partials(s, q, part + (((size_t)grp * gridDim.y + blockIdx.y) * 4 * G + 4 * g) * 2);
}

__device__ __forceinline__ void bn_moments(double sum, double sumsq, double cnt, float eps, float& mu_f, float& is_f, double& mu, double& var) {
    mu = sum / cnt; var = sumsq / cnt - mu * mu;
    if (var < 0) var = 0;
    mu_f = (float)mu; is_f = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void bn_apply_p_kernel(const float* __restrict__ x, const double* __restrict__ part, int Y,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int C, int relu,
                                                         float* __restrict__ y, int N, int G, int HW, int S, float eps, float momentum,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ tracked) {
    __shared__ float sh[8];
    const int plane = blockIdx.y, n = plane / G, g = plane - n * G, grp = S == 1 ? 0 : n % S, Cp = 4 * G;
    if (threadIdx.x < 64) {
        const int j = threadIdx.x & 3, c = 4 * g + j;
        double a, b, mu, var; float mf, isf;
        bn_sum_partials(part, Y, Cp, grp, g, a, b);
        bn_moments(a, b, (double)((N - grp + S - 1) / S) * HW, eps, mf, isf, mu, var);
        if (threadIdx.x < 4) { sh[j] = mf; sh[4 + j] = isf; }
        if (blockIdx.x == 0 && n == grp && threadIdx.x < 4 && c < C) { save_mean[grp * C + c] = mf; save_invstd[grp * C + c] = isf; }
        if (blockIdx.x == 0 && n == 0) {                                  // the running statistics take the groups' updates one after the other, in source order
            if (g == 0 && threadIdx.x == 0 && tracked) *tracked += S;
            float rm = 0.f, rv = 0.f;
            if (running_mean && c < C) { rm = running_mean[c]; rv = running_var[c]; }
            for (int k = 0; k < S; ++k) {
                if (k != grp || S > 1) { bn_sum_partials(part, Y, Cp, k, g, a, b); }
                const double cnt = (double)((N - k + S - 1) / S) * HW;
                bn_moments(a, b, cnt, eps, mf, isf, mu, var);
                const double unbiased = cnt > 1 ? var * cnt / (cnt - 1.0) : var;
                rm = (float)((1.0 - momentum) * rm + momentum * mu);
                rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
            }
            if (running_mean && threadIdx.x < 4 && c < C) { running_mean[c] = rm; running_var[c] = rv; }
        }
    }
    __syncthreads();
    float mu[4], is[4], ga[4], be[4]; bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; live[j] = c < C;
        mu[j] = live[j] ? sh[j] : 0.f; is[j] = live[j] ? sh[4 + j] : 0.f; ga[j] = live[j] ? gamma[c] : 0.f; be[j] = live[j] ? beta[c] : 0.f;
    }
    const float4* __restrict__ p = reinterpret_cast<const float4*>(x) + (size_t)plane * HW;
    float4* __restrict__ o = reinterpret_cast<float4*>(y) + (size_t)plane * HW;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const float4 v = p[pix];
        const float in[4] = {v.x, v.y, v.z, v.w}; float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = live[j] ? bn_affine(in[j], mu[j], is[j], ga[j], be[j]) : 0.f;
            if (relu && live[j]) r[j] = fmaxf(r[j], 0.f);
        }
        o[pix] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

template <bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_p_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int C, int relu,
                                                              int N, int G, int HW, double* __restrict__ part, int S, int ypi, int ni) {
    const int g = blockIdx.x, grp = blockIdx.z;
    const int i0 = blockIdx.y / ypi, chunk = blockIdx.y - i0 * ypi, step = ypi * 256;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; mu[j] = c < C ? mean[grp * C + c] : 0.f; is[j] = c < C ? invstd[grp * C + c] : 0.f;
        ga[j] = (RECOMP && c < C) ? gamma[c] : 0.f; be[j] = (RECOMP && c < C) ? beta[c] : 0.f;
    }
    for (int n = grp + i0 * S; n < N; n += ni * S) {
        const size_t base = c4_offset(n, G, g, HW, 0);
        const float4* __restrict__ px = reinterpret_cast<const float4*>(x + base);
        const float4* __restrict__ py = RECOMP ? nullptr : reinterpret_cast<const float4*>(y + base);
        const float4* __restrict__ pd = reinterpret_cast<const float4*>(dy + base);
        for (int pix = chunk * 256 + threadIdx.x; pix < HW; pix += step) {
            const float4 xv = px[pix];
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (!RECOMP) yv = py[pix];
            const float4 dv = pd[pix];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pre = RECOMP ? bn_affine(xs[j], mu[j], is[j], ga[j], be[j]) : ys[j];
                const float d = (relu && !(pre > 0.f)) ? 0.f : ds[j];
                s[j] += d; q[j] += (double)d * ((xs[j] - mu[j]) * is[j]);
            }
        }
    }
    bn_block_partials(s, q, part + (((size_t)grp * gridDim.y + blockIdx.y) * 4 * G + 4 * g) * 2);
}

template <bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_apply_p_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const double* __restrict__ part, int Y, int C, int relu,
                                                             float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int G, int HW, int S) {
    __shared__ float sh[8];
    const int plane = blockIdx.y, n = plane / G, g = plane - n * G, grp = S == 1 ? 0 : n % S, so = grp * C, Cp = 4 * G;
    if (threadIdx.x < 64) {
        const int j = threadIdx.x & 3, c = 4 * g + j;
        double a, b;
        bn_sum_partials(part, Y, Cp, grp, g, a, b);
        const double cnt = (double)((N - grp + S - 1) / S) * HW;
        if (threadIdx.x < 4) { sh[j] = (float)(a / cnt); sh[4 + j] = (float)(b / cnt); }
        if (blockIdx.x == 0 && n == 0) {                                  // the groups' parameter gradients add in source order, in fp32, as S backward passes accumulate them
            float db = 0.f, dg = 0.f;
            for (int k = 0; k < S; ++k) {
                if (k != grp || S > 1) bn_sum_partials(part, Y, Cp, k, g, a, b);
                db += (float)a; dg += (float)b;
            }
            if (threadIdx.x < 4 && c < C) { dbeta[c] = db; dgamma[c] = dg; }
        }
    }
    __syncthreads();
    float mu[4], is[4], ga[4], be[4], sd[4], sq[4]; bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j; live[j] = c < C;
        mu[j] = live[j] ? mean[so + c] : 0.f; is[j] = live[j] ? invstd[so + c] : 0.f; ga[j] = live[j] ? gamma[c] : 0.f; be[j] = (RECOMP && live[j]) ? beta[c] : 0.f;
        sd[j] = live[j] ? sh[j] : 0.f; sq[j] = live[j] ? sh[4 + j] : 0.f;
    }
    const size_t base = (size_t)plane * HW;
    const float4* __restrict__ px = reinterpret_cast<const float4*>(x) + base;
    const float4* __restrict__ py = RECOMP ? nullptr : reinterpret_cast<const float4*>(y) + base;
    const float4* __restrict__ pd = reinterpret_cast<const float4*>(dy) + base;
    float4* __restrict__ po = reinterpret_cast<float4*>(dx) + base;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const float4 xv = px[pix];
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (!RECOMP) yv = py[pix];
        const float4 dv = pd[pix];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pre = RECOMP ? bn_affine(xs[j], mu[j], is[j], ga[j], be[j]) : ys[j];
            const float d = (relu && !(pre > 0.f)) ? 0.f : ds[j];
            const float xh = (xs[j] - mu[j]) * is[j];
            o[j] = live[j] ? ga[j] * is[j] * (d - sd[j] - xh * sq[j]) : 0.f;
        }
        po[pix] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Partial-sum workspace of the *_p entry points: 2 * 4 * ceil(C / 4) doubles per (statistics group, reducing workgroup); contents irrelevant on
// entry and exit (every slot a call reads it has written), one per stream.
extern "C" size_t cnm_bn_train_partials_doubles(int N, int C, int H, int W, int groups) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || groups < 1) return 0;
    const BnGridY gy = bn_grid_y((N + groups - 1) / groups, H * W);
    return (size_t)groups * gy.ypi * gy.ni * 8 * ((C + 3) / 4);
}
// Training-mode BatchNorm + optional ReLU on a c4 tensor, `groups` statistics groups as cnm_bn_train_forward_zg_c4_f32 (same results up to
// the summation order of the fp64 sums), in TWO launches; num_batches_tracked may be NULL.
extern "C" int cnm_bn_train_forward_p_c4_f32(const float* x, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, float momentum, float eps, int relu,
                                             float* y, float* save_mean, float* save_invstd, double* partials_ws, long long* num_batches_tracked,
                                             int N, int C, int H, int W, int groups, void* stream) {
    const int S = groups;
    CNM_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && partials_ws && N > 0 && C > 0 && H > 0 && W > 0 && S >= 1 && S <= N, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!running_mean == !running_var, CNM_ERR_BAD_ARG);
    const int G = (C + 3) / 4, HW = H * W;
    CNM_REQUIRE((long long)N * G <= 65535, CNM_ERR_BAD_ARG);
    hipStream_t s = cnm_stream(stream);
    const BnGridY gy = bn_grid_y((N + S - 1) / S, HW);
    const int Y = gy.ypi * gy.ni;
    bn_stats_p_kernel<<<dim3(G, Y, S), 256, 0, s>>>(x, N, G, HW, partials_ws, S, gy.ypi, gy.ni);
    bn_apply_p_kernel<<<bn_plane_grid(N, G, HW), 256, 0, s>>>(x, partials_ws, Y, gamma, beta, C, relu, y, N, G, HW, S, eps, momentum, save_mean, save_invstd,
                                                             running_mean, running_var, num_batches_tracked);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
// Its backward (y == NULL with relu: the ReLU mask is recomputed from x, gamma, beta and the saved statistics, as cnm_bn_train_backward_zgb_c4_f32).
extern "C" int cnm_bn_train_backward_p_c4_f32(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                                              const float* save_mean, const float* save_invstd, int relu,
                                              float* dx, float* dgamma, float* dbeta, double* partials_ws,
                                              int N, int C, int H, int W, int groups, void* stream) {
    const int S = groups;
    CNM_REQUIRE(x && (y || beta || !relu) && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && partials_ws && N > 0 && C > 0 && S >= 1 && S <= N, CNM_ERR_BAD_ARG);
    const int G = (C + 3) / 4, HW = H * W;
    CNM_REQUIRE((long long)N * G <= 65535, CNM_ERR_BAD_ARG);
    hipStream_t s = cnm_stream(stream);
    const bool recomp = !y && relu;
    const BnGridY gy = bn_grid_y((N + S - 1) / S, HW);
    const int Y = gy.ypi * gy.ni;
    const dim3 rg(G, Y, S), ag = bn_plane_grid(N, G, HW);
    if (recomp) {
        bn_bwd_reduce_p_kernel<true><<<rg, 256, 0, s>>>(x, nullptr, dy, save_mean, save_invstd, gamma, beta, C, relu, N, G, HW, partials_ws, S, gy.ypi, gy.ni);
        bn_bwd_apply_p_kernel<true><<<ag, 256, 0, s>>>(x, nullptr, dy, save_mean, save_invstd, gamma, beta, partials_ws, Y, C, relu, dx, dgamma, dbeta, N, G, HW, S);
    } else {
        bn_bwd_reduce_p_kernel<false><<<rg, 256, 0, s>>>(x, y ? y : x, dy, save_mean, save_invstd, gamma, beta, C, relu, N, G, HW, partials_ws, S, gy.ypi, gy.ni);
        bn_bwd_apply_p_kernel<false><<<ag, 256, 0, s>>>(x, y ? y : x, dy, save_mean, save_invstd, gamma, beta, partials_ws, Y, C, relu, dx, dgamma, dbeta, N, G, HW, S);
    }
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ adjoint of the bilinear x2 upsample
__global__ __launch_bounds__(256) void upsample2x_bwd_c4_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                int N, int G, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * G * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(idx % W);
        long long r = idx / W;
        const int iy = (int)(r % H); r /= H;                    // r = n*G + g
        const float4* base = reinterpret_cast<const float4*>(dy) + r * (long long)Ho * Wo;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = max(2 * iy - 2, 0); oy <= min(2 * iy + 2, Ho - 1); ++oy) {
            const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
            const float ly = sy - y0;
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = max(2 * ix - 2, 0); ox <= min(2 * ix + 2, Wo - 1); ++ox) {
                const float sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
                const int x0 = (int)sx, x1 = min(x0 + 1, W - 1);
                const float lx = sx - x0;
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                const float4 v = base[(size_t)oy * Wo + ox];
                const float w = wy * wx;
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
        }
        reinterpret_cast<float4*>(dx)[idx] = acc;
    }
}

// ------------------------------------------------------------------ masked mean L1 (IdepthLoss / IdepthwithProbLoss, losses.py:30-73)
// value = sum_m w |pred - gt| / count(m),  m = gt > 0 && finite(gt) && finite(pred) && pred > 0  (losses.py:39-40, :61).
// One launch: per-thread fp64 sums over a grid-stride walk, block partials to the workspace, the LAST block (ticket counter)
// adds the partials in block order -- the same bits whatever the block schedule -- writes (mean, count) and rearms the ticket.
constexpr int kMl1Blocks = 256;
__global__ __launch_bounds__(256) void masked_l1_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ w,
                                                        long long n, double* __restrict__ ws, float* __restrict__ out) {
    __shared__ double sh[2][4];
    __shared__ bool last;
    double s = 0.0, c = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = pred[i], g = gt[i];
        if (g > 0.f && isfinite(g) && isfinite(p) && p > 0.f) { s += (double)(fabsf(p - g) * (w ? w[i] : 1.f)); c += 1.0; }
    }
    for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o); c += __shfl_down(c, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s; sh[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 + 2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        ws[3 + 2 * blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        __threadfence();
        last = atomicAdd(reinterpret_cast<unsigned*>(ws), 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    s = 0.0; c = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) { s += __hip_atomic_load(ws + 2 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); c += __hip_atomic_load(ws + 3 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // kMl1Blocks <= 256: one partial per thread
    for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o); c += __shfl_down(c, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s; sh[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), C = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        out[0] = (float)(S / C);                                         // empty mask: 0 / 0 = NaN, the reference's mean of nothing
        out[1] = (float)C;
        *reinterpret_cast<unsigned*>(ws) = 0u;                           // zero on entry, left zero
    }
}

// d value / d pred = w sign(pred - gt) / count on the mask, 0 elsewhere;  d value / d w = |pred - gt| / count on the mask.
// `go` = gradient of the value (device scalar), `stat` = what the forward wrote (mean, count).
__global__ __launch_bounds__(256) void masked_l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ w,
                                                            const float* __restrict__ go, const float* __restrict__ stat, long long n,
                                                            float* __restrict__ dpred, float* __restrict__ dw) {
    const float k = go[0] / stat[1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = pred[i], g = gt[i];
        const bool m = g > 0.f && isfinite(g) && isfinite(p) && p > 0.f;
        const float d = p - g;
        if (dpred) dpred[i] = m ? (d > 0.f ? k : d < 0.f ? -k : 0.f) * (w ? w[i] : 1.f) : 0.f;
        if (dw) dw[i] = m ? fabsf(d) * k : 0.f;
    }
}

extern "C" size_t cnm_masked_l1_workspace_doubles(void) { return 2 + 2 * kMl1Blocks; }

extern "C" int cnm_masked_l1_f32(const float* pred, const float* gt, const float* weight, long long n, double* zero_ws, float* out2, void* stream) {
    CNM_REQUIRE(pred && gt && zero_ws && out2 && n > 0, CNM_ERR_BAD_ARG);
    const int blocks = (int)((n + 1023) / 1024 < kMl1Blocks ? (n + 1023) / 1024 : kMl1Blocks);
    masked_l1_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(pred, gt, weight, n, zero_ws, out2);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_masked_l1_backward_f32(const float* pred, const float* gt, const float* weight, const float* grad_out, const float* out2,
                                          long long n, float* dpred, float* dweight, void* stream) {
    CNM_REQUIRE(pred && gt && grad_out && out2 && n > 0 && (dpred || dweight) && (!dweight || weight), CNM_ERR_BAD_ARG);
    const int blocks = (int)((n + 1023) / 1024 < 2048 ? (n + 1023) / 1024 : 2048);
    masked_l1_bwd_kernel<<<blocks, 256, 0, cnm_stream(stream)>>>(pred, gt, weight, grad_out, out2, n, dpred, dweight);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ surface-normal loss terms (surface_normal_loss, losses.py:76-122) [r6]
// Per sample b: s[b] = sum_keep (1 - cos(pred, gt)),  c[b] = count(keep),  keep = valid && finite(sum_c gt) && finite(sum_c pred);
// cos as torch.nn.functional.cosine_similarity(dim = 1, eps = 1e-8) on vectors zeroed outside `keep`: (p . g) / (max(|p|, eps) max(|g|, eps)).
// Two launches, no atomics: block partials in fp64 to fixed slots ([b][block][2]), then one block per sample adds them in block order.
// pred, gt: [B, 3, HW] planes; valid: [B, HW] bytes (torch bool).  The torch expression of the same thing is ~40 launches per term both ways.
constexpr int kNrmBlocks = 64;
__global__ __launch_bounds__(256) void normal_cos_partials_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const unsigned char* __restrict__ valid,
                                                                  int HW, double* __restrict__ part) {
    __shared__ double sh[2][4];
    const int b = blockIdx.y;
    const float* p = pred + (size_t)b * 3 * HW; const float* g = gt + (size_t)b * 3 * HW; const unsigned char* v = valid + (size_t)b * HW;
    double s = 0.0, c = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float px = p[i], py = p[i + HW], pz = p[i + 2 * HW], gx = g[i], gy = g[i + HW], gz = g[i + 2 * HW];
        if (v[i] && isfinite(px + py + pz) && isfinite(gx + gy + gz)) {
            const float np = fmaxf(sqrtf(px * px + py * py + pz * pz), 1e-8f), ng = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-8f);
            s += (double)(1.f - (px * gx + py * gy + pz * gz) / (np * ng)); c += 1.0;
        }
    }
    for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o); c += __shfl_down(c, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wave] = s; sh[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)b * gridDim.x + blockIdx.x) * 2] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
__global__ __launch_bounds__(64) void normal_cos_finish_kernel(const double* __restrict__ part, int nblk, float* __restrict__ s_out, float* __restrict__ c_out) {
    const int b = blockIdx.x;
    double s = 0.0, c = 0.0;
    if ((int)threadIdx.x < nblk) { s = part[((size_t)b * nblk + threadIdx.x) * 2]; c = part[((size_t)b * nblk + threadIdx.x) * 2 + 1]; }   // nblk <= 64: one partial per lane
    for (int o = 32; o; o >>= 1) { s += __shfl_down(s, o); c += __shfl_down(c, o); }
    if (threadIdx.x == 0) { s_out[b] = (float)s; c_out[b] = (float)c; }
}
// d s[b] / d pred = -(g / (np ng) - (p . g) p / (np^3 ng)) on `keep` (the |p| > eps branch; below it the clamp is constant: -(g / (eps ng))), 0 elsewhere.
__global__ __launch_bounds__(256) void normal_cos_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const unsigned char* __restrict__ valid,
                                                             const float* __restrict__ go, int HW, float* __restrict__ dpred) {
    const int b = blockIdx.y;
    const float* p = pred + (size_t)b * 3 * HW; const float* g = gt + (size_t)b * 3 * HW; const unsigned char* v = valid + (size_t)b * HW;
    float* d = dpred + (size_t)b * 3 * HW;
    const float k = go[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float px = p[i], py = p[i + HW], pz = p[i + 2 * HW], gx = g[i], gy = g[i + HW], gz = g[i + 2 * HW];
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (v[i] && isfinite(px + py + pz) && isfinite(gx + gy + gz)) {
            const float n2 = px * px + py * py + pz * pz, nrm = sqrtf(n2), np = fmaxf(nrm, 1e-8f), ng = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-8f);
            const float inv = 1.f / (np * ng), dot = px * gx + py * gy + pz * gz;
            const float t = nrm > 1e-8f ? dot * inv / n2 : 0.f;           // d(1 / max(|p|, eps)) / dp = -p / |p|^3 above eps, 0 below
            dx = -k * (gx * inv - t * px); dy = -k * (gy * inv - t * py); dz = -k * (gz * inv - t * pz);
        }
        d[i] = dx; d[i + HW] = dy; d[i + 2 * HW] = dz;
    }
}
extern "C" size_t cnm_normal_cos_workspace_doubles(int B) { return B > 0 ? (size_t)B * kNrmBlocks * 2 : 0; }
extern "C" int cnm_normal_cos_terms_f32(const float* pred, const float* gt, const unsigned char* valid, int B, int HW, double* ws, float* s_out, float* c_out, void* stream) {
    CNM_REQUIRE(pred && gt && valid && ws && s_out && c_out && B > 0 && HW > 0, CNM_ERR_BAD_ARG);
    const int nblk = (HW + 1023) / 1024 < kNrmBlocks ? (HW + 1023) / 1024 : kNrmBlocks;
    normal_cos_partials_kernel<<<dim3(nblk, B), 256, 0, cnm_stream(stream)>>>(pred, gt, valid, HW, ws);
    normal_cos_finish_kernel<<<B, 64, 0, cnm_stream(stream)>>>(ws, nblk, s_out, c_out);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
extern "C" int cnm_normal_cos_terms_backward_f32(const float* pred, const float* gt, const unsigned char* valid, const float* grad_s, int B, int HW, float* dpred, void* stream) {
    CNM_REQUIRE(pred && gt && valid && grad_s && dpred && B > 0 && HW > 0, CNM_ERR_BAD_ARG);
    const int nblk = (HW + 1023) / 1024 < 256 ? (HW + 1023) / 1024 : 256;
    normal_cos_bwd_kernel<<<dim3(nblk, B), 256, 0, cnm_stream(stream)>>>(pred, gt, valid, grad_s, HW, dpred);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ backward of the disparity head (depth_layer, depthNet_model.py:82-84)
// d = scale * sigmoid(s),  s = conv3x3(x; w [1,C,3,3], zero padding) + bias  ->  ds = gd * d * (1 - d / scale),
//   dx[c](p) = sum_k ds(p - (k - 1)) w[c][k],   dw[c][k] = sum_p ds(p) x[c](p + (k - 1)),   dbias = sum_p ds(p).
// The 1-channel convolution as an MFMA problem padded to 16 output channels moved 16x the data each way; these are the two
// streaming kernels it is: every x texel is read once, every dx texel written once, ds (one float per pixel) comes from cache.
__device__ __forceinline__ float head_ds(const float* __restrict__ gd, const float* __restrict__ d, float inv_scale, long long i) {
    const float v = d[i];
    return gd[i] * v * (1.f - v * inv_scale);
}

__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ gd, const float* __restrict__ d, const float* __restrict__ w,
                                                            float inv_scale, float* __restrict__ dx, int N, int G, int H, int W) {
    const int HW = H * W;
    const unsigned total = (unsigned)N * G * HW;                          // < 2^31 (checked by the entry point)
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int pix = (int)(idx % (unsigned)HW);
        const unsigned r = idx / (unsigned)HW;
        const int g = (int)(r % (unsigned)G), n = (int)(r / (unsigned)G);
        const int y = pix / W, x = pix - y * W;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wg = w + (size_t)g * 36;                            // w[c][k], c = 4 g .. 4 g + 3
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = y - (ky - 1);                                 // x(y) feeds s(y - (ky - 1)) through kernel row ky
            if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = x - (kx - 1);
                if ((unsigned)sx >= (unsigned)W) continue;
                const float t = head_ds(gd, d, inv_scale, (long long)n * HW + sy * W + sx);
                const int k = ky * 3 + kx;
                acc.x = fmaf(t, wg[k], acc.x); acc.y = fmaf(t, wg[9 + k], acc.y); acc.z = fmaf(t, wg[18 + k], acc.z); acc.w = fmaf(t, wg[27 + k], acc.w);
            }
        }
        *reinterpret_cast<float4*>(dx + c4_offset(n, G, g, HW, pix)) = acc;
    }
}

// grid (chunks, G): block (j, g) walks its share of the N * H * W pixels, every thread keeps 36 sums (4 channels x 9 taps, +1 for the
// bias in group 0); block partials [chunk][g][37] in fp64, added in chunk order by the finishing kernel (bit-reproducible).
constexpr int kHeadChunks = 128;
__global__ __launch_bounds__(256) void head_bwd_weight_kernel(const float* __restrict__ x, int Gx_tot, int gx0, const float* __restrict__ gd,
                                                              const float* __restrict__ d, float inv_scale, double* __restrict__ partial,
                                                              int N, int G, int H, int W) {
    __shared__ double sh[4][37];
    __shared__ float red[256][37];
    const int HW = H * W, g = blockIdx.y;
    const unsigned total = (unsigned)N * HW;
    float acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.f;
    float accb = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int n = (int)(i / (unsigned)HW), pix = (int)(i - (unsigned)n * HW);
        const int y = pix / W, xx = pix - y * W;
        const float4 v = *reinterpret_cast<const float4*>(x + c4_offset(n, Gx_tot, gx0 + g, HW, pix));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = y - (ky - 1);
            if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = xx - (kx - 1);
                if ((unsigned)sx >= (unsigned)W) continue;
                const float t = head_ds(gd, d, inv_scale, (long long)n * HW + sy * W + sx);
                const int k = ky * 3 + kx;
                acc[k] = fmaf(t, v.x, acc[k]); acc[9 + k] = fmaf(t, v.y, acc[9 + k]); acc[18 + k] = fmaf(t, v.z, acc[18 + k]); acc[27 + k] = fmaf(t, v.w, acc[27 + k]);
                if (k == 4) accb += t;
            }
        }
    }
    // block sum of the 37 values through LDS (cross-lane shuffles of 37 doubles cost more than the walk itself): [thread][37] floats,
    // then thread (quarter q, slot) adds its 64 rows in fp64, thread `slot` the four quarters
#pragma unroll
    for (int i = 0; i < 36; ++i) red[threadIdx.x][i] = acc[i];
    red[threadIdx.x][36] = accb;
    __syncthreads();
    if (threadIdx.x < 148) {
        const int q = threadIdx.x / 37, slot = threadIdx.x - q * 37;
        double v = 0.0;
        for (int r = q * 64; r < q * 64 + 64; ++r) v += (double)red[r][slot];
        sh[q][slot] = v;
    }
    __syncthreads();
    if (threadIdx.x < 37) partial[((size_t)blockIdx.x * G + g) * 37 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void head_bwd_finish_kernel(const double* __restrict__ partial, int chunks, int G, int C, float* __restrict__ dw, float* __restrict__ dbias) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per (g, slot): lane l adds chunks l, l + 64, ..., then the lanes in a fixed tree
    if (i >= G * 37) return;
    const int g = i / 37, slot = i - g * 37;
    double s = 0.0;
    for (int j = lane; j < chunks; j += 64) s += partial[((size_t)j * G + g) * 37 + slot];
    for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
    if (lane != 0) return;
    if (slot < 36) { const int c = 4 * g + slot / 9; if (c < C) dw[(size_t)c * 9 + slot % 9] = (float)s; }
    else if (g == 0 && dbias) dbias[0] = (float)s;
}

extern "C" size_t cnm_head_backward_workspace_doubles(int C) { return C > 0 ? (size_t)kHeadChunks * ((C + 3) / 4) * 37 : 0; }

extern "C" int cnm_head_backward_c4_f32(const float* x, int Gx_total, int gx0, int C, const float* w_oihw, const float* grad_disp, const float* disp,
                                        float scale, float* dx, float* dw_oihw, float* dbias, double* ws, int N, int H, int W, void* stream) {
    CNM_REQUIRE(x && w_oihw && grad_disp && disp && ws && (dx || dw_oihw) && C > 0 && C % 4 == 0 && N > 0 && H > 0 && W > 0 && scale > 0.f, CNM_ERR_BAD_ARG);
    const int G = C / 4;
    CNM_REQUIRE(gx0 >= 0 && gx0 + G <= Gx_total && (long long)N * G * H * W < (1ll << 31), CNM_ERR_BAD_ARG);
    hipStream_t s = cnm_stream(stream);
    if (dx) {
        const long long total = (long long)N * G * H * W;
        head_bwd_data_kernel<<<(unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536), 256, 0, s>>>(grad_disp, disp, w_oihw, 1.f / scale, dx, N, G, H, W);
        CNM_LAUNCH_CHECK();
    }
    if (dw_oihw) {
        const long long px = (long long)N * H * W;
        const int chunks = (int)((px + 255) / 256 < kHeadChunks ? (px + 255) / 256 : kHeadChunks);
        head_bwd_weight_kernel<<<dim3(chunks, G), 256, 0, s>>>(x, Gx_total, gx0, grad_disp, disp, 1.f / scale, ws, N, G, H, W);
        CNM_LAUNCH_CHECK();
        head_bwd_finish_kernel<<<(G * 37 + 3) / 4, 256, 0, s>>>(ws, chunks, G, C, dw_oihw, dbias);
        CNM_LAUNCH_CHECK();
    }
    return CNM_OK;
}

extern "C" int cnm_upsample2x_backward_c4_f32(const float* dy, float* dx, int N, int G, int H, int W, void* stream) {
    CNM_REQUIRE(dy && dx && N > 0 && G > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    const long long total = (long long)N * G * H * W;
    upsample2x_bwd_c4_kernel<<<(int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384), 256, 0, cnm_stream(stream)>>>(dy, dx, N, G, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
