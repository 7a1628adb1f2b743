// Row-wise Winograd F(4,7) for the 7x7 stride-1 layer (depthNet conv1.0: 67 -> 128 channels on the full-resolution cost
// volume, the largest single launch of a frame), LDS-staged and persistent -- the rows counterpart of conv_winograd4s.hip.
// Same arithmetic and packed filter as conv_rows_winograd_f32_kernel<7, 1, 4> (conv_winograd_rows.hip):
//     out[y, 4 t .. 4 t + 3] = AT sum_{ky, ci} (G w[co, ci, ky, :]) (.) (BT in[ci, y + ky - 3, 4 t - 3 .. 4 t + 6])
// with the (channel group, kernel row) quads of the reduction in the same order, so results are bit-equal to it as long as
// no unit is cut by a range boundary.
//
// What changes is the data path.  The gather-fed kernel loads, per 16-deep chunk and thread, ten 16-byte window pixels
// (16-byte pieces 64 bytes apart: 64 lines per instruction) and transforms four channels per thread with packed fp32
// instructions for every (output row, kernel row) pair -- 1.8 VALU-instruction equivalents per MFMA, which come out of the
// fp32 matrix pipe's time.  Here
//   * a workgroup = 8 waves = 128 output channels x 32 tiles (4 image rows x 8 tiles of 4 pixels), one per CU, persistent;
//   * the input arrives by LDS-DMA as one 10 x 38 pixel patch per CHANNEL GROUP (coalesced rows, zero padding = out-of-range
//     offsets): a seventh of the gather traffic;
//   * BT x is computed ONCE per input row of the patch -- 10 rows x 8 tiles x 4 channels = 320 windows per channel group
//     instead of 4 x 7 x 8 x 4 = 896 (the transform of input row r serves output row y with kernel row r - y): five
//     wave-tasks per group (ten ds_read_b32, the same even / odd factorisation per element, ten ds_write_b32), dealt to
//     the eight waves in rotation -- 0.3 instructions per MFMA;
//   * the MFMA B operand of (tile row ty, kernel row ky) is the transformed row ty + ky: a per-lane LDS address, no copy;
//   * a phase = 32 reduction elements (8 quads of 4 channels x 1 kernel row); per frequency point a wave reads two 16-byte
//     weight fragments from L2 and four B fragments from LDS for 16 MFMAs (two tile blocks: 80 accumulators);
//   * phases run as one software pipeline across units (multiply phase p, transform the channel groups first needed by
//     p + 1, stage those first needed by p + 2), ranges of phases are equal per CU with the sync workspace
//     (conv_winograd4s.hip), output rows leave through LDS as full lines.
// LDS: ring of four raw planes (10 rows x 41 slots of 16 bytes; row pitch 164 dwords: the 32 lanes of a ds_read_b32 group
// -- 4 channels x 4 tiles x 2 rows -- hit 32 banks), ring of four transformed planes [10 points][10 rows][8 tiles][4]
// (12.6 KB; tile index XOR 4 on rows with (row >> 1) odd, plane pitch = 32 dwords (mod 64): the 16-lane groups of the
// B-fragment ds_read_b128 -- two kernel rows x two tile rows x four tiles -- read 64 banks or the same address), and the
// eight wave-private staging corners of the output path: 112 KB.
#include "cnm_common.h"
#include "rows_args.h"
#include "sync_ws.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define ROWS7S_MFMA(a_, b_, c_, x0, x1, x2) ((ABL & 32) ? rows7s_keep(a_, b_, c_) : __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, x0, x1, x2))
__device__ __forceinline__ f32x4 rows7s_keep(float a, float b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }

namespace {
constexpr float kBT[10][10] = {                                          // F(4,7), points 0, +-1, +-2, +-1/2, +-3/2, inf (tools/wino1d_matrices.py)
    {9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1, 0},
    {0, -9. / 4, -9. / 4, 169. / 16, 169. / 16, -13. / 2, -13. / 2, 1, 1, 0}, {0, 9. / 4, -9. / 4, -169. / 16, 169. / 16, 13. / 2, -13. / 2, -1, 1, 0},
    {0, -9. / 8, -9. / 16, 49. / 8, 49. / 16, -7, -7. / 2, 2, 1, 0}, {0, 9. / 8, -9. / 16, -49. / 8, 49. / 16, 7, -7. / 2, -2, 1, 0},
    {0, -9. / 2, -9, 61. / 8, 61. / 4, -29. / 8, -29. / 4, 1. / 2, 1, 0}, {0, 9. / 2, -9, -61. / 8, 61. / 4, 29. / 8, -29. / 4, -1. / 2, 1, 0},
    {0, -3. / 2, -1, 63. / 8, 21. / 4, -63. / 8, -21. / 4, 3. / 2, 1, 0}, {0, 3. / 2, -1, -63. / 8, 21. / 4, 63. / 8, -21. / 4, -3. / 2, 1, 0},
    {0, 9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1}};
constexpr float kAT[4][10] = {{1, 1, 1, 1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 1. / 2, -1. / 2, 3. / 2, -3. / 2, 0},
                              {0, 1, 1, 4, 4, 1. / 4, 1. / 4, 9. / 4, 9. / 4, 0}, {0, 1, -1, 8, -8, 1. / 8, -1. / 8, 27. / 8, -27. / 8, 1}};
}

#ifdef ROWS7S_DMA_NT
#define ROWS7S_DMA_AUX " nt"
#else
#define ROWS7S_DMA_AUX ""
#endif
#ifndef ROWS7S_WD
#define ROWS7S_WD 2
#endif
#ifndef ROWS7S_DMA_X1
#define ROWS7S_DMA_X1 5
#endif
// Instruction arbitration between the two waves of a SIMD: with the other wave streaming MFMAs, a wave's VALU-class instructions
// (address arithmetic, v_readlane of spilled scalars, the transform) wait for a gap in that stream -- tens of cycles each
// (tools/rows7s_timeline.py).  ROWS7S_PRIO raises the priority of a wave while it is in such a section.
#ifdef ROWS7S_PRIO
#define ROWS7S_PRIO_HI() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ROWS7S_PRIO_LO() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ROWS7S_PRIO_HI()
#define ROWS7S_PRIO_LO()
#endif
__device__ __forceinline__ void rows7s_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one LDS-DMA wave-instruction (conv_winograd4s.hip): 64 lanes x 16 bytes from buffer offset voff + soff to lds_addr + 16 lane
__device__ __forceinline__ void rows7s_dma16(unsigned lds_addr, unsigned voff, unsigned base_lo, unsigned base_hi, unsigned bytes, unsigned soff) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)base_lo), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)base_hi);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((unsigned long long)lo | ((unsigned long long)hi << 32)), 0,
                                                                          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr), so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ROWS7S_DMA_AUX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(la), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}

#ifdef ROWS7S_TIMELINE
__device__ unsigned g_rows7s_tl[2][8][12];                               // [wave 0 / wave 4][phase][point starts 0..9, phase end, after barrier]: low word of s_memtime
#endif
#ifdef ROWS7S_ABLATE
__device__ unsigned long long g_rows7s_clk[2];                           // shader cycles and 100 MHz ticks of workgroup 0's last launch
#endif
// ABL (debug builds, -DROWS7S_ABLATE, tools/rows7s_ablate.sh): bit 0 no input transform, 1 no DMA, 2 weight fragments from one hot
// KB per wave, 3 no B-fragment reads after the first, 4 no output transform / stores, 5 no MFMAs.  Results are then meaningless.
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_rows7s_f32_kernel(const RowArgs a, const int SH, const int SW, const int tilesC, const int nunits,
                                                                  unsigned* __restrict__ sync_flags, float* __restrict__ sync_slots) {
    constexpr int NX = 10, NKY = 7, TSY = 4, TSX = 8, PR = TSY + 6, PC = 4 * TSX + 6, PCP = 41;   // patch 10 x 38 pixels, rows of 41 slots
    constexpr int NPIECE = (PR * PCP + 63) / 64, PLANE = NPIECE * 1024 + 128;                        // raw plane: 7 KB + pad
    constexpr int TVX = PR * TSX * 16, TVS = NX * TVX + 128;                                         // transformed plane: point pitch 1280, plane pitch 12928 bytes
    constexpr int RAW0 = 0, TV0 = 4 * PLANE, STG0 = TV0 + 4 * TVS, LDS_BYTES = STG0 + 8 * 4352;
    constexpr int WD = ROWS7S_WD;                                                // frequency points of weight fragments in flight (two fragments each)
    constexpr int SLOT_BYTES = 8 * 16 * 64 * 16;                         // partial output of one range (as in conv_winograd4s.hip; half of it used)
    static_assert(NX % WD == 0 && TVS % 256 == 128 && TV0 % 128 == 0 && LDS_BYTES <= 120 * 1024, "layout");
#ifdef ROWS7S_TIMELINE
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + 2 * 8 * 12 * 4];
#else
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
#endif
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int HW = a.H * a.W, SHW = SH * SW, ncb16 = a.Cout / 16;
    const int ngrp = a.Gin, nquad = NKY * ngrp, nch = (nquad + 7) / 8;   // channel groups, quads, phases per unit
    [[maybe_unused]] const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

    const int G = gridDim.x, rng = xcd_remap(blockIdx.x, G);
    const long long T = (long long)nunits * nch;
    const auto range_begin = [&](int r) { return sync_flags ? (int)(T * r / G) : (int)((long long)nunits * r / G) * nch; };
    const int ps = range_begin(rng), pe = range_begin(rng + 1);
    const int P = pe - ps;
    if (P <= 0) return;
#ifdef ROWS7S_ABLATE
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int nstrips = nunits / tilesC;                                 // unit = cblk * nstrips + strip (channel block slowest)

    // The channel groups whose first quad lies in phase c of a unit ("new groups of c"): g in [(8 c + 6) / 7, min(ngrp, (8 c + 14) / 7)),
    // at most two.  They are staged two phases and transformed one phase before c; ring slots (raw and transformed) of group
    // (u, g): (u ngrp + g) & 3.
    int mu = ps / nch, mc = ps - mu * nch;                               // cursor of the multiply role: unit, phase of the unit

    // ---- stage role: one plane = NPIECE pieces of 64 slots, wave w moves piece w
    const int pslot = 64 * wave + lane, prow = pslot / PCP, pcol = pslot - prow * PCP;
    const bool pok = wave < NPIECE && prow < PR && pcol < PC;
    // Everything the staging needs from the kernel arguments lives in laundered scalars: left to itself the compiler
    // re-reads the argument block (two dependent s_load + s_waitcnt lgkmcnt(0), which also drains the LDS queue) at every use.
    unsigned in1_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in), in1_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in) >> 32), in1_bytes = a.in_bytes;
    unsigned in2_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in2), in2_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in2) >> 32), in2_bytes = a.in2_bytes;
    unsigned hw16 = (unsigned)HW * 16u;
    int gsplit = a.Gsplit;
    asm volatile("" : "+s"(in1_lo), "+s"(in1_hi), "+s"(in1_bytes), "+s"(in2_lo), "+s"(in2_hi), "+s"(in2_bytes), "+s"(hw16), "+s"(gsplit));
    unsigned svoff = 0, sbase1 = 0, sbase2 = 0; int su_cur = -1;         // this lane's offset inside a channel-group plane; byte offsets of group 0 of either view in the unit's image; the unit they belong to
    auto stage_unit = [&](int u) {
        const int strip = u % nstrips;
        const int simg = strip / SHW;
        const int rem = strip - simg * SHW, sy = rem / SW, sx = rem - sy * SW;
        const int y = TSY * sy - 3 + prow, x = 4 * TSX * sx - 3 + pcol;
        svoff = pok && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W ? (unsigned)(y * a.W + x) * 16u : 0xFFFFFFFFu;
        sbase1 = (unsigned)(simg * a.Gin_tot + a.gin0) * hw16;
        sbase2 = (unsigned)(simg * a.Gin2_tot + a.gin2_0 - a.Gsplit) * hw16;
        su_cur = u;
    };
#ifdef ROWS7S_STAGE_DMA
    auto stage_groups = [&](int u, int g0, int n) {                      // groups g0 .. g0 + n - 1 of unit u, n <= 2, by LDS-DMA
        if (n <= 0 || wave >= NPIECE || (ABL & 2)) return;
        if (u != su_cur) stage_unit(u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= n) break;
            const int g = g0 + i;
            const bool s1 = g < gsplit;
            rows7s_dma16(lds0 + RAW0 + (unsigned)(((u * ngrp + g) & 3) * PLANE + wave * 1024), svoff, s1 ? in1_lo : in2_lo, s1 ? in1_hi : in2_hi, s1 ? in1_bytes : in2_bytes,
                         (s1 ? sbase1 : sbase2) + (unsigned)g * hw16);
        }
    };
    auto stage_store = [&](int, int, int) {};
#else
    // Through registers: the loads of a phase's (at most two) pieces are issued at the start of the phase and written to LDS
    // seven frequency points later.  (An LDS-DMA instruction held its wave for 500 - 1000 cycles -- one memory latency -- before
    // the wave's next instruction issued: tools/rows7s_timeline.py, tools/wino36s_timeline.py.)
    u32x4 sreg[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    auto stage_groups = [&](int u, int g0, int n) {
        if (n <= 0 || wave >= NPIECE || (ABL & 2)) return;
        if (u != su_cur) stage_unit(u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= n) break;
            const int g = g0 + i;
            const bool s1 = g < gsplit;
            const unsigned lo = s1 ? in1_lo : in2_lo, hi = s1 ? in1_hi : in2_hi;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((unsigned long long)lo | ((unsigned long long)hi << 32)), 0,
                                                                                  s1 ? in1_bytes : in2_bytes, 0x00020000);
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, svoff, (s1 ? sbase1 : sbase2) + (unsigned)g * hw16, 0);
        }
    };
    auto stage_store = [&](int u, int g0, int n) {
        if (n <= 0 || wave >= NPIECE || (ABL & 2)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= n) break;
            *reinterpret_cast<u32x4*>(smem + RAW0 + ((u * ngrp + g0 + i) & 3) * PLANE + wave * 1024 + lane * 16) = sreg[i];
        }
    };
#endif

    // ---- transform role: a wave-task = two patch rows of one group: lane = (channel e, tile x, row): lanes 0-31 tiles 0-3, 32-63 tiles 4-7
    const int te = lane & 3, ttx = ((lane >> 2) & 3) | ((lane >> 5) << 2), trr = (lane >> 4) & 1;
    const unsigned lrd = (unsigned)((trr * PCP + 4 * ttx) * 16 + te * 4), lwr = (unsigned)(trr * 128 + ttx * 16 + te * 4);
    float xw[10], vo[10];
    auto tr_read = [&](unsigned rb) {
        if (ABL & 1) return;
#pragma unroll
        for (int j = 0; j < 10; ++j) xw[j] = *reinterpret_cast<const float*>(smem + rb + j * 16);
    };
    auto tr_compute = [&]() {                                            // conv_rows_winograd_f32_kernel<7,1,4>::transform_group, one channel: rows 0 / 9, then four +-p pairs
        if (ABL & 1) return;
#pragma unroll
        for (int grp = 0; grp < 5; ++grp) {
            const int k = grp == 0 ? 0 : 2 * grp - 1;
            float e = 0.f, o = 0.f; bool fe = true, fo = true;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float cf = kBT[k][j];
                if (cf == 0.f) continue;
                if (grp == 0 || (j & 1) == 0) { e = fe ? cf * xw[j] : fmaf(cf, xw[j], e); fe = false; }
                else { o = fo ? cf * xw[j] : fmaf(cf, xw[j], o); fo = false; }
            }
            if (grp == 0) {
                vo[0] = e;
                float v9 = 0.f; bool f9 = true;
#pragma unroll
                for (int j = 0; j < 10; ++j) { const float cf = kBT[9][j]; if (cf == 0.f) continue; v9 = f9 ? cf * xw[j] : fmaf(cf, xw[j], v9); f9 = false; }
                vo[9] = v9;
            } else { vo[k] = e + o; vo[k + 1] = e - o; }
        }
    };
    auto tr_write = [&](unsigned wb) {
        if (ABL & 1) return;
#pragma unroll
        for (int x = 0; x < NX; ++x) *reinterpret_cast<float*>(smem + wb + x * TVX) = vo[x];
    };
    // tasks of phase gp: n groups from global group number G0 on, five tasks each, task k to wave (k - 3 gp) & 7: this wave's
    // tasks are k0 = (wave + 3 gp) & 7 and k0 + 8.  Sets on[] (task present) and the LDS read / write addresses.
    auto tr_plan = [&](int gp, int G0, int n, bool* on, unsigned* rd, unsigned* wr) {
        const int k0 = (wave + 3 * gp) & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + 8 * i, gi = k >= 5 ? 1 : 0, part = k - 5 * gi;
            const int slot = (G0 + gi) & 3;
            on[i] = k < 5 * n;
            rd[i] = (unsigned)(RAW0 + slot * PLANE + 2 * part * PCP * 16) + lrd;
            wr[i] = (unsigned)(TV0 + slot * TVS + 2 * part * 128) + (lwr ^ (unsigned)((part & 1) * 64));
        }
    };
    const auto new_first = [&](int c) { return (8 * c + 6) / NKY; };
    const auto new_count = [&](int c) { return max(0, min(ngrp, (8 * c + 14) / NKY) - (8 * c + 6) / NKY); };

    // ---- multiply role
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, (unsigned)((size_t)a.nchunks * ncb16 * NX * 1024), 0x00020000);
    const unsigned lane16 = lane * 16;
    auto ldA = [&](unsigned soff) { if (ABL & 4) soff = (unsigned)(wave * 1024); const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, soff, 0); return *reinterpret_cast<const float4*>(&v); };
    auto abase = [&](int cblk, int c) {                                  // fragment (16-deep chunk 2 c, point 0) of (128-channel block, phase c) for this wave
        return (unsigned)(((2 * c) * ncb16 + cblk * 8 + wave) * NX) * 1024u;
    };
    const unsigned ahalf = (unsigned)(ncb16 * NX) * 1024u;               // the second 16-deep half of a phase
    const unsigned amax = (unsigned)((size_t)(a.nchunks - 1) * ncb16 * NX) * 1024u + (unsigned)((ncb16 - 1) * NX) * 1024u;   // odd chunk counts: the second half of the last phase does not exist (its B operand is multiplied by zeros of the last chunk row instead)
    // B fragments: lane (tile rt, quad kg) of tile block tb, half h: quad Q = 8 c + 4 h + kg = (group g, kernel row ky) ->
    // transformed row R = 2 tb + (rt >> 3) + ky of plane g, tile rt & 7 (XOR 4 when (R >> 1) is odd).  The padding quads of a
    // unit's last phase carry zero weights: they read the phase's first quad (finite data).
    const int rt = lane & 15, kg = lane >> 4;
    auto baddr = [&](int u, int c, int h) {
        int Q = 8 * c + 4 * h + kg;
        Q = Q < nquad ? Q : 8 * c;
        const int g = Q / NKY, ky = Q - g * NKY, R = (rt >> 3) + ky;
        return (unsigned)(TV0 + ((u * ngrp + g) & 3) * TVS + R * 128 + (((rt & 7) ^ (((R >> 1) & 1) * 4)) * 16));
    };
    f32x4 acc[NX][2];
#pragma unroll
    for (int x = 0; x < NX; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue = the two phases before the range's first, without multiplies: every group of the first phase is staged,
    // then transformed while the new groups of the second phase are staged
    int mcblk = mu / nstrips;
    int part_c0 = mc;
    const int pgL = (8 * mc) / NKY, pgn = min(ngrp - 1, (8 * mc + 7) / NKY) - pgL + 1;
    stage_groups(mu, pgL, pgn);
    stage_store(mu, pgL, pgn);
    unsigned a_cur = abase(mcblk, mc);
    float4 af[WD][2];
#pragma unroll
    for (int s = 0; s < WD; ++s) { af[s][0] = ldA(a_cur + s * 1024); af[s][1] = ldA(min(a_cur + ahalf, amax) + s * 1024); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ton[2]; unsigned trd[2], twr[2];
    {
        const bool l1 = mc + 1 == nch;
        const int u1 = l1 ? mu + 1 : mu, c1 = l1 ? 0 : mc + 1;
        const int n1 = ps + 1 < pe ? new_count(c1) : 0;
        stage_groups(u1, new_first(c1), n1);
        stage_store(u1, new_first(c1), n1);
        tr_plan(ps - 1, mu * ngrp + pgL, pgn, ton, trd, twr);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (ton[i]) { tr_read(trd[i]); tr_compute(); tr_write(twr[i]); }
        tr_plan(ps, u1 * ngrp + new_first(c1), n1, ton, trd, twr);       // the tasks of the first phase
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rows7s_lds_barrier();
    unsigned bq[2] = {baddr(mu, mc, 0), baddr(mu, mc, 1)};

    for (int p = 0; p < P; ++p) {
        const int gp = ps + p;
        const bool lastc = mc + 1 == nch;
        const int nu = lastc ? mu + 1 : mu, nc = lastc ? 0 : mc + 1;     // the next phase
        const int ncblk = lastc && nu % nstrips == 0 ? mcblk + 1 : mcblk;
        const unsigned a_nxt = p + 1 < P ? abase(ncblk, nc) : a_cur;
        const bool last2 = nc + 1 == nch;
        const int u2 = last2 ? nu + 1 : nu, c2 = last2 ? 0 : nc + 1;     // the phase after: its new groups are staged now and transformed in the next phase
        const int n2 = gp + 2 < pe ? new_count(c2) : 0, g2 = new_first(c2);
        bool ton_n[2]; unsigned trd_n[2], twr_n[2];
        const unsigned b00 = bq[0], b01 = bq[1], b10 = (bq[0] ^ 64u) + 256u, b11 = (bq[1] ^ 64u) + 256u;   // [tile block][half]: two rows down flips the tile XOR
        float4 bf[2][2];
        bf[0][0] = *reinterpret_cast<const float4*>(smem + b00); bf[0][1] = *reinterpret_cast<const float4*>(smem + b01);
        bf[1][0] = *reinterpret_cast<const float4*>(smem + b10); bf[1][1] = *reinterpret_cast<const float4*>(smem + b11);
        __builtin_amdgcn_sched_barrier(0);
#ifdef ROWS7S_TIMELINE
        unsigned tl[12];
#endif
#pragma unroll
        for (int x = 0; x < NX; ++x) {                                   // one frequency point per step: 16 MFMAs
#ifdef ROWS7S_TIMELINE
            tl[x] = (unsigned)__builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
#endif
            const float4 a0 = af[x % WD][0], a1 = af[x % WD][1];
            float4 b[2][2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) { b[tb][0] = bf[tb][0]; b[tb][1] = bf[tb][1]; }
            if (x + 1 < NX && !(ABL & 8)) {
                bf[0][0] = *reinterpret_cast<const float4*>(smem + b00 + (x + 1) * TVX); bf[0][1] = *reinterpret_cast<const float4*>(smem + b01 + (x + 1) * TVX);
                bf[1][0] = *reinterpret_cast<const float4*>(smem + b10 + (x + 1) * TVX); bf[1][1] = *reinterpret_cast<const float4*>(smem + b11 + (x + 1) * TVX);
            }
            acc[x][0] = ROWS7S_MFMA(a0.x, b[0][0].x, acc[x][0], 0, 0, 0);
            acc[x][1] = ROWS7S_MFMA(a0.x, b[1][0].x, acc[x][1], 0, 0, 0);
            {
                const unsigned base = x + WD < NX ? a_cur + (x + WD) * 1024 : a_nxt + (x + WD - NX) * 1024;
                af[x % WD][0] = ldA(base); af[x % WD][1] = ldA(min(base + ahalf, amax + (x + WD < NX ? (x + WD) : (x + WD - NX)) * 1024u));
            }
            // between the MFMAs: staging (the two waves of a SIMD at different points: a wave's next weight-fragment wait also
            // waits for its DMA -- vmcnt counts in order), this wave's transform tasks, the next phase's B addresses
            ROWS7S_PRIO_HI();
#ifdef ROWS7S_STAGE_DMA
            if (x == 0 && wave < 4) stage_groups(u2, g2, n2);
            if (x == ROWS7S_DMA_X1 && wave >= 4) stage_groups(u2, g2, n2);
#else
            if (x == 0) stage_groups(u2, g2, n2);
            if (x == 7) stage_store(u2, g2, n2);
#endif
            if (x == 4) tr_plan(gp + 1, u2 * ngrp + g2, n2, ton_n, trd_n, twr_n);
            if (x == 1 && ton[0]) tr_read(trd[0]);
            if (x == 3 && ton[0]) tr_write(twr[0]);
            if (x == 6 && ton[1]) tr_read(trd[1]);
            if (x == 8 && ton[1]) tr_write(twr[1]);
            if (x == 9) { bq[0] = baddr(nu, nc, 0); bq[1] = baddr(nu, nc, 1); }
            ROWS7S_PRIO_LO();
            auto rest = [&]() __attribute__((always_inline)) {           // the other 14 MFMAs of the point
                acc[x][0] = ROWS7S_MFMA(a0.y, b[0][0].y, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a0.y, b[1][0].y, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a0.z, b[0][0].z, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a0.z, b[1][0].z, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a0.w, b[0][0].w, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a0.w, b[1][0].w, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a1.x, b[0][1].x, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a1.x, b[1][1].x, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a1.y, b[0][1].y, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a1.y, b[1][1].y, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a1.z, b[0][1].z, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a1.z, b[1][1].z, acc[x][1], 0, 0, 0);
                acc[x][0] = ROWS7S_MFMA(a1.w, b[0][1].w, acc[x][0], 0, 0, 0);
                acc[x][1] = ROWS7S_MFMA(a1.w, b[1][1].w, acc[x][1], 0, 0, 0);
            };
            if ((x == 2 || x == 7) && ton[x == 2 ? 0 : 1]) {
                // the task's arithmetic shares the basic block with the MFMAs: four VALU instructions behind each MFMA (a lump
                // of fifty between two MFMAs leaves the matrix pipe idle unless the SIMD's other wave happens to feed it)
                ROWS7S_PRIO_HI();
                tr_compute();
                ROWS7S_PRIO_LO();
                rest();
#ifndef ROWS7S_PRIO
#pragma unroll
                for (int k = 0; k < 14; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
#endif
            } else rest();
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef ROWS7S_TIMELINE
        tl[10] = (unsigned)__builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * WD) : "memory");   // every DMA of the phase is older than the 2 WD fragments still in flight
#ifdef ROWS7S_NOBARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // timing experiment: wrong results
#else
        rows7s_lds_barrier();
#endif
        ROWS7S_PRIO_HI();
#ifdef ROWS7S_TIMELINE
        tl[11] = (unsigned)__builtin_readcyclecounter();
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && p >= P / 2 && p < P / 2 + 8) {   // parked in LDS, copied out at the end
#pragma unroll
            for (int i = 0; i < 12; ++i) reinterpret_cast<unsigned*>(smem + LDS_BYTES)[((wave >> 2) * 8 + p - P / 2) * 12 + i] = tl[i];
        }
#endif
        a_cur = a_nxt;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ton[i] = ton_n[i]; trd[i] = trd_n[i]; twr[i] = twr_n[i]; }
        if (!lastc && p + 1 < P) { ++mc; continue; }

        // ---- a part of unit mu ends (phases part_c0 .. mc): whole unit -> finish; head part -> add the following ranges'
        // partial outputs in range order, finish; any other part -> publish the partial output (conv_winograd4s.hip)
        const bool publish = part_c0 != 0;
        int nsrc = 0;
        if (!publish && !lastc) {
            for (int rem = nch - 1 - mc; rem > 0; ++nsrc) rem -= range_begin(rng + nsrc + 2) - range_begin(rng + nsrc + 1);
            if (t == 0) {                                                // generation-valued flags, loud time-out: sync_ws.h
                const unsigned gen = sync_generation();
                for (int k = 1; k <= nsrc; ++k) sync_wait(sync_flags, rng + k, gen);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        if (ABL & 16) {
#pragma unroll
            for (int x = 0; x < NX; ++x) { asm volatile("" :: "v"(acc[x][0]), "v"(acc[x][1])); acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        } else {
            int le = lane; asm volatile("" : "+v"(le));
            const int ert = le & 15, ekg = le >> 4;
            const int strip = mu % nstrips;
            const int img = strip / SHW, rem = strip - img * SHW, sy = rem / SW, sx = rem - sy * SW;
            const int co = mcblk * 128 + wave * 16 + 4 * ekg;
            const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            const unsigned slot_lane = (unsigned)(wave * 16 * 64 + le);
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sync_slots, 0, sync_slots ? (unsigned)G * (unsigned)SLOT_BYTES : 0u, 0x00020000);
            // staging corner (conv_winograd4s.hip): [channel group][pixel of the tile][tile], pixel pitch 272 bytes; lane = pixel on the way out
            const unsigned stg = (unsigned)(STG0 + wave * 4352);
            const unsigned stw = stg + ekg * 1088 + ert * 16, str_ = stg + (le & 3) * 272 + (le >> 2) * 16;
            const int cow = mcblk * 128 + wave * 16;
            float* obase = a.out + c4_offset(img, a.Gout_tot, a.gout0 + (cow >> 2), HW, 0);
            const size_t gstride = (size_t)HW * 4;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {                             // tile block tb = image rows 2 tb, 2 tb + 1 of the unit, 8 tiles each
                f32x4 y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                        // y_i = sum_k AT[i][k] M_k on top of the bias, in the gather-fed kernel's order
                        float sacc = publish ? 0.f : bb[r];
                        if ((ABL & 128) && !publish && nsrc) { y[i][r] = sacc; continue; }   // debug: a head part contributes the bias only
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            const float cf = kAT[i][k];
                            if (cf == 1.f) sacc += acc[k][tb][r];
                            else if (cf == -1.f) sacc -= acc[k][tb][r];
                            else if (cf != 0.f) sacc = fmaf(cf, acc[k][tb][r], sacc);
                        }
                        y[i][r] = sacc;
                    }
                if (publish) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&y[i]), srsrc, (slot_lane + (unsigned)(4 * tb + i) * 64u) * 16u, (unsigned)rng * (unsigned)SLOT_BYTES, 16);
                    // the 16-byte-per-lane stores read their data registers over several cycles and hipcc knows no hazard for the
                    // SGPR-soffset form: with the registers rewritten by the very next instruction the last quarter of each
                    // 16 lanes of the last store went out with the NEW values on MI355X (tools/rows7s_debug.py)
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                for (int k = 1; k <= nsrc; ++k) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (slot_lane + (unsigned)(4 * tb + i) * 64u) * 16u, (unsigned)(rng + k) * (unsigned)SLOT_BYTES, 16);
                        if (!(ABL & 64)) y[i] += *reinterpret_cast<const f32x4*>(&pv);
                    }
                }
                if (tb) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the first block's staging reads are done before the corner is rewritten
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = y[i];
                    if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    *reinterpret_cast<f32x4*>(smem + stw + i * 272) = v;
                }
                // reading lane = pixel: tile qt = le >> 2 of the block (row 2 tb + (qt >> 3), tile column qt & 7), pixel le & 3
                const int qt = le >> 2;
                const int qrow = TSY * sy + 2 * tb + (qt >> 3), qcol = 4 * (TSX * sx + (qt & 7)) + (le & 3);
                const bool stv = qrow < a.H && qcol < a.W;
                float* orow = obase + (size_t)(qrow * a.W + qcol) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + str_ + g * 1088);
                    if (stv) *reinterpret_cast<f32x4*>(orow + g * gstride) = v;
                }
            }
            if (publish) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sync_publish(sync_flags, rng, sync_generation()); }
            }
#pragma unroll
            for (int x = 0; x < NX; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        if (lastc) { ++mu; mcblk = ncblk; }
        mc = lastc ? 0 : mc + 1; part_c0 = 0;
    }
#ifdef ROWS7S_TIMELINE
    __syncthreads();
    if (blockIdx.x == 0 && t < 2 * 8 * 12) (&g_rows7s_tl[0][0][0])[t] = reinterpret_cast<unsigned*>(smem + LDS_BYTES)[t];
#endif
#ifdef ROWS7S_ABLATE
    if (blockIdx.x == 0 && t == 0) { g_rows7s_clk[0] = __builtin_readcyclecounter() - clk0; g_rows7s_clk[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

#ifdef ROWS7S_ABLATE
static int g_rows7s_abl = 0;
extern "C" int cnm_tune_rows7s_ablate(int m) { const int old = g_rows7s_abl; g_rows7s_abl = m; return old; }
extern "C" double cnm_debug_rows7s_mhz() {                                 // average shader clock of workgroup 0 over the last launch
    unsigned long long c[2] = {0, 0};
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_rows7s_clk), sizeof(c)) != hipSuccess || !c[1]) return 0.0;
    return (double)c[0] / (double)c[1] * 100.0;
}
#endif
#ifdef ROWS7S_TIMELINE
extern "C" int cnm_debug_rows7s_timeline(unsigned* out) {                  // 2 x 8 x 12 words of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows7s_tl), sizeof(unsigned) * 2 * 8 * 12) == hipSuccess ? 0 : -1;
}
#endif
static int g_rows7_staged = 1;                                           // tuning knob: 0 off, 1 on where eligible
extern "C" int cnm_tune_rows7_staged(int on) { const int old = g_rows7_staged; if (on == 0 || on == 1) g_rows7_staged = on; return old; }

static int rows7s_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) { int n = 0; cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    return cus[dev];
}

// CNM_OK after launching, 1 when the shape is not eligible (the caller launches the gather-fed kernel), negative on failure.
int cnm_rows7s_try_launch(const RowArgs& a, float* sync_ws, size_t sync_floats, hipStream_t stream) {
    if (!g_rows7_staged || a.Cout % 128 || a.W < 32 || a.H < 4 || a.Ho != a.H || a.Wo != a.W) return 1;
    const int SH = cnm_ceil_div(a.H, 4), SW = cnm_ceil_div(a.W, 32), tilesC = a.Cout / 128;
    const long long nunits = (long long)a.N * SH * SW * tilesC;
    const int nch = (7 * a.Gin + 7) / 8;
    if (nunits <= 0 || nunits * nch > 0x7FFFFFFF || (long long)a.nchunks * (a.Cout / 16) * 10 * 1024 >= 0xFFFFFFFFll) return 1;
    const int cus = rows7s_cus();
    int grid = (int)(nunits < cus ? nunits : cus);
    unsigned* flags = nullptr; float* slots = nullptr;
    sync_ctl_upload(stream);
    if (cnm_sync_failed()) return CNM_ERR_LAUNCH;                        // an earlier hand-off timed out: refuse until cnm_engine_status(1)
    if (sync_ws && cus <= kSyncMaxRanges && sync_floats * 4 >= kSyncFlagBytes + (size_t)cus * kSyncSlotBytes) {
        const long long T = nunits * nch;
        grid = (int)(T / 4 < cus ? (T / 4 > 0 ? T / 4 : 1) : cus);
        flags = reinterpret_cast<unsigned*>(sync_ws); slots = sync_ws + kSyncFlagBytes / 4;
    }
#ifdef ROWS7S_ABLATE
#define R7S_CASE(m) case m: conv_rows7s_f32_kernel<m><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots); break;
    switch (g_rows7s_abl) { R7S_CASE(1) R7S_CASE(2) R7S_CASE(3) R7S_CASE(4) R7S_CASE(8) R7S_CASE(16) R7S_CASE(32) R7S_CASE(7) R7S_CASE(15) R7S_CASE(31) R7S_CASE(47) R7S_CASE(64) R7S_CASE(128)
    default: conv_rows7s_f32_kernel<0><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots); }
#else
    conv_rows7s_f32_kernel<0><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
#endif
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
