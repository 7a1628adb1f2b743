// Winograd F(4x4,3x3), LDS-staged and persistent: the successor of conv_winograd36_f32_kernel<4,3> (conv_winograd4.hip)
// for the layers whose output-channel count is a multiple of 128.  Same arithmetic, same packed filters, bit-identical
// results; what changes is how the operands reach the matrix cores.
//
// Why (profiles/r2_conv_pmc.txt): the gather-fed kernel needs its texture addresser as much as its matrix pipes -- per
// 16-channel chunk a wave issues 36 one-dword window gathers (16 lines per instruction) and 36 weight fragments for 144
// MFMAs -- and every VALU instruction of the input transform (180 per 144 MFMAs) takes its cycles out of the fp32 matrix
// pipe, which shares the SIMD's fp32 lanes (profiles/r1_mfma_valu_probes.txt).  Here
//   * a workgroup = 8 waves = 128 output channels x 16 tiles: the transformed input of a chunk is shared by twice the
//     MFMAs, and each thread transforms HALF a window (three of the six frequency rows): 72 instead of 180 VALU
//     instructions per 144 MFMAs;
//   * the raw 6 x 66 (2 x 8 tiles: 10 x 34) input patch of the tile block reaches LDS by LDS-DMA (buffer_load_dwordx4 ...
//     lds: 16-byte pixels, coalesced rows, overlapping window columns fetched once, zero padding = out-of-range offsets)
//     -- 28 wave-instructions per chunk for the workgroup instead of 288 dword gathers;
//   * the grid is persistent (one workgroup per CU, 128 KB of LDS): a workgroup walks units (tile block, 128-channel
//     block) and runs ONE software pipeline across them -- phase p multiplies chunk p, transforms chunk p + 1 and
//     stages chunk p + 2, whichever unit they belong to -- so only the output transform of a unit is not overlapped.
//
// LDS: V[2][36 points][16 tiles][16 ci] (72 KB, the MFMA B operand, slots XOR-swizzled as in the gather-fed kernel) and
// RAW[2][4 channel groups][plane] (56 KB; plane = patch rows x patch columns x 16 bytes, plane pitch = 16 (mod 128) bytes
// so that the 32 lanes of a ds_read_b32 group -- 16 channels x 2 neighbouring tiles -- fall on 32 different banks).
// Roles of wave w (lane l):
//   multiply   output channels 16 w .. 16 w + 15 of the unit's block, all 16 tiles, all 36 points (144 accumulators);
//   transform  tiles 4 (w & 3) + (l >> 4), channel l & 15, frequency rows {0,1,2} (w < 4) or {5,3,4} (w >= 4): the
//              column pass B^T d produces three of six rows with the SAME instruction stream for both halves (one
//              shifted base address and two constants differ), the row pass is the full 6-point transform on them;
//   stage      DMA pieces w, w + 8, ... of the chunk's 28; waves 0-3 issue theirs at the start of a phase, waves 4-7
//              a third of a phase later: vmcnt retires in order, so a weight fragment issued behind a first-touch DMA
//              waits out its HBM latency, and the two waves of a SIMD must not do that at the same time.
// The DMA is inline assembly on purpose: beside a DMA it knows about, hipcc orders every loop-carried VGPR load behind
// s_waitcnt vmcnt(0) (checked in the ISA); hidden from its counters, its counted waits for the weight fragments stay
// counted and can only over-wait.  DMA completion is guaranteed by a counted wait before the phase barrier.
#include "wino4_args.h"
#include "sync_ws.h"
#include <mutex>

#ifndef WINO4S_WD
#define WINO4S_WD 4       // weight fragments in flight per wave
#endif
#ifndef WINO4S_CBLK_SLOW
#define WINO4S_CBLK_SLOW 1 // unit order: 1 = channel block slowest (an XCD's neighbouring ranges stream ONE block's filters: a fragment is fetched from
#endif                     // beyond L2 once per ~32 workgroups), 0 = channel block fastest (the workgroups of a tile block share its input instead)
#ifndef WINO4S_EARLY_BARRIER
#define WINO4S_EARLY_BARRIER 0  // 1: the phase barrier before the last double step + prefetch of the next phase's first B fragments behind it: measured 1.5 % slower (two more live registers spill)
#endif
#ifndef WINO4S_LO_SHARE
#define WINO4S_LO_SHARE 7           // sevenths of a phase's DMA pieces issued by waves 0-3: all of them (see NLO / NHI below)
#endif
#ifndef WINO4S_LO_STEP
#define WINO4S_LO_STEP 14          // double step at which waves 0-3 issue their DMA pieces: late, so that the holds fall into their wait at the phase barrier
#endif
#ifndef WINO4S_PRIO
#define WINO4S_PRIO 0
#endif
#ifndef WINO4S_CLUMP
#define WINO4S_CLUMP 0
#endif
#ifndef WINO4S_CLUMP_A
#define WINO4S_CLUMP_A 4
#define WINO4S_CLUMP_B 10
#endif
#ifndef WINO4S_HI_STEP
#define WINO4S_HI_STEP 6  // double step at which waves 4-7 start issuing their DMA pieces (waves 0-3: step 0); >= 4
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define WINO4S_BT(x0, x1, x2, x3, x4, x5) do {                                                                  \
        const float t0 = fmaf(4.f, x0, fmaf(-5.f, x2, x4)), t5 = fmaf(4.f, x1, fmaf(-5.f, x3, x5));             \
        const float e1 = fmaf(-4.f, x2, x4), o1 = fmaf(-4.f, x1, x3);                                           \
        const float e2 = x4 - x2, o2 = x3 - x1;                                                                 \
        x0 = t0; x1 = e1 + o1; x2 = e1 - o1; x3 = fmaf(2.f, o2, e2); x4 = fmaf(-2.f, o2, e2); x5 = t5;          \
    } while (0)
// A^T of F(4,3) on four channels at once: rows [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
#define WINO4S_AT(y0, y1, y2, y3, m0, m1, m2, m3, m4, m5) do {                                                  \
        const f32x4 s1 = (m1) + (m2), d1 = (m1) - (m2), s2 = (m3) + (m4), d2 = (m3) - (m4);                     \
        y0 = (m0) + s1 + s2; y1 = d1 + 2.f * d2; y2 = s1 + 4.f * s2; y3 = d1 + 8.f * d2 + (m5);                 \
    } while (0)

// A^T of F(3,4) (3 outputs from the same 6 points): rows [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]
#define WINO3S_AT(y0, y1, y2, m0, m1, m2, m3, m4, m5) do {                                                      \
        const f32x4 s1 = (m1) + (m2), d1 = (m1) - (m2), s2 = (m3) + (m4), d2 = (m3) - (m4);                     \
        y0 = (m0) + s1 + s2; y1 = d1 + 2.f * d2; y2 = s1 + 4.f * s2 + (m5);                                     \
    } while (0)

// A^T of F(2,5) (2 outputs from the same 6 points): rows [1 1 1 1 1 0; 0 1 -1 2 -2 1]
#define WINO2S_AT(y0, y1, m0, m1, m2, m3, m4, m5) do {                                                          \
        const f32x4 s1 = (m1) + (m2), d1 = (m1) - (m2), s2 = (m3) + (m4), d2 = (m3) - (m4);                     \
        y0 = (m0) + s1 + s2; y1 = d1 + 2.f * d2 + (m5);                                                         \
    } while (0)

__device__ __forceinline__ void wino4s_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from buffer offset voff + soff to lds_addr + 16 lane (an
// out-of-range voff lands as zeros).  m0 is saved and restored: the compiler does not model it across the statement.
// The descriptor and the two scalar operands are pinned to SGPRs with readfirstlane (under register pressure the compiler
// may hold wave-uniform values in VGPRs, which the "s" constraints cannot take); the leading s_nop covers the
// readfirstlane -> buffer-instruction hazard, which hipcc does not pad inside an asm statement.
__device__ __forceinline__ void wino4s_dma16(unsigned lds_addr, unsigned voff, unsigned base_lo, unsigned base_hi, unsigned bytes, unsigned soff) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)base_lo), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)base_hi);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((unsigned long long)lo | ((unsigned long long)hi << 32)), 0,
                                                                          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr), so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(la), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}

// ABL (debug builds, -DWINO4S_ABLATE, tools/wino36s_ablate.sh): bit 0 no input transform, 1 no DMA, 2 no weight loads in the
// loop, 3 no B-fragment reads in the loop, 5 weights from one L1-hot block, 6 half the weight loads -- timing experiments,
// results are wrong.
// Register budget (256 per lane, two waves per SIMD): 144 accumulators + 16 (four weight fragments in flight) + 16 (B
// fragments of this and the next double step) leave ~80.  Hence: the transform runs as two passes (frequency row T, then
// rows P / M: at most 16 values live), the per-unit DMA offsets live in LDS, everything the output transform needs is
// derived inside it from an opaque copy of the lane id (loop-invariant code motion would otherwise park ~20 registers
// across the phase loop), and the two transform constants are wave-uniform (SGPR operands).
// M = 4: F(4x4,3x3); M = 2: F(2x2,5x5) -- the same six interpolation points, 6 x 6 window and 36 frequency points; only
// the tile pitch (M), the filter transform (in the packed filter) and the output transform (2 x 6 instead of 4 x 6) differ.
// S2: a STRIDE-2 convolution as a stride-1 convolution of the four pixel phases of its input (space to depth, never
// materialised): a 5x5 stride-2 filter is four 3x3 filters (M = 4), a 7x7 one four 4x4 filters (M = 3: F(3x3,4x4), again the
// same six points), one per phase image P[py][px](v, u) = in(2 v + py, 2 u + px).  a.H / a.W are the OUTPUT (= phase image)
// dimensions, chunk c = 4 * (16-channel chunk of the input) + 2 py + px; the four phases of a patch are staged back to back,
// so the half of every 32-byte sector a phase leaves behind is still in the cache when the next phase asks for it.  Only the
// DMA addressing knows: lane offsets step two pixels / two rows, the phase is a scalar offset.
#ifdef WINO4S_TIMELINE
__device__ unsigned g_wino4s_tl[2 * 8 * 24];                             // [wave 0 / wave 4][phase][double-step starts 0..17, phase end, after barrier]: low word of s_memtime
#endif
template <int TSX, bool UPS, int ABL = 0, int M = 4, bool S2 = false>   // tile block = (16 / TSX) x TSX tiles of M x M outputs
__global__ __launch_bounds__(512, 2) void conv_winograd36s_f32_kernel(const Wino4Args a, const int SH, const int SW, const int tilesC, const int nunits,
                                                                       unsigned* __restrict__ sync_flags, float* __restrict__ sync_slots) {
    static_assert((M == 4 || M == 2 || (M == 3 && (S2 || UPS))) && !(UPS && (M == 2 || S2)), "F(4x4,3x3) (optionally on a 2x upsampled input or on the phases of a stride-2 5x5), F(2x2,5x5), F(3x3,4x4) on the phases of a stride-2 7x7 or as its phase-scattering data gradient");
    constexpr int TSY = 16 / TSX, PR = M * TSY + 6 - M, PC = M * TSX + 6 - M, NSLOT = PR * PC;
    constexpr int PADW = M == 3 ? (S2 ? 2 : 1) : (6 - M) / 2;            // the window starts PADW pixels before the tile (4-tap phase filters of a 7x7: taps -2 .. 1; of its data gradient: -1 .. 2)
    constexpr int NPX = M == 3 ? 3 : 4, NST = M == 3 ? 3 : M * M / 4;    // output transform: pixels per step (M = 3: one tile row), steps
    // Channel-group planes of the raw patch: NPIECE KB each, at pitch PLANE plus a pad per plane chosen so that the 32 lanes
    // of a ds_read_b32 group (16 channels x 2 neighbouring tiles, M pixels = 4 M dwords apart) fall on 32 different banks:
    // M = 4 (tiles 16 dwords apart): planes 4 dwords apart (mod 32); M = 2 (tiles 8 apart): planes at 0, 4, 16, 20; M = 3 (tiles 12 apart): 0, 8, 16, 24.
    constexpr int NPIECE = (NSLOT + 63) / 64, PLANE = NPIECE * 1024 + 128, RAWBUF = 4 * PLANE;   // bytes
#define WINO4S_PLANE_OFF(q) ((q) * PLANE + (M == 4 ? (q) * 16 : M == 3 ? (q) * 32 : ((q) & 1) * 16 + ((q) >> 1) * 64))
    constexpr int VBUF = 36 * 16 * 64, RAW0 = 2 * VBUF;                  // bytes
    // DMA pieces per chunk, and per wave: waves 0-3 take NLO each, waves 4-7 NHI.  The two waves of a SIMD do not share the matrix pipe
    // evenly -- the older one (0-3) runs ahead and then waits ~2500 cycles at the phase barrier for the other -- and a DMA instruction
    // holds its wave for ~600 cycles (tools/wino36s_timeline.py).  So the waiting waves issue ALL the pieces, late in the phase
    // (WINO4S_LO_STEP), where the holds eat into that wait and not into the trailing wave's critical path: 5.68 -> 5.60 ms per
    // step's launches (tools/wino36s_variants.sh; the same split issued at step 0: no gain).
    constexpr int NDMA = 4 * NPIECE, NLO = (NDMA * WINO4S_LO_SHARE / 7 + 3) / 4, NHI = (NDMA - 4 * NLO + 3) / 4 > 0 ? (NDMA - 4 * NLO + 3) / 4 : 0, DPW = NLO > NHI ? NLO : NHI;
    constexpr int DV0 = RAW0 + 2 * RAWBUF;                               // [DPW][512] per-thread DMA offsets of the unit being staged
    constexpr int WD = WINO4S_WD, NXI = 36;
    constexpr int SLOT_BYTES = 8 * 16 * 64 * 16;                         // one range's partial output: 8 waves x 16 pixels x 64 lanes x float4 = 128 KB
    static_assert(PLANE % 128 == 0 && NXI % WD == 0 && WD % 2 == 0, "layout");
#ifdef WINO4S_TIMELINE
    constexpr int TL0 = DV0 + DPW * 512 * 4;
    __shared__ __attribute__((aligned(16))) char smem[TL0 + 2 * 8 * 24 * 4];
#else
    __shared__ __attribute__((aligned(16))) char smem[DV0 + DPW * 512 * 4];   // 136 KB (TSX 16) / 126 KB (TSX 8)
#endif
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = wave >> 2;
    const int HW = a.H * a.W, SHW = SH * SW, ncb16 = a.Cout / 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

    // Work = the flat list of phases (unit, chunk), unit = tile block x 128-channel block (channel block fastest).  Range r
    // of the G equal contiguous ranges belongs to this workgroup (ranges of one XCD are neighbours).  With a sync
    // workspace the ranges cut units wherever the arithmetic says (every CU gets the same number of phases, whatever the
    // unit count); without one they are rounded to unit boundaries.
    const int G = gridDim.x, rng = xcd_remap(blockIdx.x, G), nch = a.nchunks;
    const long long T = (long long)nunits * nch;
    const auto range_begin = [&](int r) { return sync_flags ? (int)(T * r / G) : (int)((long long)nunits * r / G) * nch; };
    const int ps = range_begin(rng), pe = range_begin(rng + 1);
    const int P = pe - ps;                                               // phases of this workgroup
    if (P <= 0) return;

    // ---- stage role: pieces n = wave + 8 m of the chunk's NDMA (plane q = n / NPIECE, piece k = n % NPIECE)
    // What a DMA instruction needs from the kernel arguments lives in laundered scalars: left to itself the compiler re-reads the
    // argument block at every piece -- two dependent s_load + s_waitcnt lgkmcnt(0) (which also drains the LDS queue): with
    // four pieces that was 2500 - 4000 cycles of a 9200-cycle phase (tools/wino36s_timeline.py)
    unsigned in1_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in), in1_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in) >> 32), in1_bytes = a.in_bytes;
    unsigned in2_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in2), in2_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in2) >> 32), in2_bytes = a.in2_bytes;
    unsigned hw16 = (unsigned)HW * (S2 ? 64u : 16u);                      // bytes between channel groups of the input image (S2: 2H x 2W pixels)
    int gsplit = a.Gsplit, gin = a.Gin;
    unsigned row32 = (unsigned)a.W * 32u;                                // S2: bytes of one input row
    asm volatile("" : "+s"(in1_lo), "+s"(in1_hi), "+s"(in1_bytes), "+s"(in2_lo), "+s"(in2_hi), "+s"(in2_bytes), "+s"(hw16), "+s"(gsplit), "+s"(gin));
    if constexpr (S2) asm volatile("" : "+s"(row32));
    int du = ps / nch, dc = ps - du * nch, dgp = ps;                     // cursor of the stage role: unit, chunk, global phase
    unsigned dbase1 = 0, dbase2 = 0;                                     // byte offsets of channel group 0 of either view in that unit's image
    auto dma_n = [&](int m) { return hi ? 4 * NLO + (wave - 4) + 4 * m : wave + 4 * m; };   // piece m of this wave
    auto dma_unit = [&]() {                                              // per-lane offsets for unit du -> LDS (each thread re-reads only its own words)
        const int nstrips = nunits / tilesC;
        const int strip = WINO4S_CBLK_SLOW ? du % nstrips : du / tilesC;
        const int dimg = strip / SHW;
        dbase1 = (unsigned)(dimg * a.Gin_tot + a.gin0) * hw16;
        dbase2 = (unsigned)(dimg * a.Gin2_tot + a.gin2_0 - a.Gsplit) * hw16;
        const int rem = strip - dimg * SHW, sy = rem / SW, sx = rem - sy * SW;
        const int y0 = M * TSY * sy - PADW, x0 = M * TSX * sx - PADW;     // the window starts PADW pixels before the tile
#pragma unroll
        for (int m = 0; m < DPW; ++m) {
            const int n = dma_n(m), k = n % NPIECE;
            const int slot = 64 * k + lane, r = slot / PC, pc = slot - r * PC;
            int y = y0 + r, x = x0 + pc;
            bool ok = slot < NSLOT;
            if (UPS && !a.ups_zero) { y = min(max(y, 0), a.H - 1); x = min(max(x, 0), a.W - 1); }
            else ok = ok & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            *reinterpret_cast<unsigned*>(smem + DV0 + (m * 512 + t) * 4) = ok ? (unsigned)(S2 ? 4 * y * a.W + 2 * x : y * a.W + x) * 16u : 0xFFFFFFFFu;
        }
    };
    auto dma_piece = [&](int m) {                                        // piece m of phase (du, dc) into the buffer of that phase
        int wv = wave; asm volatile("" : "+s"(wv));                      // opaque here: what follows is recomputed on the scalar unit every phase (free) instead of hoisted, spilled and read back with v_readlane (VALU cycles = matrix-pipe cycles)
        const int n = hi ? 4 * NLO + (wv - 4) + 4 * m : wv + 4 * m;
        if ((ABL & 2) || m >= (hi ? NHI : NLO) || n >= NDMA || dgp >= pe) return;
        const int q = n / NPIECE, g = (S2 ? dc >> 2 : dc) * 4 + q;
        const unsigned phoff = S2 ? ((dc >> 1) & 1) * row32 + (dc & 1) * 16u : 0u;   // S2: the chunk's pixel phase
        const bool s1 = g < gsplit;
        const unsigned bytes = g < gin ? (s1 ? in1_bytes : in2_bytes) : 0u;
        const unsigned voff = *reinterpret_cast<const unsigned*>(smem + DV0 + (m * 512 + t) * 4);
        wino4s_dma16(lds0 + RAW0 + (unsigned)(((dgp - ps) & 1) * RAWBUF + WINO4S_PLANE_OFF(q) + (n - q * NPIECE) * 1024), voff, s1 ? in1_lo : in2_lo, s1 ? in1_hi : in2_hi, bytes,
                     (s1 ? dbase1 : dbase2) + (unsigned)g * hw16 + phoff);
    };
    auto dma_advance = [&]() { ++dgp; if (++dc == nch) { dc = 0; ++du; if (dgp < pe) dma_unit(); } };

    // ---- transform role
    const int ttile = 4 * (wave & 3) + (lane >> 4), tci = lane & 15, tcg = tci >> 2;
    const unsigned rbase = RAW0 + WINO4S_PLANE_OFF(tcg) + (((ttile / TSX) * M) * PC + (ttile % TSX) * M) * 16 + (tci & 3) * 4;   // + hi * PC * 16 for the shifted taps
    const unsigned wbase = ttile * 64 + ((tcg ^ ((ttile >> 1) & 3)) * 4 + (tci & 3)) * 4;   // row T at + hi * 5 * 6144, row P at + 6144 + hi * 2 * 6144, row M 6144 further
    const float beta = __int_as_float(__builtin_amdgcn_readfirstlane(hi ? 0xbf800000 : 0xc0800000));    // -1 : -4
    const float gamma = __int_as_float(__builtin_amdgcn_readfirstlane(hi ? 0x40000000 : 0x3f800000));   //  2 :  1
    auto ldsf = [&](unsigned off) { return *reinterpret_cast<const float*>(smem + off); };
    float tq[6][3], pq[6][4], rowT[6], rowP[6], rowM[6];
    // rows {0,1,2} (waves 0-3) / {5,3,4} (waves 4-7) of B^T d, column j: the gather-fed kernel's column_pass, half of it.
    // Row T = 4 d[0+s] - 5 d[2+s] + d[4+s] (s = 0 / 1: the shifted base rT), rows P, M = E +- gamma O with
    // E = beta d2 + d4, O = beta d1 + d3 (beta, gamma = -4, 1 / -1, 2).
    auto trT_read = [&](int j, unsigned rT) { if (ABL & 1) return; tq[j][0] = ldsf(rT + (0 * PC + j) * 16); tq[j][1] = ldsf(rT + (2 * PC + j) * 16); tq[j][2] = ldsf(rT + (4 * PC + j) * 16); };
    auto trT_col = [&](int j) { if (ABL & 1) return; rowT[j] = fmaf(4.f, tq[j][0], fmaf(-5.f, tq[j][1], tq[j][2])); };
    auto trP_read = [&](int j, unsigned r0) { if (ABL & 1) return; pq[j][0] = ldsf(r0 + (1 * PC + j) * 16); pq[j][1] = ldsf(r0 + (2 * PC + j) * 16); pq[j][2] = ldsf(r0 + (3 * PC + j) * 16); pq[j][3] = ldsf(r0 + (4 * PC + j) * 16); };
    auto trP_col = [&](int j) {
        if (ABL & 1) return;
        const float E = fmaf(beta, pq[j][1], pq[j][3]), O = fmaf(beta, pq[j][0], pq[j][2]);
        rowP[j] = fmaf(gamma, O, E); rowM[j] = fmaf(-gamma, O, E);
    };
    auto tr_row = [&](float* x, unsigned wb) {                           // full row pass + store of the six points of one frequency row
        if (ABL & 1) return;
        WINO4S_BT(x[0], x[1], x[2], x[3], x[4], x[5]);
#pragma unroll
        for (int l = 0; l < 6; ++l) *reinterpret_cast<float*>(smem + wb + l * 1024) = x[l];
    };

    // ---- multiply role
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, (unsigned)((size_t)a.nchunks * ncb16 * NXI * 1024), 0x00020000);
    const unsigned lane16 = lane * 16;
    const unsigned bvoff = ((lane & 15) * 16 + ((lane >> 4) ^ (((lane & 15) >> 1) & 3)) * 4) * 4;
    auto ldA = [&](unsigned soff) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, soff, 0); return *reinterpret_cast<const float4*>(&v); };
    auto abase = [&](int cblk, int c) {                                  // byte offset of fragment 0 of (128-channel block, chunk c) for this wave
        return (unsigned)((c * ncb16 + cblk * 8 + wave) * NXI) * 1024u;
    };
    f32x4 acc[NXI];
#pragma unroll
    for (int x = 0; x < NXI; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stage phases 0 and 1, first weight fragments, transform phase 0
    dma_unit();
#pragma unroll
    for (int m = 0; m < DPW; ++m) dma_piece(m);
    dma_advance();
#pragma unroll
    for (int m = 0; m < DPW; ++m) dma_piece(m);
    dma_advance();
    int mu = ps / nch, mc = ps - mu * nch;                               // cursor of the multiply role: unit, chunk
    const int nstrips_m = nunits / tilesC;
    int mcblk = WINO4S_CBLK_SLOW ? mu / nstrips_m : mu % tilesC;          // channel block of unit mu
    int part_c0 = mc;                                                    // first chunk of the part of unit mu this workgroup multiplies
    unsigned a_cur = abase(mcblk, mc);
    float4 af[WD];
#pragma unroll
    for (int s = 0; s < WD; ++s) af[s] = ldA(a_cur + s * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wino4s_lds_barrier();
    {
        const unsigned r0 = rbase, rT = rbase + hi * PC * 16, w0 = wbase;
#pragma unroll
        for (int j = 0; j < 6; ++j) { trT_read(j, rT); trT_col(j); trP_read(j, r0); trP_col(j); }
        tr_row(rowT, w0 + hi * 5 * 6144); tr_row(rowP, w0 + 6144 + hi * 2 * 6144); tr_row(rowM, w0 + 2 * 6144 + hi * 2 * 6144);
    }
    wino4s_lds_barrier();

    // WINO4S_EARLY_BARRIER = 1 (experiment, off): the phase barrier at the start of the LAST double step (every LDS write of the
    // phase and every read of V(p) has been issued by then) and the first B fragments of phase p + 1 read behind it, under the
    // step's MFMAs.  The last phase of a part keeps the barrier behind the loop (the output transform wants the registers).
    float4 bf0, bf1;
    bool bf_ready = false;
#if WINO4S_PRIO                                                          // static priority for the second-dispatched half (guide: "Two waves per SIMD", item 4); 2 = the first half instead
    if ((WINO4S_PRIO == 1) == (hi != 0)) __builtin_amdgcn_s_setprio(1);
#endif
    for (int p = 0; p < P; ++p) {
        const unsigned vc = (p & 1) * VBUF + bvoff;
        const unsigned r0 = rbase + ((p + 1) & 1) * RAWBUF, rT = r0 + hi * PC * 16;      // raw buffer of phase p + 1
        const unsigned w0 = wbase + ((p + 1) & 1) * VBUF;                                 // V buffer of phase p + 1
        const bool lastc = mc + 1 == nch;
        int ncblk = mcblk;                                               // channel block of unit mu + 1 (worked out at unit boundaries only: a division)
        if (lastc) ncblk = WINO4S_CBLK_SLOW ? (mu + 1) / nstrips_m : (mcblk + 1 == tilesC ? 0 : mcblk + 1);
        const unsigned a_nxt = p + 1 < P ? (lastc ? abase(ncblk, 0) : a_cur + (unsigned)(ncb16 * NXI) * 1024u) : a_cur;
        const bool early = WINO4S_EARLY_BARRIER && !lastc && p + 1 < P;   // wave-uniform
        if (!bf_ready) {
            bf0 = *reinterpret_cast<const float4*>(smem + vc);
            bf1 = *reinterpret_cast<const float4*>(smem + vc + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef WINO4S_TIMELINE
        const bool tl_on = blockIdx.x == 0 && (wave == 0 || wave == 4) && p >= P / 2 && p < P / 2 + 8;
        unsigned* const tl_row = reinterpret_cast<unsigned*>(smem + TL0) + ((wave >> 2) * 8 + (tl_on ? p - P / 2 : 0)) * 24;
#define WINO4S_TL(i) do { const unsigned tv_ = (unsigned)__builtin_readcyclecounter(); if (tl_on && lane == 0) tl_row[i] = tv_; } while (0)
#endif
#pragma unroll
        for (int xp = 0; xp < NXI / 2; ++xp) {
#ifdef WINO4S_TIMELINE
            WINO4S_TL(xp);
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int x0 = 2 * xp, x1 = x0 + 1;
            const float4 a0 = af[x0 % WD], a1 = af[x1 % WD];
            const float4 b0 = bf0, b1 = bf1;
            if (xp + 1 < NXI / 2 && !(ABL & 8)) {
                bf0 = *reinterpret_cast<const float4*>(smem + vc + (x0 + 2) * 1024);
                bf1 = *reinterpret_cast<const float4*>(smem + vc + (x1 + 2) * 1024);
            }
            if (xp + 1 == NXI / 2 && early) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WD) : "memory");   // the phase's DMA data (older than the WD fragments in flight)
                wino4s_lds_barrier();
                const unsigned vn = ((p + 1) & 1) * VBUF + bvoff;
                bf0 = *reinterpret_cast<const float4*>(smem + vn);
                bf1 = *reinterpret_cast<const float4*>(smem + vn + 1024);
            }
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[x1], 0, 0, 0);
            if (ABL & 32) {                                              // every wave reads the same 36 KB: fragments from L1 / L2-hot lines
                af[x0 % WD] = ldA(((x0 + WD) % NXI) * 1024);
                af[x1 % WD] = ldA(((x1 + WD) % NXI) * 1024);
            } else if (ABL & 64) {                                       // half the fragment loads
                af[x0 % WD] = ldA(x0 + WD < NXI ? a_cur + (x0 + WD) * 1024 : a_nxt + (x0 + WD - NXI) * 1024);
            } else if (!(ABL & 4)) {
                af[x0 % WD] = ldA(x0 + WD < NXI ? a_cur + (x0 + WD) * 1024 : a_nxt + (x0 + WD - NXI) * 1024);
                af[x1 % WD] = ldA(x1 + WD < NXI ? a_cur + (x1 + WD) * 1024 : a_nxt + (x1 + WD - NXI) * 1024);
            }
            // between the MFMAs: the transform of phase p + 1 -- row T (double steps 0-4), rows P and M (3-11) -- and the
            // DMA of phase p + 2 (waves 0-3: double steps 0.., waves 4-7: double steps 6..)
            if (!WINO4S_CLUMP && xp >= 1 && xp <= 3) { trT_col(2 * xp - 2); trT_col(2 * xp - 1); }
            if (xp <= 2) { trT_read(2 * xp, rT); trT_read(2 * xp + 1, rT); }
            if (!WINO4S_CLUMP && xp == 4) tr_row(rowT, w0 + hi * 5 * 6144);
            if (!WINO4S_CLUMP && xp >= 4 && xp <= 9) trP_col(xp - 4);
            if (xp >= 3 && xp <= 8) trP_read(xp - 3, r0);
            if (!WINO4S_CLUMP && xp == 10) tr_row(rowP, w0 + 6144 + hi * 2 * 6144);
            if (!WINO4S_CLUMP && xp == 11) tr_row(rowM, w0 + 2 * 6144 + hi * 2 * 6144);
#if defined(WINO4S_DMA_SPREAD)                                            // one piece every WINO4S_DMA_SPREAD double steps (waves 4-7: half a stride later)
            {
                constexpr int SP = WINO4S_DMA_SPREAD;
                const int xs = xp - (hi ? SP / 2 : 0);
#pragma unroll
                for (int m = 0; m < DPW; ++m)
                    if (xs == SP * m) { dma_piece(m); if (m == DPW - 1) dma_advance(); }
            }
#else
            // all pieces of a wave in ONE step: vmcnt retires in order, so the weight fragments loaded after a DMA instruction wait
            // for its data -- once per phase instead of once per piece (tools/wino36s_timeline.py)
            // (with every piece on waves 0-3 -- NHI == 0, the shipped split -- waves 4-7 carry no stage role at all: they are the trailing
            // waves of their SIMDs, i.e. the phase's critical path, and the cursor bookkeeping alone was 25 spill reloads per phase)
            if ((xp == WINO4S_LO_STEP && !hi) || (NHI > 0 && xp == WINO4S_HI_STEP && hi)) {
#ifdef WINO4S_TIMELINE
                __builtin_amdgcn_sched_barrier(0); WINO4S_TL(20); __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int m = 0; m < DPW; ++m) dma_piece(m);
#ifdef WINO4S_TIMELINE
                __builtin_amdgcn_sched_barrier(0); WINO4S_TL(21); __builtin_amdgcn_sched_barrier(0);
#endif
                dma_advance();
#ifdef WINO4S_TIMELINE
                __builtin_amdgcn_sched_barrier(0); WINO4S_TL(22); __builtin_amdgcn_sched_barrier(0);
#endif
            }
#endif
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[x1], 0, 0, 0);
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc[x1], 0, 0, 0);
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc[x1], 0, 0, 0);
            // WINO4S_CLUMP [r6]: the transform's VALU work in TWO clumps behind the MFMAs of a double step (24 + 48 operations) instead of groups of
            // 2-12 between them: an fp32 VALU instruction between two fp32 MFMAs drains and refills the matrix pipe (conv_winograd4q.hip)
            if (WINO4S_CLUMP && xp == WINO4S_CLUMP_A) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) trT_col(j);
                tr_row(rowT, w0 + hi * 5 * 6144);
            }
            if (WINO4S_CLUMP && xp == WINO4S_CLUMP_B) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) trP_col(j);
                tr_row(rowP, w0 + 6144 + hi * 2 * 6144);
                tr_row(rowM, w0 + 2 * 6144 + hi * 2 * 6144);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // every DMA this wave issued in the phase is older than the WD weight fragments still in flight: vmcnt retires in order
#ifdef WINO4S_TIMELINE
        WINO4S_TL(18);
#endif
        if (!early) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WD) : "memory");
            wino4s_lds_barrier();                                        // V / RAW of phase p + 1 complete; the buffers of phase p are free
        }
        bf_ready = early;
#ifdef WINO4S_TIMELINE
        WINO4S_TL(19);
#endif
        a_cur = a_nxt;
        if (!lastc && p + 1 < P) { ++mc; continue; }

        // ---- a part of unit mu ends here (chunks part_c0 .. mc).  Whole unit: finish it.  Head part (chunk 0 .. mc <
        // last): the following ranges hold the rest -- add their published partial outputs in range order, then finish.
        // Any other part: publish the partial output (no bias / ReLU) in this range's slot.
        const bool publish = part_c0 != 0;
        int nsrc = 0;                                                    // partial outputs to add: ranges rng + 1 .. rng + nsrc
        if (!publish && !lastc) {
            for (int rem = nch - 1 - mc; rem > 0; ++nsrc) rem -= range_begin(rng + nsrc + 2) - range_begin(rng + nsrc + 1);
            if (t == 0) {                                                // one lane polls (relaxed) for this launch's generation and re-arms what it saw, one acquire for the workgroup; a time-out is reported to the host (sync_ws.h)
                const unsigned gen = sync_generation();
                for (int k = 1; k <= nsrc; ++k) sync_wait(sync_flags, rng + k, gen);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        // ---- output transform of the part: acc row = cout 4 kg + r (one c4 group), col = tile rtile
        if (ABL & 16) {                                                  // timing experiment: no output transform / stores (the accumulators stay live)
#pragma unroll
            for (int x = 0; x < NXI; ++x) { asm volatile("" :: "v"(acc[x])); acc[x] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        } else {
            int le = lane; asm volatile("" : "+v"(le));                  // opaque: nothing below is loop-invariant to the compiler
            const int rtile = le & 15, kg = le >> 4;
            const int strip = WINO4S_CBLK_SLOW ? mu % nstrips_m : mu / tilesC;
            const int img = strip / SHW, rem = strip - img * SHW, sy = rem / SW, sx = rem - sy * SW;
            const int oty = sy * TSY + rtile / TSX, otx = sx * TSX + rtile % TSX;
            const int co = mcblk * 128 + wave * 16 + 4 * kg;
            const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const f32x4 bb = {bv.x, bv.y, bv.z, bv.w};
            f32x4 s[M][6];                                               // A^T M: M rows x 6 columns
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if constexpr (M == 4) WINO4S_AT(s[0][j], s[1][j], s[2][j], s[3][j], acc[0 * 6 + j], acc[1 * 6 + j], acc[2 * 6 + j], acc[3 * 6 + j], acc[4 * 6 + j], acc[5 * 6 + j]);
                else if constexpr (M == 3) WINO3S_AT(s[0][j], s[1][j], s[2][j], acc[0 * 6 + j], acc[1 * 6 + j], acc[2 * 6 + j], acc[3 * 6 + j], acc[4 * 6 + j], acc[5 * 6 + j]);
                else WINO2S_AT(s[0][j], s[1][j], acc[0 * 6 + j], acc[1 * 6 + j], acc[2 * 6 + j], acc[3 * 6 + j], acc[4 * 6 + j], acc[5 * 6 + j]);
            }
            // partial outputs travel as [range][wave][pixel][lane] float4: every lane re-reads exactly what its twin wrote
            const unsigned slot_lane = (unsigned)(wave * 16 * 64 + le);
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sync_slots, 0, sync_slots ? (unsigned)G * (unsigned)SLOT_BYTES : 0u, 0x00020000);
            // Finished rows leave through LDS: a lane owns 4 channels of an M x M tile, i.e. 16-byte pieces 16 M bytes apart --
            // stored directly, a unit is thousands of partial-line writes, and with every workgroup of the chip reaching its
            // output transform at the same time that burst costs 10-30 % of a layer (tools/wino36s_ablate.sh).  Four pixels per
            // lane and step (M = 4: row `st` of the tile; M = 2: the whole tile) are transposed in a wave-private 4.25 KB
            // corner of the V buffer the multiply role has just released ([channel group][pixel slot][tile], slot pitch 272
            // bytes: conflict-free writes, two-way reads) so that lane = pixel: one store instruction = 64 (M = 2: 2 x 32)
            // consecutive pixels of one channel group = 1 KB (two 512-byte rows for 2 x 8 tile blocks).
            const unsigned stg = (unsigned)((p & 1) * VBUF + wave * 4352);
            const unsigned stw = stg + kg * 1088 + rtile * 16;
            // reading lane = pixel: tile qt, slot qs (M = 4: column le & 3 of row st; M = 2: row le >> 5, column le & 1)
            // (M = 3: tile le / 3, column le % 3 of row st -- 48 lanes, 768 bytes per store)
            const int qt = M == 4 ? le >> 2 : M == 3 ? min(le / 3, 15) : (le & 31) >> 1, qs = M == 4 ? le & 3 : M == 3 ? le % 3 : ((le >> 5) << 1) | (le & 1);
            const unsigned str_ = stg + qs * 272 + qt * 16;
            const int qty = sy * TSY + qt / TSX, qtx = sx * TSX + qt % TSX;
            const int qcol = M * qtx + (M == 4 ? le & 3 : M == 3 ? qs : le & 1);
            const bool qok = qty < a.TH && qtx < a.TW && (M != 3 || le < 48);
            const int cow = mcblk * 128 + wave * 16;                                                // the wave's 16 output channels
            const int Cr = a.Cout >> 2, ph = UPS ? cow / Cr : 0, pa = ph >> 1, pb = ph & 1;        // UPS: virtual channels -> (phase, real channels)
            const int Ho = 2 * a.H, Wo = 2 * a.W;
            float* obase = UPS ? a.out + c4_offset(img, a.Gout_tot, a.gout0 + ((cow - ph * Cr) >> 2), 4 * HW, 0)
                               : a.out + c4_offset(img, a.Gout_tot, a.gout0 + (cow >> 2), HW, 0);
            const size_t gstride = (size_t)(UPS ? 4 * HW : HW) * 4;                                 // floats between channel groups
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                f32x4 y[4];                                              // the step's four pixels: (row, column) = M = 4: (st, sl); M = 2: (sl >> 1, sl & 1); M = 3: three pixels (st, sl)
                if constexpr (M == 4) WINO4S_AT(y[0], y[1], y[2], y[3], s[st][0], s[st][1], s[st][2], s[st][3], s[st][4], s[st][5]);
                else if constexpr (M == 3) WINO3S_AT(y[0], y[1], y[2], s[st][0], s[st][1], s[st][2], s[st][3], s[st][4], s[st][5]);
                else { WINO2S_AT(y[0], y[1], s[0][0], s[0][1], s[0][2], s[0][3], s[0][4], s[0][5]); WINO2S_AT(y[2], y[3], s[1][0], s[1][1], s[1][2], s[1][3], s[1][4], s[1][5]); }
                if (publish) {                                           // write-through (sc1) stores: no release fence needed before the flag
#pragma unroll
                    for (int x = 0; x < NPX; ++x)
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&y[x]), srsrc, (slot_lane + (unsigned)(4 * st + x) * 64u) * 16u, (unsigned)rng * (unsigned)SLOT_BYTES, 16);
                    // the 16-byte-per-lane stores read their data registers over several cycles and hipcc knows no hazard for the
                    // SGPR-soffset form: with the registers rewritten by the very next instruction the last quarter of each
                    // 16 lanes of the last store went out with the NEW values on MI355X (tools/rows7s_debug.py)
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                for (int k = 1; k <= nsrc; ++k) {                        // fixed order: own part, then the following ranges
                    // sc1 loads, matching the sc1 stores: with plain loads behind the acquire a few 64-byte pieces per slot
                    // came back stale on MI355X (tools/wino36s_vis.sh)
#pragma unroll
                    for (int x = 0; x < NPX; ++x) {
                        const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (slot_lane + (unsigned)(4 * st + x) * 64u) * 16u, (unsigned)(rng + k) * (unsigned)SLOT_BYTES, 16);
                        y[x] += *reinterpret_cast<const f32x4*>(&pv);
                    }
                }
#pragma unroll
                for (int x = 0; x < NPX; ++x) {                          // bias, ReLU (own channels), then into the staging row
                    f32x4 v = y[x];
                    bool fin = true;
                    if constexpr (UPS) {
                        const int oy = 2 * (M * oty + st) + pa, ox = 2 * (M * otx + x) + pb;
                        fin = !(a.ring && ((oy == 0) | (oy == Ho - 1) | (ox == 0) | (ox == Wo - 1)));   // ring pixels are finished by the ring kernel
                    }
                    if (fin) {
                        v += bb;
                        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    }
                    *reinterpret_cast<f32x4*>(smem + stw + x * 272) = v;
                }
                const int qrow = M * qty + (M == 2 ? le >> 5 : st);
                const bool stv = qok && qrow < a.H && qcol < a.W;
                float* orow = UPS ? obase + (size_t)((2 * qrow + pa) * Wo + 2 * qcol + pb) * 4 : obase + (size_t)(qrow * a.W + qcol) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {                            // LDS operations of one wave complete in order: no wait between the writes above and these reads
                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + str_ + g * 1088);
                    if (stv) *reinterpret_cast<f32x4*>(orow + g * gstride) = v;
                }
            }
            if (publish) {                                               // every storing wave drains its stores, then ONE lane raises the flag
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) sync_publish(sync_flags, rng, sync_generation());
            }
#pragma unroll
            for (int x = 0; x < NXI; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            wino4s_lds_barrier();                                        // every wave is done with its staging corner before the next phase's transform writes that V buffer
        }
        if (lastc) { ++mu; mcblk = ncblk; }
        mc = 0; part_c0 = 0;
    }
#ifdef WINO4S_TIMELINE
    __syncthreads();
    if (blockIdx.x == 0 && t < 2 * 8 * 24) g_wino4s_tl[t] = reinterpret_cast<unsigned*>(smem + TL0)[t];
#endif
}

#ifdef WINO4S_TIMELINE
extern "C" int cnm_debug_wino4s_timeline(unsigned* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino4s_tl), sizeof(unsigned) * 2 * 8 * 24) == hipSuccess ? 0 : -1; }
#endif
#ifdef WINO4S_ONE_INSTANCE   // codegen probes (tools/hotloop_proxy.sh): only the dominant instance, no host code
template __global__ void conv_winograd36s_f32_kernel<16, false, 0, 4, false>(const Wino4Args, const int, const int, const int, const int, unsigned* __restrict__, float* __restrict__);
#else
static int g_wino36_staged = 1;                                          // tuning knob (A/B against the gather-fed kernel): 0 off, 1 where it pays, 2 wherever eligible
extern "C" int cnm_tune_wino36_staged(int on) { const int old = g_wino36_staged; if (on >= 0 && on <= 2) g_wino36_staged = on; return old; }

#ifdef WINO4S_ABLATE
static int g_wino36s_ablate = 0;
extern "C" int cnm_tune_wino36s_ablate(int m) { const int old = g_wino36s_ablate; if (m >= 0) g_wino36s_ablate = m; return old; }
#endif

static int wino4s_cus() {                                                // compute units of the current device, queried once per device
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) { int n = 0; cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    return cus[dev];
}

// Sync workspace of one launch (sync_ws.h): kSyncFlagBytes of flag / generation words followed by one partial-output slot per range.
extern "C" size_t cnm_wino36_sync_floats(void) { return (kSyncFlagBytes + (size_t)wino4s_cus() * kSyncSlotBytes) / 4; }

// The pinned status words a timed-out poll writes (sync_ws.h): ONE PER DEVICE [r5] -- a time-out on one device makes the staged
// entry points refuse on THAT device only (the header promises per-call state for DataParallel-style callers with one thread
// per device).  A device's word is allocated on its first staged launch that is not being captured into a graph (allocation is
// not allowed during capture; launches recorded before it exists run without host reporting: cnm_engine_status() allocates it
// too, so a caller that checks the status once before capturing has it).
static unsigned* g_sync_status[64] = {nullptr};                          // per device; host address == device address (mapped, portable)
static unsigned g_sync_spin_limit = 0, g_sync_version[64] = {0}, g_sync_knob_version = 0;   // 0 = kSyncDefaultSpins; the versions count changes
static std::mutex g_sync_mutex;
static int sync_device() { int dev = 0; return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? dev : -1; }
static unsigned* sync_status_word(int dev, hipStream_t stream, bool may_allocate) {
    if (dev < 0) return nullptr;
    if (!g_sync_status[dev] && may_allocate) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            std::lock_guard<std::mutex> lock(g_sync_mutex);
            if (!g_sync_status[dev]) {
                void* p = nullptr;
                if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
                    *reinterpret_cast<volatile unsigned*>(p) = 0u;
                    g_sync_status[dev] = reinterpret_cast<unsigned*>(p);
                    ++g_sync_version[dev];
                } else (void)hipGetLastError();
            }
        }
    }
    return g_sync_status[dev];
}
SyncCtl cnm_sync_ctl(hipStream_t stream) {
    const int dev = sync_device();
    unsigned* const w = sync_status_word(dev, stream, true);
    return SyncCtl{w, g_sync_spin_limit, (dev >= 0 ? g_sync_version[dev] : 0u) + 64u * g_sync_knob_version};
}
bool cnm_sync_failed() { const int dev = sync_device(); return dev >= 0 && g_sync_status[dev] && *reinterpret_cast<volatile unsigned*>(g_sync_status[dev]) != 0u; }
// Test hook (not in the header): the generation the kernels of one launch see -- tests/test_gpu_parity.py checks that it differs
// from launch to launch, eager and in HIP-graph replays.
__global__ void sync_generation_probe_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = sync_generation(); }
extern "C" int cnm_debug_sync_generation(unsigned* out, int nblocks, void* stream) {
    CNM_REQUIRE(out && nblocks > 0, CNM_ERR_BAD_ARG);
    sync_generation_probe_kernel<<<nblocks, 64, 0, cnm_stream(stream)>>>(out);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// CNM_OK, or CNM_ERR_LAUNCH when a stream-K hand-off timed out since the last clear (the outputs of that launch are
// wrong).  Reads a pinned host word: synchronise the stream first if the launch in question may still be running.
extern "C" int cnm_engine_status(int clear) {
    const int dev = sync_device();
    (void)sync_status_word(dev, nullptr, true);                          // the current device's word exists from the first status query on
    const bool failed = cnm_sync_failed();
    if (failed && clear) *reinterpret_cast<volatile unsigned*>(g_sync_status[dev]) = 0u;
    return failed ? CNM_ERR_LAUNCH : CNM_OK;
}
// Debug / test only: polls before a hand-off gives up (0 = the default, 2^24); bit 31 = fault injection (publishers keep their flags down).
extern "C" unsigned cnm_tune_sync_spin_limit(unsigned v) { const unsigned old = g_sync_spin_limit; if (v != old) { g_sync_spin_limit = v; ++g_sync_knob_version; } return old ? old : kSyncDefaultSpins; }

int cnm_wino36s_try_launch(const Wino4Args& a, int M, int ups, hipStream_t stream, int s2) {
    const bool only_here = s2 || M == 3;                                 // forms without a gather-fed twin: the A/B knob and the balance heuristic below do not apply
    if ((!g_wino36_staged && !only_here) || (M != 4 && M != 2 && !(M == 3 && (s2 || ups))) || (ups && (M == 2 || s2)) || (s2 && M == 2) || a.Cout % 128) return 1;
    int tsx = 0;
    if (a.TW >= 12) tsx = 16; else if (a.TW >= 6 && a.TH >= 2) tsx = 8; else if (M == 4 && !ups && !s2 && a.TW >= 3 && a.TH >= 3) tsx = 4;   // tile block 1 x 16, 2 x 8, 4 x 4
    if (s2) {
        // stride-2 form: the tile block shape that pads the tile grid least (ties: the widest).  conv1.3 (32 x 43 tiles of 3 x 3
        // outputs): 4 x 4 blocks cover 32 x 44 tiles, 1 x 16 blocks 32 x 48 -- 0.99 -> 0.90 ms (tools/s2_probe.py)
        long long best = -1;
        for (int c = 16; c >= 4; c >>= 1) {
            const int cy = 16 / c;
            if (!(c == 16 ? a.TW >= 12 : c == 8 ? (a.TW >= 6 && a.TH >= 2) : (a.TW >= 3 && a.TH >= 3))) continue;
            const long long cover = (long long)cnm_ceil_div(a.TW, c) * c * cnm_ceil_div(a.TH, cy) * cy;
            if (best < 0 || cover < best) { best = cover; tsx = c; }
        }
    }
    if (!tsx) return 1;
    const int tsy = 16 / tsx;
    const int SH = cnm_ceil_div(a.TH, tsy), SW = cnm_ceil_div(a.TW, tsx), tilesC = a.Cout / 128;
    const long long nunits = (long long)a.N * SH * SW * tilesC;
    if (nunits <= 0 || nunits * a.nchunks > 0x7FFFFFFF || (long long)a.nchunks * (a.Cout / 16) * 36 * 1024 >= 0xFFFFFFFFll) return 1;
    const int cus = wino4s_cus();
    int grid = (int)(nunits < cus ? nunits : cus);
    unsigned* flags = nullptr; float* slots = nullptr;
    sync_ctl_upload(stream);
    if (cnm_sync_failed()) return CNM_ERR_LAUNCH;                        // an earlier hand-off timed out: refuse until cnm_engine_status(1) has acknowledged it
    if (a.sync_ws && cus <= kSyncMaxRanges && a.sync_floats * 4 >= kSyncFlagBytes + (size_t)cus * kSyncSlotBytes) {
        // phase ranges may cut units: every CU gets the same number of phases (at least four, or the prologue and the
        // fix-up of a range cost more than they balance)
        const long long T = nunits * a.nchunks;
        grid = (int)(T / 4 < cus ? (T / 4 > 0 ? T / 4 : 1) : cus);
        flags = reinterpret_cast<unsigned*>(a.sync_ws);
        slots = a.sync_ws + kSyncFlagBytes / 4;
    } else if (g_wino36_staged != 2 && !only_here && nunits > cus && (double)((nunits + cus - 1) / cus) * cus / (double)nunits > 1.15) {
        // Without a sync workspace ranges end on unit boundaries: one workgroup per CU walks ceil(units / CUs) units, and a
        // mostly empty last round costs more than the kernel gains (1.09-1.13x on whole rounds, 0.85-0.89x at 1.5 rounds,
        // tools/wino36s_probe.py); the gather-fed kernel, two independent workgroups per CU, balances those better.
        return 1;
    }
#ifdef WINO4S_ABLATE
    if (g_wino36s_ablate && tsx == 16 && ups && M == 4 && !s2) {          // [r6] the fused up_conv instance: where its 0.57 of the roof goes
        switch (g_wino36s_ablate) {
#define WINO4S_UCASE(n) case n: conv_winograd36s_f32_kernel<16, true, n><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots); break;
            WINO4S_UCASE(1) WINO4S_UCASE(2) WINO4S_UCASE(4) WINO4S_UCASE(16) WINO4S_UCASE(7)
            default: return CNM_ERR_BAD_ARG;
        }
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
    if (g_wino36s_ablate && tsx == 16 && !ups && M == 4 && !s2) {
        switch (g_wino36s_ablate) {
#define WINO4S_CASE(n) case n: conv_winograd36s_f32_kernel<16, false, n><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots); break;
            WINO4S_CASE(1) WINO4S_CASE(2) WINO4S_CASE(3) WINO4S_CASE(4) WINO4S_CASE(7) WINO4S_CASE(8) WINO4S_CASE(15) WINO4S_CASE(16) WINO4S_CASE(18) WINO4S_CASE(32) WINO4S_CASE(64)
            default: return CNM_ERR_BAD_ARG;
        }
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
#endif
    if (s2) {                                                            // stride 2 on the four pixel phases: 5x5 -> F(4x4,3x3), 7x7 -> F(3x3,4x4)
        if (M == 4) {
            if (tsx == 16) conv_winograd36s_f32_kernel<16, false, 0, 4, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
            else if (tsx == 4) conv_winograd36s_f32_kernel<4, false, 0, 4, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
            else conv_winograd36s_f32_kernel<8, false, 0, 4, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        } else {
            if (tsx == 16) conv_winograd36s_f32_kernel<16, false, 0, 3, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
            else if (tsx == 4) conv_winograd36s_f32_kernel<4, false, 0, 3, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
            else conv_winograd36s_f32_kernel<8, false, 0, 3, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        }
    } else if (M == 3) {                                                 // F(3x3,4x4) phase scatter: the data gradient of a stride-2 7x7
        if (tsx == 16) conv_winograd36s_f32_kernel<16, true, 0, 3><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        else conv_winograd36s_f32_kernel<8, true, 0, 3><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
    } else if (M == 2) {
        if (tsx == 16) conv_winograd36s_f32_kernel<16, false, 0, 2><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        else conv_winograd36s_f32_kernel<8, false, 0, 2><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
    } else if (tsx == 4) {
        conv_winograd36s_f32_kernel<4, false><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
    } else if (tsx == 16) {
        if (ups) conv_winograd36s_f32_kernel<16, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        else conv_winograd36s_f32_kernel<16, false><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
    } else {
        if (ups) conv_winograd36s_f32_kernel<8, true><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
        else conv_winograd36s_f32_kernel<8, false><<<grid, 512, 0, stream>>>(a, SH, SW, tilesC, (int)nunits, flags, slots);
    }
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
#endif   // WINO4S_ONE_INSTANCE
