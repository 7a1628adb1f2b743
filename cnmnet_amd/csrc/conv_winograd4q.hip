// Winograd F(4x4,3x3), LDS-staged, persistent, FOUR waves per workgroup -- one per SIMD, 512 registers each [r6].
//
// The answer to the round-5 go / no-go (profiles/r5_wino_tile_ablation.txt): in the eight-wave machine (conv_winograd4s.hip) a wave
// multiplies 16 output channels x 16 tiles, so every 1 KB weight fragment it pulls out of L2 feeds ONE group of MFMAs -- 295 KB of
// fragments per 9216 matrix-pipe cycles and CU, half of the CU's vector-memory path.  Here a wave multiplies 16 channels x 32 tiles
// (288 accumulator registers: only a wave alone on its SIMD has them), every fragment feeds TWO groups, and the stream per flop is
// halved.  What the tile costs is everything that is amortised over the output channels of a workgroup -- 64 here, 128 there: the
// input transform, the raw patch and the B-fragment reads are spent on half the MFMAs.  So the machine around the loop is built
// for one wave per SIMD:
//   * workgroup = 4 waves = 64 output channels x 32 tiles (2 x 16 or 4 x 8 tiles of 4 x 4 outputs); a phase = 8 input channels
//     (two c4 planes): 144 MFMAs (v_mfma_f32_16x16x4_f32, exact fp32) per wave, 4608 matrix-pipe cycles;
//   * no role split, no partner wave: every wave interleaves, between its OWN MFMAs, (i) the 18 weight fragments of the NEXT phase
//     (a whole phase of prefetch distance: 72 registers nobody else wants), (ii) the transform of ONE 6 x 6 window of the next
//     phase (thread = tile x channel: 36 ds_read_b32, 144 fp32 operations, 36 ds_write_b32), (iii) its six 1 KB pieces of the raw
//     patch two phases ahead;
//   * the raw patch travels through REGISTERS (buffer_load_dwordx4 -> ds_write_b128, each piece re-loaded right after it was
//     written: a full phase between a load and its use), not LDS-DMA: a single wave per SIMD has no partner to absorb the issue
//     hold of a DMA piece (the guide prices it at 60-185 cycles beside MFMAs; a plain 16-byte load at ~0), vmcnt retires in order,
//     and with the compiler seeing every load its counted waits are exact -- nothing hidden, no m0 juggling;
//   * LDS (120 KB): V[2][36 points][64 reader lanes][4] (72 KB) -- one ds_read_b128 per point gives a lane its four B operands
//     (tile half h x channel plane s) -- and RAW[2][2 planes][12 KB].  Raw pixels sit in COLUMN-QUAD-MAJOR slots (slot = row * RP
//     + (col & 3) * QP + (col >> 2)): the 32 lanes of a ds_read_b32 group are 4 neighbouring tiles x 2 planes x 4 channels, tiles
//     are 4 pixels = ONE slot apart, planes 4 slots (mod 8): 32 different banks.  V's reader slots are XOR-swizzled so that the
//     transform's ds_write_b32 are two-way (free) and the ds_read_b128 lane groups stay conflict-free;
//   * stream-K ranges, partial outputs and the hand-off protocol are those of the eight-wave machine (sync_ws.h), as are the
//     full-line output stores through a wave-private LDS transpose.
// The filter is the 36-point packed filter re-ordered for 8-channel phases (cnm_repack_winograd4_quad_f32): [8-channel chunk]
// [cout / 16][point pair][lane][4], lane (i = l & 15, kg = l >> 4), element 2 (x & 1) + s = U[x][co 16 cb + i][ci 8 chunk + 4 s + kg].
// Results are NOT bit-identical to the eight-wave kernel (the reduction over input channels is grouped 4 + 4 instead of strided),
// they are bit-reproducible run to run, with and without unit-cutting ranges.
#include "wino4_args.h"
#include "sync_ws.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define WQ_BT(x0, x1, x2, x3, x4, x5) do {                                                                      \
        const float t0 = fmaf(4.f, x0, fmaf(-5.f, x2, x4)), t5 = fmaf(4.f, x1, fmaf(-5.f, x3, x5));             \
        const float e1 = fmaf(-4.f, x2, x4), o1 = fmaf(-4.f, x1, x3);                                           \
        const float e2 = x4 - x2, o2 = x3 - x1;                                                                 \
        x0 = t0; x1 = e1 + o1; x2 = e1 - o1; x3 = fmaf(2.f, o2, e2); x4 = fmaf(-2.f, o2, e2); x5 = t5;          \
    } while (0)
// A^T of F(4,3) on four channels at once: rows [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] (the eight-wave kernel's expression)
#define WQ_AT(y0, y1, y2, y3, m0, m1, m2, m3, m4, m5) do {                                                      \
        const f32x4 s1 = (m1) + (m2), d1 = (m1) - (m2), s2 = (m3) + (m4), d2 = (m3) - (m4);                     \
        y0 = (m0) + s1 + s2; y1 = d1 + 2.f * d2; y2 = s1 + 4.f * s2; y3 = d1 + 8.f * d2 + (m5);                 \
    } while (0)

#ifndef WINO4Q_WD
#define WINO4Q_WD 18          // weight fragments in flight per wave (18: a whole phase ahead)
#endif
#ifndef WINO4Q_WRITE_STEP
#define WINO4Q_WRITE_STEP 12   // double step at which a wave starts moving its six raw pieces (one per step: ds_write_b128, then the re-load two phases ahead)
#endif

// The MFMAs are inline assembly so that the ACCUMULATOR CLASS is ours to choose: 288 accumulator registers are 256 AGPRs (points 0-31)
// + 32 VGPRs (points 32-35).  Left to the register allocator (the builtin), the overflow is handled by rotating accumulators between
// the two files -- ~340 v_accvgpr moves per 288 MFMAs in tools/wino_tile_ablation.hip, each a VALU instruction that takes fp32-lane
// cycles from the fp32 matrix pipe.  The compiler does not know these statements are MFMAs: every accumulator chain is MFMA -> MFMA
// on the same registers (interlocked by the hardware), the A / B operands come from memory instructions (counted waits), and the
// epilogue waits out the last MFMA's write-back with explicit s_nops before the first VALU touches an accumulator.
#ifndef WINO4Q_NACC_A
#define WINO4Q_NACC_A 32   // points whose accumulators live in AGPRs (8 registers each)
#endif
#define WQ_MFMA(ACC, X, H, AV, BV) do { if ((X) < WINO4Q_NACC_A) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC[X][H]) : "v"(AV), "v"(BV)); \
                                        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC[X][H]) : "v"(AV), "v"(BV)); } while (0)

// Filler schedule of the input transform (slots of 1 MFMA each, 144 per phase): its 144 VALU operations run in slots WQ_VS .. WQ_VS + WQ_VR - 1
// (WQ_VR = 1: ONE clump), the 36 point writes from slot WQ_WS on, spread over WQ_WR slots.
// Measured (tools/wino36q_variants.sh, profiles/r6_wino36q_variants.txt; 256 -> 512 channels at 48 x 64, 16 images): the operations spread at 1.26 per
// slot 0.374 ms, in blocks of 12-16 (the first build) 0.347, in ONE clump 0.324-0.329 against 0.286-0.289 without any transform.  An fp32 VALU
// instruction and the fp32 MFMA share the SIMD's fp32 lanes: every VALU instruction BETWEEN two MFMAs costs a drain and a refill of the matrix
// pipe on top of its own cycles (15 cycles per instruction when spread), a clump pays that once (6.6 per instruction).
#ifndef WQ_VS
#define WQ_VS 60
#define WQ_VR 1
#define WQ_WS 64
#define WQ_WR 76
#endif
template <int LO, int HI, class F> __device__ __forceinline__ void wq_unroll(F& f) { if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); wq_unroll<LO + 1, HI>(f); } }
template <int LO, int HI, class F> __device__ __forceinline__ void wq_unroll_if(F f) { if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); wq_unroll_if<LO + 1, HI>(f); } }

__device__ __forceinline__ void wq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ABL (debug builds -DWINO4Q_ABLATE, tools/wino36q_ablate.sh; results are wrong): bit 0 no input transform, 1 no raw loads / stores,
// 2 no weight loads in the loop, 3 no B-fragment reads in the loop, 4 no output transform / stores.
template <int TSX, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_winograd36q_f32_kernel(const Wino4Args a, const int SH, const int SW, const int ncblk, const int nunits,
                                                                       unsigned* __restrict__ sync_flags, float* __restrict__ sync_slots) {
    constexpr int TSY = 32 / TSX, PR = 4 * TSY + 2, PC = 4 * TSX + 2, QP = (PC + 3) / 4, RP = 4 * QP;
    constexpr int NPIECE = 12, PLANE1 = NPIECE * 1024 + 64, RAWBUF = PLANE1 + NPIECE * 1024;   // bytes; plane 1 starts 4 slots (mod 8) after plane 0
    constexpr int VBUF = 36 * 1024, RAW0 = 2 * VBUF;
    constexpr int NP = 18;                                               // point pairs = double steps = weight fragments per phase
    constexpr int SLOT_BYTES = 4 * 32 * 64 * 16;                         // one range's partial output: 4 waves x 32 pixels x 64 lanes x float4 = 128 KB
    static_assert(PR * RP < NPIECE * 64 && PR * PC <= NPIECE * 64 && (TSX == 16 || TSX == 8) && SLOT_BYTES == (int)kSyncSlotBytes, "layout");
    __shared__ __attribute__((aligned(16))) char smem[RAW0 + 2 * RAWBUF];   // 120.1 KB
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int HW = a.H * a.W, SHW = SH * SW, ncb16 = a.Cout / 16, nch = a.nchunks, nstrips = nunits / ncblk;

    // Work = the flat list of phases (unit, 8-channel chunk), unit = tile block x 64-channel block, channel block slowest; range r of
    // the G equal ranges belongs to this workgroup (conv_winograd4s.hip).
    const int G = gridDim.x, rng = xcd_remap(blockIdx.x, G);
    const long long T = (long long)nunits * nch;
    const auto range_begin = [&](int r) { return sync_flags ? (int)(T * r / G) : (int)((long long)nunits * r / G) * nch; };
    const int ps = range_begin(rng), pe = range_begin(rng + 1);
    const int P = pe - ps;
    if (P <= 0) return;

    // ---- raw patch: pieces k = wave, wave + 4, wave + 8 of either plane; lane = slot 64 k + lane
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2), 0, a.in2_bytes, 0x00020000);
    const unsigned hw16 = (unsigned)HW * 16u;
    int lu = ps / nch, lc = ps - lu * nch, lgp = ps;                     // cursor of the load role: unit, chunk, global phase
    unsigned lbase1 = 0, lbase2 = 0, lvoff[3];
    unsigned lslot[3];                                                   // LDS byte offset (within a plane) of the pixel this lane moves in piece wave + 4 m
    auto load_unit = [&]() {                                             // per-lane pixel offsets of unit lu: lane = pixel 64 k + lane of the patch, row-major (coalesced rows)
        const int strip = lu % nstrips, limg = strip / SHW, rem = strip - limg * SHW, sy = rem / SW, sx = rem - sy * SW;
        lbase1 = (unsigned)(limg * a.Gin_tot + a.gin0) * hw16;
        lbase2 = (unsigned)(limg * a.Gin2_tot + a.gin2_0 - a.Gsplit) * hw16;
        const int y0 = 4 * TSY * sy - 1, x0 = 4 * TSX * sx - 1;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int pix = 64 * (wave + 4 * m) + lane, r = pix / PC, c = pix - r * PC;
            const int y = y0 + r, x = x0 + c;
            const bool ok = (r < PR) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            lvoff[m] = ok ? (unsigned)(y * a.W + x) * 16u : 0xFFFFFFFFu;
        }
    };
#pragma unroll
    for (int m = 0; m < 3; ++m) {                                        // pixels past the patch (the last pieces' tails) land in the plane's spare slots
        const int pix = 64 * (wave + 4 * m) + lane, r = pix / PC, c = pix - r * PC;
        lslot[m] = (unsigned)(r < PR ? r * RP + (c & 3) * QP + (c >> 2) : PR * RP + (pix - PR * PC) % (NPIECE * 64 - PR * RP)) * 16u;   // (zeros on zeros)
    }
    u32x4 rawr[6];                                                       // the wave's six pieces in flight
    auto load_piece = [&](int m) {                                       // piece m (plane m / 3) of phase (lu, lc)
        if (ABL & 2) return;
        const int g = 2 * lc + m / 3;
        const bool s1 = g < a.Gsplit;
        const unsigned voff = (g < a.Gin && lgp < pe) ? lvoff[m % 3] : 0xFFFFFFFFu;   // past the range / past the input's groups: out of range = zeros, no branch
        rawr[m] = __builtin_amdgcn_raw_buffer_load_b128(s1 ? rsrc1 : rsrc2, voff, (s1 ? lbase1 : lbase2) + (unsigned)g * hw16, 0);
    };
    auto load_advance = [&]() { ++lgp; if (++lc == nch) { lc = 0; ++lu; if (lgp < pe) load_unit(); } };
    auto store_piece = [&](int m, int buf) {
        if (ABL & 2) return;
        *reinterpret_cast<u32x4*>(smem + RAW0 + buf * RAWBUF + (m / 3) * PLANE1 + lslot[m % 3]) = rawr[m];
    };
    // The same inside the phase loop, with everything that is not a memory instruction hoisted to the TOP of the phase [r6]: the first build
    // worked out the plane's view, descriptor, scalar offset and validity next to every piece -- ~16 SALU + 2 VALU instructions between two
    // MFMAs, three times the 32-cycle gap, six times per phase: the raw pieces cost 21 % of the launch (tools/wino36q_ablate.sh).  Per phase now:
    // eight pinned scalars (descriptor base / size and scalar offset of either plane; an invalid plane is a ZERO-SIZE buffer, so its loads
    // return zeros without a per-lane select) and three LDS write addresses; a piece is s_waitcnt + ds_write_b128 + buffer_load.
    unsigned in1_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in), in1_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in) >> 32);
    unsigned in2_lo = (unsigned)reinterpret_cast<unsigned long long>(a.in2), in2_hi = (unsigned)(reinterpret_cast<unsigned long long>(a.in2) >> 32);
    unsigned pq_lo[2], pq_hi[2], pq_bytes[2], pq_soff[2], pq_wa[3];
    auto phase_scalars = [&](int buf) {                                  // of the phase the load cursor points at (lu, lc, lgp); stores go to RAW[buf]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = 2 * lc + q;
            const bool s1 = g < a.Gsplit, ok = g < a.Gin && lgp < pe;
            pq_lo[q] = s1 ? in1_lo : in2_lo; pq_hi[q] = s1 ? in1_hi : in2_hi;
            pq_bytes[q] = ok ? (s1 ? a.in_bytes : a.in2_bytes) : 0u;
            pq_soff[q] = (s1 ? lbase1 : lbase2) + (unsigned)g * hw16;
            asm volatile("" : "+s"(pq_lo[q]), "+s"(pq_hi[q]), "+s"(pq_bytes[q]), "+s"(pq_soff[q]));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { pq_wa[j] = lslot[j] + (unsigned)(RAW0 + buf * RAWBUF); asm volatile("" : "+v"(pq_wa[j])); }
    };
    auto move_piece = [&](int m) {                                       // piece m: out of its registers into RAW, and the next phase's into the registers
        if (ABL & 2) return;
        const int q = m / 3, j = m % 3;
        *reinterpret_cast<u32x4*>(smem + pq_wa[j] + q * PLANE1) = rawr[m];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((unsigned long long)pq_lo[q] | ((unsigned long long)pq_hi[q] << 32)), 0, pq_bytes[q], 0x00020000);
        rawr[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, lvoff[j], pq_soff[q], 0);
    };

    // ---- transform role: thread = one window (tile tt, channel 4 s + comp of the chunk)
    const int comp = lane & 3, txl = (lane >> 2) & 3, tsl = (lane >> 4) & 1, sel = (lane >> 5) | (wave << 1);
    const int tty = TSX == 16 ? (sel & 1) : (sel >> 1), ttx = 4 * (TSX == 16 ? (sel >> 1) : (sel & 1)) + txl, tt = tty * TSX + ttx;
    const unsigned rbase = RAW0 + tsl * PLANE1 + ((4 * tty) * RP + ttx) * 16 + comp * 4;
    const unsigned wbase = ((((tt & 15) ^ ((comp & 1) * 12)) + 16 * comp) * 4 + 2 * (tt >> 4) + tsl) * 4;
    float d[36];                                                         // d[6 j + i]: column j, row i of the window
    auto tr_read = [&](int idx, unsigned rb) {                           // column-major order: idx = 6 j + i
        if (ABL & 1) return;
        const int j = idx / 6, i = idx - 6 * j;
        d[idx] = *reinterpret_cast<const float*>(smem + rb + (i * RP + (j & 3) * QP + (j >> 2)) * 16);
    };
    auto tr_col = [&](int j) { if (ABL & 1) return; WQ_BT(d[6 * j + 0], d[6 * j + 1], d[6 * j + 2], d[6 * j + 3], d[6 * j + 4], d[6 * j + 5]); };
    auto tr_row = [&](int i, unsigned wb) {                              // row pass of frequency row i and its six points
        if (ABL & 1) return;
        WQ_BT(d[0 + i], d[6 + i], d[12 + i], d[18 + i], d[24 + i], d[30 + i]);
#pragma unroll
        for (int l = 0; l < 6; ++l) *reinterpret_cast<float*>(smem + wb + (6 * i + l) * 1024) = d[6 * l + i];
    };

    // ---- multiply role
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, (unsigned)((size_t)nch * ncb16 * NP * 1024), 0x00020000);
    const unsigned lane16 = lane * 16;
    const unsigned bvoff = (((lane & 15) ^ (((lane >> 4) & 1) * 12)) + 16 * (lane >> 4)) * 16;
    auto ldA = [&](unsigned soff) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, soff, 0); return *reinterpret_cast<const f32x4*>(&v); };
    auto abase = [&](int cblk, int c) { return (unsigned)((c * ncb16 + cblk * 4 + wave) * NP) * 1024u; };
    f32x4 acc[36][2];
#pragma unroll
    for (int x = 0; x < 36; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue: raw(0) -> RAW[0] -> V[0]; raw(1) -> RAW[1]; raw(2) in flight; the weight fragments of phase 0
    int mu = ps / nch, mc = ps - mu * nch;                               // cursor of the multiply role
    int mcblk = mu / nstrips;
    int part_c0 = mc;
    unsigned a_cur = abase(mcblk, mc);
    constexpr int WD = WINO4Q_WD;                                        // weight fragments in flight per wave: the prefetch distance in double steps (divides 18)
    static_assert(NP % WD == 0, "ring");
    f32x4 af[WD];
#pragma unroll
    for (int s = 0; s < WD; ++s) af[s] = (ABL & 4) ? f32x4{1.f, 0.5f, 0.25f, 2.f} : ldA(a_cur + s * 1024);
    load_unit();
#pragma unroll
    for (int m = 0; m < 6; ++m) load_piece(m);
    load_advance();
#pragma unroll
    for (int m = 0; m < 6; ++m) store_piece(m, 0);
#pragma unroll
    for (int m = 0; m < 6; ++m) load_piece(m);
    load_advance();
    wq_lds_barrier();
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) tr_read(idx, rbase);
#pragma unroll
    for (int j = 0; j < 6; ++j) tr_col(j);
#pragma unroll
    for (int i = 0; i < 6; ++i) tr_row(i, wbase);
#pragma unroll
    for (int m = 0; m < 6; ++m) store_piece(m, 1);
#pragma unroll
    for (int m = 0; m < 6; ++m) load_piece(m);
    load_advance();
    wq_lds_barrier();

    int p = 0;
    while (p < P) {
      bool lastc;
      int ncblk_;
      for (;;) {                                                         // the phases of one part: nothing but the phase body lives in this loop
        const unsigned vc = (p & 1) * VBUF + bvoff;
        const unsigned rb = rbase + ((p + 1) & 1) * RAWBUF;              // raw buffer of phase p + 1
        const unsigned wb = wbase + ((p + 1) & 1) * VBUF;                // V buffer of phase p + 1
        lastc = mc + 1 == nch;
        ncblk_ = mcblk;
        if (lastc) ncblk_ = (mu + 1) / nstrips;
        const unsigned a_nxt = p + 1 < P ? (lastc ? abase(ncblk_, 0) : a_cur + (unsigned)(ncb16 * NP) * 1024u) : a_cur;
        phase_scalars(p & 1);
        f32x4 bq[2][2];                                                  // B fragments: [step parity][point of the pair]
        bq[0][0] = *reinterpret_cast<const f32x4*>(smem + vc); bq[0][1] = *reinterpret_cast<const f32x4*>(smem + vc + 1024);
        __builtin_amdgcn_sched_barrier(0);
        // Everything that is not an MFMA is a FILLER with a slot: slot g = 8 xp + k follows MFMA k of double step xp, and nothing
        // crosses a slot boundary (sched_barrier).  A matrix-pipe gap is 32 cycles: a clump of a dozen fillers behind one MFMA
        // idles the pipe for what exceeds the gap, and in-order issue never gets it back (the first build placed the transform
        // in blocks of 12-16 instructions: transform, raw pieces and weight loads each cost 14-16 % of the launch).  Schedule:
        //   B fragments of step xp + 1          slots 8 xp, 8 xp + 1
        //   weight fragment of the next phase   slot 8 xp + 7 (behind the pair's last MFMA: loaded into the registers it is read from)
        //   window reads (36)                   slots 0 .. 35
        //   transform operations (144)          slots 24 .. 137, 1.26 per slot, columns then rows; each point is written two slots
        //                                       after the operation that finishes it
        //   raw piece m: ds_write + re-load     slot 8 (12 + m) + 4
        float tmp[6];
        auto tr_op = [&](auto N_) {                                      // operation N of the 144: pass N / 12 -- column 0-5 on d[6 j + 0..5], then row 0-5 on d[i], d[6 + i], ... -- step N % 12
            constexpr int n = decltype(N_)::value, ps_ = n / 12, o = n - 12 * ps_, st = ps_ < 6 ? 1 : 6, b = ps_ < 6 ? 6 * ps_ : ps_ - 6;
            float &x0 = d[b], &x1 = d[b + st], &x2 = d[b + 2 * st], &x3 = d[b + 3 * st], &x4 = d[b + 4 * st], &x5 = d[b + 5 * st];
            if constexpr (o == 0) tmp[0] = fmaf(-5.f, x2, x4);
            else if constexpr (o == 1) tmp[1] = fmaf(-5.f, x3, x5);
            else if constexpr (o == 2) tmp[2] = fmaf(-4.f, x2, x4);
            else if constexpr (o == 3) tmp[3] = fmaf(-4.f, x1, x3);
            else if constexpr (o == 4) tmp[4] = x4 - x2;
            else if constexpr (o == 5) tmp[5] = x3 - x1;
            else if constexpr (o == 6) x0 = fmaf(4.f, x0, tmp[0]);
            else if constexpr (o == 7) x5 = fmaf(4.f, x1, tmp[1]);
            else if constexpr (o == 8) x1 = tmp[2] + tmp[3];
            else if constexpr (o == 9) x2 = tmp[2] - tmp[3];
            else if constexpr (o == 10) x3 = fmaf(2.f, tmp[5], tmp[4]);
            else x4 = fmaf(-2.f, tmp[5], tmp[4]);
        };
        auto tr_write = [&](auto W_) {                                   // write w = 6 i + q of the 36: row i, the point its operation 6 + q finishes (6 -> point 0, 7 -> 5, 8-11 -> 1-4)
            constexpr int w = decltype(W_)::value, i = w / 6, q = w - 6 * i, l = q == 0 ? 0 : q == 1 ? 5 : q - 1;
            *reinterpret_cast<float*>(smem + wb + (6 * i + l) * 1024) = d[6 * l + i];
        };
        auto fill = [&](auto G_) {
            constexpr int g = decltype(G_)::value, xp = g >> 3, k = g & 7;
            if constexpr (k < 2 && xp + 1 < NP && !(ABL & 8)) bq[(xp + 1) & 1][k] = *reinterpret_cast<const f32x4*>(smem + vc + (2 * xp + 2 + k) * 1024);
            if constexpr (g < 36) tr_read(g, rb);
            if constexpr (!(ABL & 1)) {
                if constexpr (g >= WQ_VS && g < WQ_VS + WQ_VR) {
                    constexpr int n_lo = ((g - WQ_VS) * 144 + WQ_VR - 1) / WQ_VR, n_hi = ((g + 1 - WQ_VS) * 144 + WQ_VR - 1) / WQ_VR;
                    wq_unroll<n_lo, (n_hi < 144 ? n_hi : 144)>(tr_op);
                }
                // write w goes out at slot max(two slots behind its operation, WQ_WS + w WQ_WR / 36)
                wq_unroll_if<0, 36>([&](auto W_) { constexpr int w = decltype(W_)::value, n = 72 + 12 * (w / 6) + 6 + w % 6, so = WQ_VS + n * WQ_VR / 144 + 2, sw = WQ_WS + w * WQ_WR / 36;
                                                   if constexpr ((so > sw ? so : sw) == g) tr_write(W_); });
            }
            if constexpr (k == 4 && xp >= 12) {
                move_piece(xp - 12);
                if constexpr (xp == 17) load_advance();
            }
        };
        auto step = [&](auto XP_) {
            constexpr int xp = decltype(XP_)::value, x0 = 2 * xp, x1 = x0 + 1;
            const f32x4 av = af[xp % WD], c0 = bq[xp & 1][0], c1 = bq[xp & 1][1];
#define WQ_SLOT(K, X, H, A, B) WQ_MFMA(acc, X, H, A, B); fill(std::integral_constant<int, 8 * xp + K>{}); __builtin_amdgcn_sched_barrier(0)
            WQ_SLOT(0, x0, 0, av[0], c0[0]);
            WQ_SLOT(1, x1, 0, av[2], c1[0]);
            WQ_SLOT(2, x0, 1, av[0], c0[2]);
            WQ_SLOT(3, x1, 1, av[2], c1[2]);
            WQ_SLOT(4, x0, 0, av[1], c0[1]);
            WQ_SLOT(5, x1, 0, av[3], c1[1]);
            WQ_SLOT(6, x0, 1, av[1], c0[3]);
            WQ_MFMA(acc, x1, 1, av[3], c1[3]);
            fill(std::integral_constant<int, 8 * xp + 7>{});
            if constexpr (!(ABL & 4)) af[xp % WD] = xp + WD < NP ? ldA(a_cur + (xp + WD) * 1024) : ldA(a_nxt + (xp + WD - NP) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#undef WQ_SLOT
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
        wq_lds_barrier();                                                // V / RAW of phase p + 1 complete; the buffers of phase p are free
        a_cur = a_nxt;
        if (lastc || p + 1 >= P) break;
        ++mc; ++p;
      }

        // ---- a part of unit mu ends here (chunks part_c0 .. mc): whole unit or head part -> finish (adding the following ranges'
        // partial outputs in range order); any other part -> publish the partial output transform in this range's slot
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                // the last MFMAs' write-back (16 passes at most) before any VALU reads or zeroes an accumulator
        const bool publish = part_c0 != 0;
        int nsrc = 0;
        if (!publish && !lastc) {
            for (int rem = nch - 1 - mc; rem > 0; ++nsrc) rem -= range_begin(rng + nsrc + 2) - range_begin(rng + nsrc + 1);
            if (t == 0) {
                const unsigned gen = sync_generation();
                for (int k = 1; k <= nsrc; ++k) sync_wait(sync_flags, rng + k, gen);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        if (ABL & 16) {
#pragma unroll
            for (int x = 0; x < 36; ++x) { asm volatile("" :: "v"(acc[x][0]), "v"(acc[x][1])); acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        } else {
            int le = lane; asm volatile("" : "+v"(le));                  // opaque: nothing below is loop-invariant to the compiler
            const int rtile = le & 15, kg = le >> 4;
            const int strip = mu % nstrips;
            const int img = strip / SHW, rem = strip - img * SHW, sy = rem / SW, sx = rem - sy * SW;
            const int co = mcblk * 64 + wave * 16 + 4 * kg;
            const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const f32x4 bb = {bv.x, bv.y, bv.z, bv.w};
            const unsigned slot_lane = (unsigned)(wave * 32 * 64 + le);   // partial outputs: [range][wave][pixel 16 h + 4 st + x][lane] float4
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sync_slots, 0, sync_slots ? (unsigned)G * (unsigned)SLOT_BYTES : 0u, 0x00020000);
            // full-line stores through a wave-private 4.25 KB corner of the V buffer the multiply role has just released:
            // [channel group][pixel slot][tile], slot pitch 272 bytes, then lane = pixel (conv_winograd4s.hip)
            const unsigned stg = (unsigned)((p & 1) * VBUF + wave * 4352);
            const unsigned stw = stg + kg * 1088 + rtile * 16;
            const int qt = le >> 2, qs = le & 3;
            const unsigned str_ = stg + qs * 272 + qt * 16;
            const int cow = mcblk * 64 + wave * 16;
            float* const obase = a.out + c4_offset(img, a.Gout_tot, a.gout0 + (cow >> 2), HW, 0);
            const size_t gstride = (size_t)HW * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qtt = 16 * h + qt, qty = sy * TSY + qtt / TSX, qtx = sx * TSX + qtt % TSX;
                const int qcol = 4 * qtx + qs;
                const bool qok = qty < a.TH && qtx < a.TW;
                f32x4 s[4][6];                                           // A^T M: 4 rows x 6 columns
#pragma unroll
                for (int j = 0; j < 6; ++j) WQ_AT(s[0][j], s[1][j], s[2][j], s[3][j], acc[0 * 6 + j][h], acc[1 * 6 + j][h], acc[2 * 6 + j][h], acc[3 * 6 + j][h], acc[4 * 6 + j][h], acc[5 * 6 + j][h]);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    f32x4 y[4];
                    WQ_AT(y[0], y[1], y[2], y[3], s[st][0], s[st][1], s[st][2], s[st][3], s[st][4], s[st][5]);
                    if (publish) {                                       // write-through (sc1) stores: no release fence needed before the flag
#pragma unroll
                        for (int x = 0; x < 4; ++x)
                            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&y[x]), srsrc, (slot_lane + (unsigned)(16 * h + 4 * st + x) * 64u) * 16u, (unsigned)rng * (unsigned)SLOT_BYTES, 16);
                        __builtin_amdgcn_sched_barrier(0);               // store-data hazard of the SGPR-soffset form (conv_winograd4s.hip)
                        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    for (int k = 1; k <= nsrc; ++k) {                    // fixed order: own part, then the following ranges (sc1 loads)
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (slot_lane + (unsigned)(16 * h + 4 * st + x) * 64u) * 16u, (unsigned)(rng + k) * (unsigned)SLOT_BYTES, 16);
                            y[x] += *reinterpret_cast<const f32x4*>(&pv);
                        }
                    }
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        f32x4 v = y[x] + bb;
                        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        *reinterpret_cast<f32x4*>(smem + stw + x * 272) = v;
                    }
                    const int qrow = 4 * qty + st;
                    const bool stv = qok && qrow < a.H && qcol < a.W;
                    float* const orow = obase + (size_t)(qrow * a.W + qcol) * 4;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {                        // LDS operations of one wave complete in order
                        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + str_ + g * 1088);
                        if (stv) *reinterpret_cast<f32x4*>(orow + g * gstride) = v;
                    }
                }
            }
            if (publish) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) sync_publish(sync_flags, rng, sync_generation());
            }
#pragma unroll
            for (int x = 0; x < 36; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            wq_lds_barrier();                                            // staging corners done before the next phase's transform writes that V buffer
        }
        // gfx9 counts loads and stores in ONE vmcnt and the compiler assumes they retire out of order with respect to each other: with
        // the epilogue's stores possibly pending at the top of the phase loop, EVERY wait for a weight fragment in it became vmcnt(0)
        // (checked in the ISA) -- the whole-phase prefetch serialised.  Drained here (a builtin: the compiler's counter model sees it).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (lastc) { ++mu; mcblk = ncblk_; }
        mc = 0; part_c0 = 0; ++p;
    }
}

// The 36-point packed filter (cnm_pack_winograd4_bn_f32: [16-channel chunk][cout / 16][point][lane][4], element e of lane (i, kg) =
// U[x][16 cb + i][16 chunk + 4 kg + e]) re-ordered for 8-channel phases -- a pure permutation, the same values.
__global__ __launch_bounds__(256) void repack_winograd4_quad_kernel(const float* __restrict__ u, float* __restrict__ uq, int ncb16, int nch8) {
    const int blk = blockIdx.x;                                          // (c8, cb16, pair)
    const int pair = blk % 18, cb = (blk / 18) % ncb16, c8 = blk / (18 * ncb16);
    const int t = threadIdx.x, e = t & 3, lane = t >> 2;
    const int x = 2 * pair + (e >> 1), s = e & 1, i = lane & 15, kg = lane >> 4;
    const int c16 = c8 >> 1, kg16 = 2 * (c8 & 1) + s, e16 = kg;          // channel 8 c8 + 4 s + kg = 16 c16 + 4 kg16 + e16
    uq[(size_t)blk * 256 + t] = u[(((size_t)c16 * ncb16 + cb) * 36 + x) * 256 + (i + 16 * kg16) * 4 + e16];
}

extern "C" size_t cnm_packed_winograd4_quad_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
    const int nch8 = ((Cin + 3) / 4 + 1) / 2;
    return (size_t)nch8 * (Cout / 16) * 18 * 256;
}

extern "C" int cnm_repack_winograd4_quad_f32(const float* u_packed, int Cout, int Cin, float* uq_packed, void* stream) {
    CNM_REQUIRE(u_packed && uq_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0, CNM_ERR_BAD_ARG);
    const int nch8 = ((Cin + 3) / 4 + 1) / 2, ncb16 = Cout / 16;
    repack_winograd4_quad_kernel<<<(unsigned)(nch8 * ncb16 * 18), 256, 0, cnm_stream(stream)>>>(u_packed, uq_packed, ncb16, nch8);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

static int g_wino36_quad = 1;                                            // tuning knob: 0 never, 1 where the executors' table says it pays, 2 wherever eligible
extern "C" int cnm_tune_wino36_quad(int mode) { const int old = g_wino36_quad; if (mode >= 0 && mode <= 2) g_wino36_quad = mode; return old; }
int cnm_wino36_quad_mode() { return g_wino36_quad; }

#ifdef WINO4Q_ABLATE
static int g_wino36q_ablate = 0;
extern "C" int cnm_tune_wino36q_ablate(int m) { const int old = g_wino36q_ablate; if (m >= 0) g_wino36q_ablate = m; return old; }
#endif

static int wq_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) { int n = 0; cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    return cus[dev];
}

// 1 = shape not eligible (2 x 16 tile blocks want >= 12 tile columns and >= 2 tile rows, 4 x 8 blocks >= 6 columns and >= 3 rows),
// CNM_OK after launching, a negative status otherwise.  a.u = the quad-packed filter, a.nchunks = 8-channel chunks.
int cnm_wino36q_try_launch(const Wino4Args& a, hipStream_t stream) {
    if (a.Cout % 64) return 1;
    int tsx = 0;
    if (a.TW >= 12 && a.TH >= 2) tsx = 16; else if (a.TW >= 6 && a.TH >= 3) tsx = 8;
    if (!tsx) return 1;
    const int tsy = 32 / tsx;
    const int SH = cnm_ceil_div(a.TH, tsy), SW = cnm_ceil_div(a.TW, tsx), ncblk = a.Cout / 64;
    const long long nunits = (long long)a.N * SH * SW * ncblk;
    if (nunits <= 0 || nunits * a.nchunks > 0x7FFFFFFF || (long long)a.nchunks * (a.Cout / 16) * 18 * 1024 >= 0xFFFFFFFFll) return 1;
    const int cus = wq_cus();
    int grid = (int)(nunits < cus ? nunits : cus);
    unsigned* flags = nullptr; float* slots = nullptr;
    sync_ctl_upload(stream);
    if (cnm_sync_failed()) return CNM_ERR_LAUNCH;
    if (a.sync_ws && cus <= kSyncMaxRanges && a.sync_floats * 4 >= kSyncFlagBytes + (size_t)cus * kSyncSlotBytes) {
        const long long T = nunits * a.nchunks;                          // at least eight phases per range: the prologue and the fix-up of a range are worth two
        grid = (int)(T / 8 < cus ? (T / 8 > 0 ? T / 8 : 1) : cus);
        flags = reinterpret_cast<unsigned*>(a.sync_ws);
        slots = a.sync_ws + kSyncFlagBytes / 4;
    }
#ifdef WINO4Q_ABLATE
    if (g_wino36q_ablate && tsx == 16) {
        switch (g_wino36q_ablate) {
#define WQ_CASE(n) case n: conv_winograd36q_f32_kernel<16, n><<<grid, 256, 0, stream>>>(a, SH, SW, ncblk, (int)nunits, flags, slots); break;
            WQ_CASE(1) WQ_CASE(2) WQ_CASE(3) WQ_CASE(4) WQ_CASE(7) WQ_CASE(8) WQ_CASE(15) WQ_CASE(16) WQ_CASE(31)
            default: return CNM_ERR_BAD_ARG;
        }
        CNM_LAUNCH_CHECK();
        return CNM_OK;
    }
#endif
    if (tsx == 16) conv_winograd36q_f32_kernel<16><<<grid, 256, 0, stream>>>(a, SH, SW, ncblk, (int)nunits, flags, slots);
    else conv_winograd36q_f32_kernel<8><<<grid, 256, 0, stream>>>(a, SH, SW, ncblk, (int)nunits, flags, slots);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv3x3_winograd4q_ok(int Cout, int H, int W) {
    const int TH = (H + 3) / 4, TW = (W + 3) / 4;
    return Cout > 0 && Cout % 64 == 0 && ((TW >= 12 && TH >= 2) || (TW >= 6 && TH >= 3));
}

// Conv2d(3x3, stride 1, pad 1) + folded BatchNorm + ReLU on one or two c4 views, four-wave F(4x4,3x3) kernel.  u_packed_quad from
// cnm_repack_winograd4_quad_f32; sync workspace as for cnm_conv3x3_winograd4_sync_c4_f32 (NULL / 0: ranges end on unit boundaries).
extern "C" int cnm_conv3x3_winograd4q_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
                                                  float* out, int Gout_total, int gout0, int Cout, const float* u_packed_quad, const float* b_packed,
                                                  int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    CNM_REQUIRE(in_a && out && u_packed_quad && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(cnm_conv3x3_winograd4q_ok(Cout, H, W) && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    Wino4Args a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed_quad; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    a.N = N; a.H = H; a.W = W; a.TH = (H + 3) / 4; a.TW = (W + 3) / 4;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (a.Gin + 1) / 2; a.T = N * a.TH * a.TW; a.relu = relu; a.ring = 0; a.ups_zero = 0;
    a.sync_ws = sync_ws; a.sync_floats = sync_floats;
    const int e = cnm_wino36q_try_launch(a, cnm_stream(stream));
    return e == 1 ? CNM_ERR_BAD_ARG : e;
}
