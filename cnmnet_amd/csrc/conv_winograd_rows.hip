// Row-wise Winograd convolution for the 5x5 / 7x7 layers (depthNet conv1.*, conv2.*: 42 % of a frame's conv FLOPs),
// fp32 MFMA.  template <R, S, M>: filter size, stride, outputs per tile along the row.
//
// A 2-D F(m x m, R x R) needs (m+R-1)^2 frequency points -- for R = 7 more accumulators than a wave has registers --
// so the transform runs along image rows only and the R kernel rows stay in the GEMM reduction:
//     out[y, M t .. M t + M-1] = AT  sum_{ky, ci} [ (G w[co, ci, ky, :]) (.) (BT in[ci, y+ky-R/2, M t - R/2 .. ]) ]
//   (M+R-1)/M multiplies per output and kernel row instead of R:  F(2,7) 4, F(4,7) 2.5 (conv1.0), F(2,5) 3;
//   stride 2 (S = 2): the two column phases of an input row are stride-1 correlations with ceil(R/2) taps, both run
//   through the same F(M,ceil(R/2)) and accumulate in the same frequency-domain registers: F(4,4) 3.5 instead of 7
//   (conv1.3), F(4,3) 3 instead of 5 (conv2.3).  fp32 data and accumulation; Toom-Cook tables from
//   tools/wino1d_matrices.py (exact in rationals); measured per-layer fp32 error: tiles of 2 ~5e-5, F(4,7) ~3e-4 on
//   O(1) outputs (the training path keeps tiles of 2).
//
// Same machine as conv_winograd.hip: workgroup = 4 waves = 64 couts x 64 (48 for F(4,7)) row-tiles, TWO workgroups per
// CU; every wave 16 couts x all tiles x ALL frequency points on v_mfma_f32_16x16x4_f32 (<= 128 accumulator
// registers); per 16-deep chunk of the (ky [, phase], ci) reduction every thread gathers the window of ONE (tile,
// channel-quad) with buffer loads (out of range = 0), transforms it in registers and writes V[xi][tile][k] to LDS
// (double buffered, swizzled); per frequency point one 16-byte weight fragment straight from L2 (MFMA operand order,
// private to the wave) and one ds_read_b128 of V per 16-tile block feed 12-16 MFMAs; the transform of chunk c+1 and the
// gather of chunk c+2 ride between the MFMAs of chunk c; one LDS-only barrier per chunk.
#include "cnm_common.h"
#include "rows_args.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt -- here that would wait, at
// every chunk, for the weight fragments and windows deliberately left in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// float4 as two packed pairs: the arithmetic below compiles to v_pk_{add,mul,fma}_f32
typedef float v2f __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(16))) f4p { v2f lo, hi; };
__device__ __forceinline__ f4p f4_add(f4p a, f4p b) { return {a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ f4p f4_sub(f4p a, f4p b) { return {a.lo - b.lo, a.hi - b.hi}; }
__device__ __forceinline__ f4p f4_mul(float c, f4p a) { const v2f cc = {c, c}; return {cc * a.lo, cc * a.hi}; }
__device__ __forceinline__ f4p f4_fma(float c, f4p a, f4p b) {
    const v2f cc = {c, c};
    return {__builtin_elementwise_fma(cc, a.lo, b.lo), __builtin_elementwise_fma(cc, a.hi, b.hi)};
}

template <int R> struct RowWino;
// F(2,5), interpolation points 0, 1, -1, 2, -2, inf
template <> struct RowWino<5> {
    static constexpr float BT[6][6]  /* documentation: the kernel uses the factored form in transform_group */ = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    static constexpr float AT1[6] = {0, 1, -1, 2, -2, 1};   // AT0 = {1, ..., 1, 0}
    static constexpr double G[6][5] = {{1. / 4, 0, 0, 0, 0}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6, 1. / 3, 2. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3, 2. / 3}, {0, 0, 0, 0, 1}};
};
// F(2,7), interpolation points 0, 1, -1, 2, -2, 1/2, -1/2, inf
template <> struct RowWino<7> {
    static constexpr float BT[8][8] = {{-1, 0, 21. / 4, 0, -21. / 4, 0, 1, 0}, {0, 1, 1, -17. / 4, -17. / 4, 1, 1, 0}, {0, -1, 1, 17. / 4, -17. / 4, -1, 1, 0}, {0, 1. / 2, 1. / 4, -5. / 2, -5. / 4, 2, 1, 0}, {0, -1. / 2, 1. / 4, 5. / 2, -5. / 4, -2, 1, 0}, {0, 2, 4, -5. / 2, -5, 1. / 2, 1, 0}, {0, -2, 4, 5. / 2, -5, -1. / 2, 1, 0}, {0, -1, 0, 21. / 4, 0, -21. / 4, 0, 1}};
    static constexpr float AT1[8] = {0, 1, -1, 2, -2, 1. / 2, -1. / 2, 1};   // AT0 = {1, ..., 1, 0}
    static constexpr double G[8][7] = {{-1, 0, 0, 0, 0, 0, 0}, {-2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9}, {-2. / 9, 2. / 9, -2. / 9, 2. / 9, -2. / 9, 2. / 9, -2. / 9}, {1. / 90, 1. / 45, 2. / 45, 4. / 45, 8. / 45, 16. / 45, 32. / 45}, {1. / 90, -1. / 45, 2. / 45, -4. / 45, 8. / 45, -16. / 45, 32. / 45}, {32. / 45, 16. / 45, 8. / 45, 4. / 45, 2. / 45, 1. / 45, 1. / 90}, {32. / 45, -16. / 45, 8. / 45, -4. / 45, 2. / 45, -1. / 45, 1. / 90}, {0, 0, 0, 0, 0, 0, 1}};
};

// F(2,4), interpolation points 0, 1, -1, 2, inf  (even / odd column phases of the 7-tap stride-2 rows)
template <> struct RowWino<4> {
    static constexpr float AT1[5] = {0, 1, -1, 2, 1};
    static constexpr double G[5][4] = {{1. / 2, 0, 0, 0}, {-1. / 2, -1. / 2, -1. / 2, -1. / 2}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6}, {1. / 6, 1. / 3, 2. / 3, 4. / 3}, {0, 0, 0, 1}};
};
// F(2,3), interpolation points 0, 1, -1, inf  (column phases of the 5-tap stride-2 rows)
template <> struct RowWino<3> {
    static constexpr float AT1[4] = {0, 1, -1, 1};
    static constexpr double G[4][3] = {{-1, 0, 0}, {1. / 2, 1. / 2, 1. / 2}, {1. / 2, -1. / 2, 1. / 2}, {0, 0, 1}};
};

// F(4,7), interpolation points 0, +-1, +-2, +-1/2, +-3/2, inf: 10 multiplies per 4 outputs and kernel row (conv1.0).
// Measured fp32 error of the 1-D algorithm 2.5e-5 on O(1) outputs (F(2,7): 5.7e-6); tools/wino1d_matrices.py.
template <> struct RowWino<74> {
    static constexpr float BT[10][10] = {
        {9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1, 0},
        {0, -9. / 4, -9. / 4, 169. / 16, 169. / 16, -13. / 2, -13. / 2, 1, 1, 0}, {0, 9. / 4, -9. / 4, -169. / 16, 169. / 16, 13. / 2, -13. / 2, -1, 1, 0},
        {0, -9. / 8, -9. / 16, 49. / 8, 49. / 16, -7, -7. / 2, 2, 1, 0}, {0, 9. / 8, -9. / 16, -49. / 8, 49. / 16, 7, -7. / 2, -2, 1, 0},
        {0, -9. / 2, -9, 61. / 8, 61. / 4, -29. / 8, -29. / 4, 1. / 2, 1, 0}, {0, 9. / 2, -9, -61. / 8, 61. / 4, 29. / 8, -29. / 4, -1. / 2, 1, 0},
        {0, -3. / 2, -1, 63. / 8, 21. / 4, -63. / 8, -21. / 4, 3. / 2, 1, 0}, {0, 3. / 2, -1, -63. / 8, 21. / 4, 63. / 8, -21. / 4, -3. / 2, 1, 0},
        {0, 9. / 4, 0, -205. / 16, 0, 273. / 16, 0, -15. / 2, 0, 1}};
    static constexpr float AT[4][10] = {{1, 1, 1, 1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 1. / 2, -1. / 2, 3. / 2, -3. / 2, 0},
                                        {0, 1, 1, 4, 4, 1. / 4, 1. / 4, 9. / 4, 9. / 4, 0}, {0, 1, -1, 8, -8, 1. / 8, -1. / 8, 27. / 8, -27. / 8, 1}};
    static constexpr double G[10][7] = {
        {4. / 9, 0, 0, 0, 0, 0, 0},
        {8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45, 8. / 45}, {8. / 45, -8. / 45, 8. / 45, -8. / 45, 8. / 45, -8. / 45, 8. / 45},
        {2. / 315, 4. / 315, 8. / 315, 16. / 315, 32. / 315, 64. / 315, 128. / 315}, {2. / 315, -4. / 315, 8. / 315, -16. / 315, 32. / 315, -64. / 315, 128. / 315},
        {-16. / 45, -8. / 45, -4. / 45, -2. / 45, -1. / 45, -1. / 90, -1. / 180}, {-16. / 45, 8. / 45, -4. / 45, 2. / 45, -1. / 45, 1. / 90, -1. / 180},
        {-16. / 315, -8. / 105, -4. / 35, -6. / 35, -9. / 35, -27. / 70, -81. / 140}, {-16. / 315, 8. / 105, -4. / 35, 6. / 35, -9. / 35, 27. / 70, -81. / 140},
        {0, 0, 0, 0, 0, 0, 1}};
};

// F(4,4), interpolation points 0, +-1, +-1/2, 2, inf (column phases of the 7-tap stride-2 rows, 4 outputs per tile)
template <> struct RowWino<44> {
    static constexpr float BT[7][7] = {{-1. / 2, 1. / 4, 5. / 2, -5. / 4, -2, 1, 0}, {0, 1. / 2, 1. / 4, -9. / 4, -1, 1, 0}, {0, -1. / 2, 3. / 4, 7. / 4, -3, 1, 0},
                                       {0, 1, 3. / 2, -2, -3. / 2, 1, 0}, {0, -1, 5. / 2, 0, -5. / 2, 1, 0}, {0, 1. / 4, 0, -5. / 4, 0, 1, 0},
                                       {0, -1. / 2, 1. / 4, 5. / 2, -5. / 4, -2, 1}};
    static constexpr float AT[4][7] = {{1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 1. / 2, -1. / 2, 2, 0}, {0, 1, 1, 1. / 4, 1. / 4, 4, 0}, {0, 1, -1, 1. / 8, -1. / 8, 8, 1}};
    static constexpr double G[7][4] = {{-2, 0, 0, 0}, {-2. / 3, -2. / 3, -2. / 3, -2. / 3}, {-2. / 9, 2. / 9, -2. / 9, 2. / 9}, {16. / 9, 8. / 9, 4. / 9, 2. / 9},
                                       {16. / 15, -8. / 15, 4. / 15, -2. / 15}, {2. / 45, 4. / 45, 8. / 45, 16. / 45}, {0, 0, 0, 1}};
};
// F(4,2), interpolation points 0, 1, -1, 2, inf (column phases of the 3-tap stride-2 rows, 4 outputs per tile; B^T of F(2,4)):
// 5 multiplies per 4 outputs, phase and kernel row instead of 6 -- the 3x3 stride-2 layers leave the implicit-GEMM kernel
template <> struct RowWino<24> {
    static constexpr float AT[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 0}, {0, 1, -1, 8, 1}};
    static constexpr double G[5][2] = {{1. / 2, 0}, {-1. / 2, -1. / 2}, {-1. / 6, 1. / 6}, {1. / 6, 1. / 3}, {0, 1}};
};
// F(4,3), interpolation points 0, +-1, +-2, inf (column phases of the 5-tap stride-2 rows, 4 outputs per tile; B^T of F(2,5))
template <> struct RowWino<34> {
    static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    static constexpr double G[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
};

// Stride 2: out[y][X] = sum_ky sum_p sum_j w[ky][2j + 2 START + p + R/2] in[2y + ky - R/2][2 (X + j + START) + p]: the
// two column phases p of the input are stride-1 correlations with RT = ceil(R/2) taps (the shorter phase padded
// with a zero tap), both transformed with the same F(2,RT) and accumulated in the same frequency-domain registers;
// (RT+1) instead of R multiplies per two outputs and kernel row: 1.4x (R=7) / 1.25x (R=5) fewer MFMA flops.
template <int R, int S, int M> struct RowCfg {                            // M outputs per tile (2, or 4 for the 7-tap stride-1 rows)
    static constexpr int RT = S == 1 ? R : (R + 1) / 2;                 // taps seen by the transform
    static constexpr int NX = RT + M - 1;                                // frequency points
    static constexpr int ID = M == 2 ? RT : 10 * RT + M;                // table key
    static constexpr int NB = (NX * 16 <= 128 && ID != 24) ? 4 : 3;      // 16-tile blocks per wave (accumulators: NX * NB * 4 <= 128); F(4,2): 3, so that 8 images of 96x128 / 16 of 24x32 outputs make 512 workgroups
    static constexpr int START = -(RT / 2);                              // first window sample relative to the tile's first output (phase samples)
    static constexpr int NKP = R * S;                                    // (kernel row, column phase) pairs in the reduction
};

// output-transform coefficient AT[i][k]: the two-output tables store only row 1 (row 0 = 1 ... 1 0)
template <int ID, int M> __device__ __forceinline__ constexpr float at_coef(int i, int k) {
    if constexpr (M == 2) return i == 0 ? (k < (int)(sizeof(RowWino<ID>::AT1) / sizeof(float)) - 1 ? 1.f : 0.f) : RowWino<ID>::AT1[k];
    else return RowWino<ID>::AT[i][k];
}


// NW = 4: workgroup = 4 waves = 64 couts, two workgroups per CU.  NW = 8: 8 waves = 128 couts over the SAME transformed
// tiles, one workgroup per CU: waves 0-3 gather and transform exactly as before (one channel quad each), waves 4-7 only
// multiply -- the transform's VALU work, which comes out of the fp32 matrix pipe's time (profiles/r1_mfma_valu_probes.txt,
// 18 % of the kernel in profiles/r2_conv_pmc.txt), and the window gathers are shared by twice the MFMAs, and every SIMD
// still hosts one transforming and one multiplying wave.
template <int R, int S, int M, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void conv_rows_winograd_f32_kernel(const RowArgs a) {
    using CF = RowCfg<R, S, M>;
    using WM = RowWino<CF::ID>;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int NX = CF::NX, NB = CF::NB, TT = 16 * NB, NG = (NX + 1) / 2;
    constexpr int VBUF = NX * TT * 16;                                   // V[buf][xi][tile][16 k], slots XOR-swizzled with ((tile >> 1) & 3): conflict-free for the four non-contiguous 16-lane groups of ds_read_b128 and for the writes
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];           // 64 KB (R = 7) / 48 KB (R = 5) / less for stride 2
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tilesC = a.Cout / (16 * NW);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = tile % tilesC, t0 = (tile / tilesC) * TT;
    const int HW = a.H * a.W, HWo = a.Ho * a.Wo, THW = a.Ho * a.TW;
    const bool loader = NW == 4 || wave < 4;                             // wave-uniform

    // ---- loader: thread = (tile tl, channel quad qd = wave of the chunk)
    const int tl = lane, qd = wave & 3;
    const int tg = t0 + tl;
    const bool tvalid = (tl < TT) & (tg < a.T);                         // TT = 48: the last 16 lanes of a wave carry no tile
    int img, py, ptx;
    { const int tt = tvalid ? tg : 0; img = tt / THW; const int rem = tt - img * THW; py = rem / a.TW; ptx = rem - py * a.TW; }
    // per-thread invariants of the window loads: x offset and x validity of the NX samples (per column phase for
    // stride 2); per chunk the (kernel row, phase, channel group) triple is wave-uniform, only the row validity is per
    // lane: two vector instructions per load
    unsigned xoff[S][NX]; bool xok[S][NX];
#pragma unroll
    for (int p = 0; p < S; ++p)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int ix = S == 1 ? M * ptx - R / 2 + j : 2 * (M * ptx + j + CF::START) + p;
            xok[p][j] = tvalid & ((unsigned)ix < (unsigned)a.W); xoff[p][j] = (unsigned)ix * 16u;
        }
    float4 d[NX];
    // reduction order: channel group outer, (kernel row [, phase]) inner -- the R kernel rows of one group are consecutive
    // quads, so the R (2R) reads of an input row come within a few chunks of each other and hit L2 (kernel-row-major
    // order streamed the whole input once per kernel row: 7.6x the input bytes fetched for conv1.3)
    int gq = qd / CF::NKP, kp = qd - gq * CF::NKP;                       // channel group and (kernel row [, phase]) of the next chunk to gather
    __amdgpu_buffer_rsrc_t grsrc; unsigned gbase; bool gok; int gph;
    auto gather_begin = [&]() {
        const int g = gq;
        const bool s1 = g < a.Gsplit;
        grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 ? a.in : a.in2), 0, s1 ? a.in_bytes : a.in2_bytes, 0x00020000);
        const int ky = S == 1 ? kp : kp >> 1;
        gph = S == 1 ? 0 : kp & 1;
        const int iy = S * py + ky - R / 2;
        gbase = ((s1 ? (unsigned)((img * a.Gin_tot + a.gin0 + g) * HW) : (unsigned)((img * a.Gin2_tot + a.gin2_0 + g - a.Gsplit) * HW)) + (unsigned)(iy * a.W)) * 16u;
        gok = (g < a.Gin) & ((unsigned)iy < (unsigned)a.H);
        kp += 4;                                                         // advance to the following chunk's quad (NKP >= 5: at most one wrap)
        { const bool wrap = kp >= CF::NKP; kp -= wrap ? CF::NKP : 0; gq += wrap; }
    };
    auto gather_load = [&](int j) {
        const bool ok = gok & (S == 1 ? xok[0][j] : (gph ? xok[S - 1][j] : xok[0][j]));
        const unsigned xo = S == 1 ? xoff[0][j] : (gph ? xoff[S - 1][j] : xoff[0][j]);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(grsrc, ok ? gbase + xo : 0xFFFFFFFFu, 0, 0);
        d[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // V = BT d in NG = (R+1)/2 groups of two frequency points (even/odd factorisation of the +-p point pairs):
    // group 0 = (V0, V_R) [points 0, inf], group i = (V_{2i-1}, V_{2i}) [points +p_i, -p_i].
    const int wofs = tl * 16 + (qd ^ ((tl >> 1) & 3)) * 4;
    auto transform_group = [&](int grp, float* Vdst) {
        const f4p* x = reinterpret_cast<const f4p*>(&d[0]);
        f4p va, vb; int ka, kb = -1;
        if constexpr (CF::ID == 74) {                                    // 10 points: rows 0 / 9, then four +-p pairs (even / odd parts from the table)
            f4p e = f4_mul(0.f, x[0]), o = e; bool fe = true, fo = true;
            const int k = grp == 0 ? 0 : 2 * grp - 1;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float cf = WM::BT[k][j];
                if (cf == 0.f) continue;
                if (grp == 0 || (j & 1) == 0) { e = fe ? f4_mul(cf, x[j]) : f4_fma(cf, x[j], e); fe = false; }
                else { o = fo ? f4_mul(cf, x[j]) : f4_fma(cf, x[j], o); fo = false; }
            }
            if (grp == 0) {
                va = e; ka = 0; kb = 9; bool f9 = true;
#pragma unroll
                for (int j = 0; j < 10; ++j) { const float cf = WM::BT[9][j]; if (cf == 0.f) continue; vb = f9 ? f4_mul(cf, x[j]) : f4_fma(cf, x[j], vb); f9 = false; }
            } else { va = f4_add(e, o); vb = f4_sub(e, o); ka = k; kb = k + 1; }
        } else if constexpr (CF::ID == 44) {                             // 7 points without +-p structure: rows straight from the table, two per step
            auto row = [&](int k) { f4p v = f4_mul(0.f, x[0]); bool f = true;
#pragma unroll
                for (int j = 0; j < 7; ++j) { const float cf = WM::BT[k][j]; if (cf == 0.f) continue; v = f ? f4_mul(cf, x[j]) : f4_fma(cf, x[j], v); f = false; }
                return v; };
            if (grp == 0) { va = row(0); vb = row(6); ka = 0; kb = 6; }
            else if (grp < 3) { va = row(2 * grp - 1); vb = row(2 * grp); ka = 2 * grp - 1; kb = 2 * grp; }
            else { va = row(5); ka = 5; }
        } else if constexpr (CF::RT == 7) {
            if (grp == 0) { va = f4_fma(5.25f, f4_sub(x[2], x[4]), f4_sub(x[6], x[0])); vb = f4_fma(5.25f, f4_sub(x[3], x[5]), f4_sub(x[7], x[1])); ka = 0; kb = 7; }
            else {
                f4p e, o;
                if (grp == 1) { e = f4_fma(-4.25f, x[4], f4_add(x[2], x[6])); o = f4_fma(-4.25f, x[3], f4_add(x[1], x[5])); }
                else if (grp == 2) { e = f4_fma(0.25f, x[2], f4_fma(-1.25f, x[4], x[6])); o = f4_fma(0.5f, x[1], f4_fma(-2.5f, x[3], f4_mul(2.f, x[5]))); }
                else { e = f4_fma(4.f, x[2], f4_fma(-5.f, x[4], x[6])); o = f4_fma(2.f, x[1], f4_fma(-2.5f, x[3], f4_mul(0.5f, x[5]))); }
                va = f4_add(e, o); vb = f4_sub(e, o); ka = 2 * grp - 1; kb = 2 * grp;
            }
        } else if constexpr (CF::ID == 5 || CF::ID == 34) {              // six points 0, +-1, +-2, inf: F(2,5) and F(4,3)
            if (grp == 0) { va = f4_fma(4.f, x[0], f4_fma(-5.f, x[2], x[4])); vb = f4_fma(4.f, x[1], f4_fma(-5.f, x[3], x[5])); ka = 0; kb = 5; }
            else {
                f4p e, o;
                if (grp == 1) { e = f4_fma(-4.f, x[2], x[4]); o = f4_fma(-4.f, x[1], x[3]); }
                else { e = f4_sub(x[4], x[2]); o = f4_mul(2.f, f4_sub(x[3], x[1])); }
                va = f4_add(e, o); vb = f4_sub(e, o); ka = 2 * grp - 1; kb = 2 * grp;
            }
        } else if constexpr (CF::RT == 4 || CF::ID == 24) {              // five points 0, 1, -1, 2, inf (F(2,4) and F(4,2)): BT = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
            const f4p v3 = f4_sub(x[3], x[1]);
            if (grp == 0) { va = f4_fma(2.f, f4_sub(x[0], x[2]), v3); vb = f4_fma(-2.f, v3, f4_sub(x[4], x[2])); ka = 0; kb = 4; }
            else if (grp == 1) { const f4p q = f4_sub(x[3], x[2]); va = f4_fma(-2.f, x[1], q); vb = f4_fma(2.f, f4_sub(x[1], x[2]), q); ka = 1; kb = 2; }
            else { va = v3; ka = 3; }
        } else {                                                         // RT == 3: BT = [-1 0 1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]
            if (grp == 0) { va = f4_sub(x[2], x[0]); vb = f4_sub(x[3], x[1]); ka = 0; kb = 3; }
            else { va = f4_add(x[1], x[2]); vb = f4_sub(x[2], x[1]); ka = 1; kb = 2; }
        }
        if (TT == 64 || tl < TT) {
            *reinterpret_cast<f4p*>(Vdst + (size_t)ka * TT * 16 + wofs) = va;
            if (kb >= 0) *reinterpret_cast<f4p*>(Vdst + (size_t)kb * TT * 16 + wofs) = vb;
        }
    };

    f32x4 acc[NX][NB];
#pragma unroll
    for (int x = 0; x < NX; ++x)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weights in MFMA operand order: [chunk][cout/16][xi][lane][4], lane (i = l&15, kg = l>>4) = U[xi][co 16cb+i][k 16chunk+4kg+e]
    const int cb16 = cblk * NW + wave, ncb16 = a.Cout / 16;
    const float4* ubase = reinterpret_cast<const float4*>(a.u) + lane + (size_t)cb16 * NX * 64;
    const size_t ustride = (size_t)ncb16 * NX * 64;                      // float4 per chunk
    const int rtile = lane & 15, kg = lane >> 4;
    const int voff = rtile * 16 + (kg ^ ((rtile >> 1) & 3)) * 4;         // tile block tb adds tb*16 rows (same swizzle: 16 % 16 == 0)

    constexpr int WD = NX > 8 ? NX / 2 : NX;                             // weight fragments in flight: a whole chunk, half of one for 10 points (registers)
    float4 af[WD];
#pragma unroll
    for (int x = 0; x < WD; ++x) af[x] = ubase[(size_t)x * 64];
    if (loader) {
        gather_begin();                                                  // chunk 0
#pragma unroll
        for (int j = 0; j < NX; ++j) gather_load(j);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) transform_group(grp, V);
        gather_begin();                                                  // chunk 1
#pragma unroll
        for (int j = 0; j < NX; ++j) gather_load(j);
    }
    lds_barrier();
    for (int c = 0; c < a.nchunks; ++c) {
        const float* Vc = V + (c & 1) * VBUF;
        float* Vn = V + ((c + 1) & 1) * VBUF;
        const float4* uc = ubase + (size_t)c * ustride;
        const float4* un = ubase + (size_t)(c + 1 < a.nchunks ? c + 1 : c) * ustride;
        float4 bf[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bf[b] = *reinterpret_cast<const float4*>(Vc + b * 16 * 16 + voff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < NX; ++x) {                                   // one frequency point per step: 16 MFMAs
            const float4 aw = af[x % WD];
            float4 bw[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) bw[b] = bf[b];
            if (x + 1 < NX) {
#pragma unroll
                for (int b = 0; b < NB; ++b) bf[b] = *reinterpret_cast<const float4*>(Vc + (size_t)(x + 1) * TT * 16 + b * 16 * 16 + voff);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.x, bw[b].x, acc[x][b], 0, 0, 0);
            af[x % WD] = x + WD < NX ? uc[(size_t)(x + WD) * 64] : un[(size_t)(x + WD - NX) * 64];   // WD steps ahead (wraps into the next chunk)
            // between the MFMAs: one transform group of chunk c+1 per step, then (registers free) the windows of chunk
            // c+2, two loads per step (past the last chunk all out of range = 0, written to the idle buffer)
            if (loader) {
                if (x < NG) transform_group(x, Vn);
                if (x >= NG) {                        // LPS loads per step, spread over the steps after the transform
                    constexpr int LPS = (NX + (NX - NG) - 1) / (NX - NG);
                    if (x == NG) gather_begin();
#pragma unroll
                    for (int l = LPS * (x - NG); l < LPS * (x - NG + 1) && l < NX; ++l) gather_load(l);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.y, bw[b].y, acc[x][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.z, bw[b].z, acc[x][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.w, bw[b].w, acc[x][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                                   // V[c+1] complete, V[c] free for chunk c+2
    }

    // ---- epilogue: y_i = sum_k AT[i][k] M_k, i < M; acc row = cout 4*(lane>>4)+r (one c4 group), col = tile lane&15 (+16 tb)
    const int co = cblk * 16 * NW + wave * 16 + 4 * kg;
    const float4 bias = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
        const int to = t0 + tb * 16 + rtile;
        if (to >= a.T) continue;
        const int oimg = to / THW, orem = to - oimg * THW, oy = orem / a.TW, otx = orem - oy * a.TW;
        float* op = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), HWo, oy * a.Wo + M * otx);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sacc = bb[r];
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const float cf = at_coef<CF::ID, M>(i, k);
                    if (cf == 1.f) sacc += acc[k][tb][r];
                    else if (cf == -1.f) sacc -= acc[k][tb][r];
                    else if (cf != 0.f) sacc = fmaf(cf, acc[k][tb][r], sacc);
                }
                y[r] = a.relu ? fmaxf(sacc, 0.f) : sacc;
            }
            if (M * otx + i < a.Wo) *reinterpret_cast<float4*>(op + 4 * i) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

// U[xi][co][k = 4*(g*NKP + kp) + e] = (sum_j G[xi][j] w'[co][ci][kp][j]) * BN scale, in MFMA A-operand order
// [chunk][cout/16][xi][lane][4]:  co = cb*16 + (lane&15), k = chunk*16 + 4*(lane>>4) + e.  Stride 1: kp = kernel row,
// w' = the row's taps; stride 2: kp = (kernel row, column phase), w' = that phase's taps (zero where it has none).
template <int R, int S, int M>
__global__ void pack_rows_winograd_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                          float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up) {
    using CF = RowCfg<R, S, M>;
    constexpr int NX = CF::NX;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb16 = Cout / 16;
    const long long total = (long long)nchunks * ncb16 * NX * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int xi = (int)(r % NX); r /= NX;
    const int cb = (int)(r % ncb16), chunk = (int)(r / ncb16);
    const int co = cb * 16 + (lane & 15);
    const int q = chunk * 4 + (lane >> 4);
    const int gi = q / CF::NKP, kp = q - gi * CF::NKP, cp = 4 * gi + e;
    float v = 0.f;
    if (cp < Cin) {
        const int ci = (cp + rot) % Cin;
        const int ky = S == 1 ? kp : kp >> 1, ph = S == 1 ? 0 : kp & 1;
        const float* g = w + (((size_t)co * Cin + ci) * R + ky) * R;
        double s = 0;
        for (int j = 0; j < CF::RT; ++j) {
            const int kx = S == 1 ? j : 2 * j + 2 * CF::START + ph + R / 2;
            if (kx >= 0 && kx < R) s += RowWino<CF::ID>::G[xi][j] * (double)g[kx];
        }
        if (gamma) s *= (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)s;
    }
    up[idx] = v;
}

static int g_rows_wide = 1;                                              // 0: 64-cout workgroups only, 1: 128-cout workgroups where they fill whole rounds, 2: wherever Cout % 128 == 0
extern "C" int cnm_tune_rows_wide(int on) { const int old = g_rows_wide; if (on >= 0 && on <= 2) g_rows_wide = on; return old; }

static int rows_chunks(int Cin, int ksize, int stride) { return (ksize * stride * ((Cin + 3) / 4) + 3) / 4; }
static bool rows_ksize_ok(int ksize, int stride, int tile) { return ksize == 5 || ksize == 7 || (ksize == 3 && stride == 2 && tile == 4); }   // 3x3: stride 2 only (stride 1 has the 2-D kernels)
static bool rows_tile_ok(int ksize, int stride, int tile) { return tile == 2 || (tile == 4 && !(ksize == 5 && stride == 1)); }   // 4 outputs per tile: F(4,7), and the stride-2 phases F(4,4) / F(4,3)
static int rows_points(int ksize, int stride, int tile) { return (stride == 1 ? ksize : (ksize + 1) / 2) + tile - 1; }

extern "C" size_t cnm_packed_winograd_rows_floats(int Cout, int Cin, int ksize, int stride, int tile) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || !rows_ksize_ok(ksize, stride, tile) || (stride != 1 && stride != 2) || !rows_tile_ok(ksize, stride, tile)) return 0;
    return (size_t)rows_chunks(Cin, ksize, stride) * rows_points(ksize, stride, tile) * Cout * 16;
}

extern "C" int cnm_pack_winograd_rows_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                             int Cout, int Cin, int ksize, int stride, int tile, int rot, float* u_packed, void* stream) {
    CNM_REQUIRE(w_oihw && u_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(rows_ksize_ok(ksize, stride, tile) && (stride == 1 || stride == 2) && rows_tile_ok(ksize, stride, tile) && !bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = rows_chunks(Cin, ksize, stride);
    const long long total = (long long)nchunks * rows_points(ksize, stride, tile) * Cout * 16;
    const unsigned nb = (unsigned)cnm_ceil_div_ll(total, 256);
    hipStream_t st = cnm_stream(stream);
#define CNM_PACK_ROWS(R, S, M) pack_rows_winograd_kernel<R, S, M><<<nb, 256, 0, st>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed)
    if (ksize == 3) CNM_PACK_ROWS(3, 2, 4);
    else if (ksize == 5 && stride == 1) CNM_PACK_ROWS(5, 1, 2); else if (ksize == 5 && tile == 4) CNM_PACK_ROWS(5, 2, 4); else if (ksize == 5) CNM_PACK_ROWS(5, 2, 2);
    else if (stride == 1 && tile == 4) CNM_PACK_ROWS(7, 1, 4); else if (stride == 1) CNM_PACK_ROWS(7, 1, 2); else if (tile == 4) CNM_PACK_ROWS(7, 2, 4); else CNM_PACK_ROWS(7, 2, 2);
#undef CNM_PACK_ROWS
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

static int conv_rows(const float* in_a, int Ga_total, int ga0, int Ga,
                     const float* in_b, int Gb_total, int gb0, int Gb,
                     float* out, int Gout_total, int gout0, int Cout,
                     const float* u_packed, const float* b_packed,
                     int N, int H, int W, int ksize, int stride, int tile, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    CNM_REQUIRE(in_a && out && u_packed && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0 && rows_ksize_ok(ksize, stride, tile), CNM_ERR_BAD_ARG);
    CNM_REQUIRE((stride == 1 || stride == 2) && rows_tile_ok(ksize, stride, tile), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    RowArgs a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    const int pad = ksize / 2;
    const int m = tile;
    a.N = N; a.H = H; a.W = W; a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1; a.TW = (a.Wo + m - 1) / m;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (ksize * stride * a.Gin + 3) / 4; a.T = N * a.Ho * a.TW; a.relu = relu;
    const int ntb = cnm_ceil_div(a.T, ((ksize == 7 && stride == 1 && m == 4) || ksize == 3) ? 48 : 64);
    const int nblocks = (Cout / 64) * ntb;
    hipStream_t st = cnm_stream(stream);
    if (ksize == 7 && stride == 1 && m == 4) {                           // LDS-staged persistent kernel (conv_rows_staged.hip)
        const int rc = cnm_rows7s_try_launch(a, sync_ws, sync_floats, st);
        if (rc <= 0) return rc;
    }
    if (g_rows_wide && m == 4 && Cout % 128 == 0) {
        // 128 couts per workgroup, one workgroup per CU (rounds of 256): taken unless the last round would be mostly empty
        const long long nb8 = (long long)(Cout / 128) * ntb;
        const double waste = (double)((nb8 + 255) / 256) * 256.0 / (double)nb8;
        // measured (tools/rows_wide_probe.py, 16 pairs at 192x256): 7x7 stride 2 1.07x; 7x7 stride 1 0.99x, 3x3 stride 2 0.91-0.98x, 5x5
        // stride 2 0.81x (1.5 rounds) -- with one workgroup per CU nothing overlaps a workgroup's prologue, barriers and output
        // transform any more, which costs what the shared transform saves; mode 1 therefore takes the 7x7 stride-2 layer only
        if (g_rows_wide == 2 || (waste <= 1.2 && ksize == 7 && stride == 2)) {
            if (ksize == 3) conv_rows_winograd_f32_kernel<3, 2, 4, 8><<<(unsigned)nb8, 512, 0, st>>>(a);
            else if (ksize == 5) conv_rows_winograd_f32_kernel<5, 2, 4, 8><<<(unsigned)nb8, 512, 0, st>>>(a);
            else if (stride == 1) conv_rows_winograd_f32_kernel<7, 1, 4, 8><<<(unsigned)nb8, 512, 0, st>>>(a);
            else conv_rows_winograd_f32_kernel<7, 2, 4, 8><<<(unsigned)nb8, 512, 0, st>>>(a);
            CNM_LAUNCH_CHECK();
            return CNM_OK;
        }
    }
    if (ksize == 3) conv_rows_winograd_f32_kernel<3, 2, 4><<<nblocks, 256, 0, st>>>(a);
    else if (ksize == 5 && stride == 1) conv_rows_winograd_f32_kernel<5, 1, 2><<<nblocks, 256, 0, st>>>(a);
    else if (ksize == 5 && tile == 4) conv_rows_winograd_f32_kernel<5, 2, 4><<<nblocks, 256, 0, st>>>(a);
    else if (ksize == 5) conv_rows_winograd_f32_kernel<5, 2, 2><<<nblocks, 256, 0, st>>>(a);
    else if (stride == 1 && tile == 4) conv_rows_winograd_f32_kernel<7, 1, 4><<<nblocks, 256, 0, st>>>(a);
    else if (stride == 1) conv_rows_winograd_f32_kernel<7, 1, 2><<<nblocks, 256, 0, st>>>(a);
    else if (tile == 4) conv_rows_winograd_f32_kernel<7, 2, 4><<<nblocks, 256, 0, st>>>(a);
    else conv_rows_winograd_f32_kernel<7, 2, 2><<<nblocks, 256, 0, st>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv_rows_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                             const float* in_b, int Gb_total, int gb0, int Gb,
                                             float* out, int Gout_total, int gout0, int Cout,
                                             const float* u_packed, const float* b_packed,
                                             int N, int H, int W, int ksize, int stride, int tile, int relu, void* stream) {
    return conv_rows(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, ksize, stride, tile, relu, nullptr, 0, stream);
}

// The same with a sync workspace (cnm_wino36_sync_floats() floats, zero before the first use, one per stream): the staged
// 7x7 stride-1 kernel then gives every CU an equal share of the reduction phases.  Other shapes ignore it.
extern "C" int cnm_conv_rows_winograd_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                                  const float* in_b, int Gb_total, int gb0, int Gb,
                                                  float* out, int Gout_total, int gout0, int Cout,
                                                  const float* u_packed, const float* b_packed,
                                                  int N, int H, int W, int ksize, int stride, int tile, int relu,
                                                  float* sync_ws, size_t sync_floats, void* stream) {
    return conv_rows(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, ksize, stride, tile, relu, sync_ws, sync_floats, stream);
}
