// Winograd F(4x4,3x3) convolution for the large 3x3 stride-1 layers, fp32 MFMA.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 4x4 output tile, 6x6 input window d, 3x3 filter g, 36 frequency
//   points (interpolation points 0, +-1, +-2, inf): 36 multiplies per 16 outputs instead of 144 -> 4x fewer MFMA
//   flops than the direct convolution, 1.78x fewer than F(2x2,3x3) (conv_winograd.hip).  fp32 data and accumulation;
//   the larger transform constants (|B^T| <= 5, |A^T| <= 8) cost accuracy: per layer 1-3e-5 on O(1) outputs (F(2x2):
//   1-3e-6), end to end 6e-5 on the refined inverse depth against an fp64 evaluation (direct fp32: 3e-5) -- inside
//   the 1e-3 parity bar with more than a decade to spare (tools/wino4_e2e_error.py, tests/test_gpu_parity.py).
//
// Same machine as conv_winograd.hip (v_mfma_f32_16x16x4_f32, two workgroups per CU):
//   workgroup = 4 waves = 64 couts x 16 tiles (256 output pixels); wave = 16 couts x 16 tiles x 36 points = 144
//   accumulator registers; per 16-channel chunk
//     - wave q gathers channel quad q: lane = (tile, channel) loads its 6x6 window with 36 buffer_load_dword
//       (offset = saturating add of a per-row and a per-column term, both loop-invariant and 0xFFFFFFFF outside the
//       image; everything that changes per chunk is scalar), transforms it in place in registers (12 one-dimensional
//       6-point transforms of 12 operations) and writes the 36 points to LDS V[xi][tile][ci] (double buffered, 72 KB);
//     - per point one ds_read_b128 of V and one 16-byte weight fragment straight from L2 feed 4 MFMAs;
//   epilogue: the 36 values of an output tile sit in ONE lane -> A^T M A in registers (100 operations per cout),
//   bias, ReLU, sixteen float4 stores.
// The executors use it for layers with enough tiles to fill the chip (conv_winograd.hip otherwise).
#include "cnm_common.h"

#ifndef WINO4_LPS
#define WINO4_LPS 4      // window loads per double step (divides 36; at most 12 double steps are available)
#endif
#ifndef WINO4_WD
#define WINO4_WD 4        // weight fragments in flight per wave (even, divides 36; 6 and more spill)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier4() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned sat_add(unsigned a, unsigned b) {    // 0xFFFFFFFF (= out of range for the buffer load) absorbs
    unsigned r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#include "wino4_args.h"

// B^T (6 points): rows [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1], in place
#define WINO4_BT(x0, x1, x2, x3, x4, x5) do {                                                                   \
        const float t0 = fmaf(4.f, x0, fmaf(-5.f, x2, x4)), t5 = fmaf(4.f, x1, fmaf(-5.f, x3, x5));             \
        const float e1 = fmaf(-4.f, x2, x4), o1 = fmaf(-4.f, x1, x3);                                           \
        const float e2 = x4 - x2, o2 = x3 - x1;                                                                 \
        x0 = t0; x1 = e1 + o1; x2 = e1 - o1; x3 = fmaf(2.f, o2, e2); x4 = fmaf(-2.f, o2, e2); x5 = t5;          \
    } while (0)
// A^T of F(2,5) (2 outputs from the same 6 points): rows [1 1 1 1 1 0; 0 1 -1 2 -2 1]
#define WINO2_AT(y0, y1, m0, m1, m2, m3, m4, m5) do {                                                           \
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;                                     \
        y0 = m0 + s1 + s2; y1 = fmaf(2.f, d2, d1) + m5;                                                         \
    } while (0)
// A^T of F(4,3) (4 outputs from 6 points): rows [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
#define WINO4_AT(y0, y1, y2, y3, m0, m1, m2, m3, m4, m5) do {                                                   \
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;                                     \
        y0 = m0 + s1 + s2; y1 = fmaf(2.f, d2, d1); y2 = fmaf(4.f, s2, s1); y3 = fmaf(8.f, d2, d1) + m5;         \
    } while (0)

// UPS: the convolution runs on the bilinear 2x upsampling of its input without materialising it.  Upsample-then-3x3 is,
// per output phase (row parity a, column parity b), a 3x3 convolution of the LOW-resolution input with a composed filter
// (ops.compose_upsample_filters); the four phases are four blocks of "virtual" output channels (Cout = 4 x real, phase
// major) written pixel-shuffled: virtual channel (2a+b) Cr + c at low-resolution (y, x) -> channel c at (2y+a, 2x+b).
// The composition equals the convolution of the upsampled image with REPLICATE padding (window samples outside the
// low-resolution image are clamped, not zeroed); the one-pixel output ring where zero padding differs is corrected by
// conv_upsampled_ring_kernel, so ring pixels are stored before bias / ReLU here.
template <int M, int R, bool UPS = false>                                // M x M outputs per tile, R x R filter, M + R - 1 == 6
__global__ __launch_bounds__(256, 2) void conv_winograd36_f32_kernel(const Wino4Args a) {
    static_assert(M + R - 1 == 6, "36-point kernel");
    static_assert(!UPS || (M == 4 && R == 3), "fused upsampling: F(4x4,3x3) only");
    constexpr int TT = 16, NXI = 36, VBUF = NXI * TT * 16;               // V[buf][xi][tile][16 ci], slots XOR-swizzled with ((tile >> 1) & 3): conflict-free for the four non-contiguous 16-lane groups of ds_read_b128 and for the writes
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];           // 72 KB
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tilesC = a.Cout / 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = tile % tilesC, t0 = (tile / tilesC) * TT;
    const int HW = a.H * a.W, THW = a.TH * a.TW;

    // ---- loader: wave = channel quad of the chunk, lane = (tile tl, channel cc of the quad)
    const int tl = lane >> 2, cc = lane & 3, qd = wave;
    unsigned roff[6], coff[6];                                           // loop-invariant byte offsets: window rows (image, row, channel) and columns
    unsigned imgdelta;                                                   // image term of view 2 minus view 1
    {
        const int tg = t0 + tl;
        const bool tvalid = tg < a.T;
        const int tt = tvalid ? tg : 0; const int img = tt / THW; const int rem = tt - img * THW; const int ty = rem / a.TW;
        const int py = M * ty - R / 2, px = M * (rem - ty * a.TW) - R / 2;
        const unsigned imgterm = (unsigned)img * (unsigned)a.Gin_tot * (unsigned)HW * 16u;
        imgdelta = (unsigned)img * (unsigned)a.Gin2_tot * (unsigned)HW * 16u - imgterm;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int iy = py + i, ix = px + i;
            if constexpr (UPS) { if (!a.ups_zero) { iy = min(max(iy, 0), a.H - 1); ix = min(max(ix, 0), a.W - 1); } }
            roff[i] = (tvalid & ((unsigned)iy < (unsigned)a.H)) ? imgterm + (unsigned)(iy * a.W) * 16u + cc * 4u : 0xFFFFFFFFu;
            coff[i] = (unsigned)ix < (unsigned)a.W ? (unsigned)ix * 16u : 0xFFFFFFFFu;
        }
    }
    float d[36];
    bool view2 = false;                                                  // wave-uniform: roff[] already rebased to the second view
    __amdgpu_buffer_rsrc_t grsrc; unsigned gsoff;
    auto gather_begin = [&](int chunk) {
        const int g = chunk * 4 + qd;                                   // channel group of the (possibly concatenated) input
        const bool s1 = g < a.Gsplit;
        if (!s1 && !view2) {                                            // once per wave, when its quad crosses into the second view
            view2 = true;
#pragma unroll
            for (int i = 0; i < 6; ++i) roff[i] = roff[i] == 0xFFFFFFFFu ? roff[i] : roff[i] + imgdelta;
        }
        const unsigned bytes = g < a.Gin ? (s1 ? a.in_bytes : a.in2_bytes) : 0u;
        grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 ? a.in : a.in2), 0, bytes, 0x00020000);
        gsoff = (unsigned)(s1 ? a.gin0 + g : a.gin2_0 + g - a.Gsplit) * (unsigned)HW * 16u;
    };
    auto gather_load = [&](int ij) {
        d[ij] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grsrc, sat_add(roff[ij / 6], coff[ij % 6]), gsoff, 0));
    };
    const int wofs = tl * 16 + (qd ^ ((tl >> 1) & 3)) * 4 + cc;
    auto column_pass = [&](int j) { WINO4_BT(d[0 * 6 + j], d[1 * 6 + j], d[2 * 6 + j], d[3 * 6 + j], d[4 * 6 + j], d[5 * 6 + j]); };
    auto row_pass = [&](int i, float* Vdst) {
        WINO4_BT(d[i * 6 + 0], d[i * 6 + 1], d[i * 6 + 2], d[i * 6 + 3], d[i * 6 + 4], d[i * 6 + 5]);
#pragma unroll
        for (int j = 0; j < 6; ++j) Vdst[(size_t)(i * 6 + j) * TT * 16 + wofs] = d[i * 6 + j];
    };

    f32x4 acc[NXI];
#pragma unroll
    for (int x = 0; x < NXI; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weights in MFMA operand order: [chunk][cout/16][xi][lane][4], lane (i = l&15, kg = l>>4) = U[xi][co 16cb+i][ci 16chunk+4kg+e]
    const int cb16 = cblk * 4 + wave, ncb16 = a.Cout / 16;
    const float4* ubase = reinterpret_cast<const float4*>(a.u) + lane + (size_t)cb16 * NXI * 64;
    const size_t ustride = (size_t)ncb16 * NXI * 64;                     // float4 per chunk
    const int rtile = lane & 15, kg = lane >> 4;
    const int voff = rtile * 16 + (kg ^ ((rtile >> 1) & 3)) * 4;

    constexpr int WD = WINO4_WD;                                         // weight fragments in flight (steps of 4 MFMAs)
    float4 af[WD];
#pragma unroll
    for (int s = 0; s < WD; ++s) af[s] = ubase[(size_t)s * 64];
    gather_begin(0);
#pragma unroll
    for (int ij = 0; ij < 36; ++ij) gather_load(ij);
#pragma unroll
    for (int j = 0; j < 6; ++j) column_pass(j);
#pragma unroll
    for (int i = 0; i < 6; ++i) row_pass(i, V);
    gather_begin(1);
#pragma unroll
    for (int ij = 0; ij < 36; ++ij) gather_load(ij);
    lds_barrier4();
    for (int c = 0; c < a.nchunks; ++c) {
        const float* Vc = V + (c & 1) * VBUF;
        float* Vn = V + ((c + 1) & 1) * VBUF;
        const float4* uc = ubase + (size_t)c * ustride;
        const float4* un = ubase + (size_t)(c + 1 < a.nchunks ? c + 1 : c) * ustride;
        // double step xp = frequency points (2 xp, 2 xp + 1): their MFMAs alternate, so consecutive MFMAs never chain on
        // the same accumulator (one 16-tile block per wave: a single point per step would be one dependent chain)
        float4 bf0 = *reinterpret_cast<const float4*>(Vc + voff);
        float4 bf1 = *reinterpret_cast<const float4*>(Vc + TT * 16 + voff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xp = 0; xp < NXI / 2; ++xp) {
            const int x0 = 2 * xp, x1 = x0 + 1;
            const float4 a0 = af[x0 % WD], a1 = af[x1 % WD];
            const float4 b0 = bf0, b1 = bf1;
            if (xp + 1 < NXI / 2) {
                bf0 = *reinterpret_cast<const float4*>(Vc + (size_t)(x0 + 2) * TT * 16 + voff);
                bf1 = *reinterpret_cast<const float4*>(Vc + (size_t)(x1 + 2) * TT * 16 + voff);
            }
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[x1], 0, 0, 0);
            af[x0 % WD] = x0 + WD < NXI ? uc[(size_t)(x0 + WD) * 64] : un[(size_t)(x0 + WD - NXI) * 64];
            af[x1 % WD] = x1 + WD < NXI ? uc[(size_t)(x1 + WD) * 64] : un[(size_t)(x1 + WD - NXI) * 64];
            // between the MFMAs: the transform of chunk c+1 (double steps 0..5), then the window of chunk c+2 (WINO4_LPS
            // loads per double step; past the last chunk all out of range = 0, written to the idle buffer)
            if (xp < 3) { column_pass(2 * xp); column_pass(2 * xp + 1); }
            else if (xp < 6) { row_pass(2 * (xp - 3), Vn); row_pass(2 * (xp - 3) + 1, Vn); }
            else if (xp < 6 + 36 / WINO4_LPS) {      // row-major (column-major issue order measured 5 % slower: worse line locality)
                if (xp == 6) gather_begin(c + 2);
#pragma unroll
                for (int l = WINO4_LPS * (xp - 6); l < WINO4_LPS * (xp - 5); ++l) gather_load(l);
            }
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[x1], 0, 0, 0);
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc[x1], 0, 0, 0);
            acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[x0], 0, 0, 0);
            acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc[x1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier4();                                                  // V[c+1] complete, V[c] free for chunk c+2
    }

    // ---- epilogue: acc row = cout 4*(lane>>4)+r (one c4 group), col = tile lane&15
    const int to = t0 + rtile;
    if (to >= a.T) return;
    const int oimg = to / THW, orem = to - oimg * THW, oty = orem / a.TW, otx = orem - oty * a.TW;
    const int co = cblk * 64 + wave * 16 + 4 * kg;
    const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb[4] = {b.x, b.y, b.z, b.w};
    float y[M * M][4];                                                   // [pixel M*py+px][channel r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s[M][6];                                                   // A^T M: M rows x 6 columns
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if constexpr (M == 4) WINO4_AT(s[0][j], s[1][j], s[2][j], s[3][j], acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r], acc[5 * 6 + j][r]);
            else WINO2_AT(s[0][j], s[1][j], acc[0 * 6 + j][r], acc[1 * 6 + j][r], acc[2 * 6 + j][r], acc[3 * 6 + j][r], acc[4 * 6 + j][r], acc[5 * 6 + j][r]);
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if constexpr (M == 4) WINO4_AT(y[i * 4 + 0][r], y[i * 4 + 1][r], y[i * 4 + 2][r], y[i * 4 + 3][r], s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5]);
            else WINO2_AT(y[i * 2 + 0][r], y[i * 2 + 1][r], s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5]);
        }
    }
    if constexpr (UPS) {
        const int Cr = a.Cout >> 2, ph = co / Cr, cr = co - ph * Cr, pa = ph >> 1, pb = ph & 1;
        const int Ho = 2 * a.H, Wo = 2 * a.W;
        float* obase = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (cr >> 2), 4 * HW, 0);
#pragma unroll
        for (int p = 0; p < M * M; ++p) {
            const int ly = M * oty + p / M, lx = M * otx + p % M;
            const int oy = 2 * ly + pa, ox = 2 * lx + pb;
            const bool ring = (oy == 0) | (oy == Ho - 1) | (ox == 0) | (ox == Wo - 1);   // finished by the ring kernel (zero instead of replicate padding)
            float4 v = make_float4(y[p][0], y[p][1], y[p][2], y[p][3]);
            if (!ring || !a.ring) {
                v = make_float4(v.x + bb[0], v.y + bb[1], v.z + bb[2], v.w + bb[3]);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (ly < a.H && lx < a.W) *reinterpret_cast<float4*>(obase + (size_t)(oy * Wo + ox) * 4) = v;
        }
        return;
    }
    float* obase = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), HW, 0);
#pragma unroll
    for (int p = 0; p < M * M; ++p) {
        const int oy = M * oty + p / M, ox = M * otx + p % M;
        float4 v = make_float4(y[p][0] + bb[0], y[p][1] + bb[1], y[p][2] + bb[2], y[p][3] + bb[3]);
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (oy < a.H && ox < a.W) *reinterpret_cast<float4*>(obase + (size_t)(oy * a.W + ox) * 4) = v;   // ragged H / W: partial last tiles
    }
}

// U = G g G^T for F(4,3) / F(2,5) (same six points; with the folded BatchNorm scale), packed in MFMA A-operand order
// [chunk][cout/16][xi][lane][4].
// One workgroup per (chunk, 16-cout block): thread = (lane, e) of the fragment reads its R x R filter once and writes the 36
// frequency points as 36 coalesced 1 KB rows (one thread per output float spent its time in 64-bit index arithmetic and wrote
// at 1.6 TB/s: the training step re-packs ~90 filters, 2 ms).  Same fp64 expression per point as before: bit-identical output.
// S2 (stride-2 convolution on the four pixel phases of its input, conv_winograd4s.hip): w is the K x K stride-2 filter (dgrad
// carries K: 7 -> R = 4; 5 or 3 -> R = 3), chunk = 4 * (input chunk) + 2 py + px, and the R x R filter of phase (py, px) is
// tap (jy, jx) -> w[2 jy + py - o][2 jx + px - o] (o = 0 for 5x5, 1 for 7x7 and 3x3: the window starts (K - 1) / 2 input pixels
// = 1 or 2 phase pixels before the output pixel), zero where that leaves the filter (a 3x3 filter fills 1 / 2 / 2 / 4 of the 9 taps).
template <int R, bool S2 = false>
__device__ __forceinline__ void pack_winograd36_body(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                                     float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up, int dgrad, int block) {
    const int ncb16 = Cout / 16;
    const int cb = block % ncb16, chunk = block / ncb16;
    const int t = threadIdx.x, e = t & 3, lane = t >> 2;
    const int co = cb * 16 + (lane & 15), cp = (S2 ? chunk >> 2 : chunk) * 16 + 4 * (lane >> 4) + e;
    float* out = up + (size_t)block * 36 * 256 + t;
    if (cp >= Cin) {
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) out[xi * 256] = 0.f;
        return;
    }
    const int ci = (cp + rot) % Cin;
    // dgrad: the filter of the data gradient, w'[co][ci] = w[ci][co] rotated by 180 degrees, read straight from w [Cin][Cout][R][R]
    const float* g = dgrad ? w + ((size_t)ci * Cout + co) * R * R : w + ((size_t)co * Cin + ci) * R * R;
    double gv[R * R];
    if constexpr (S2) {
        const int K = dgrad, O = K == 5 ? 0 : 1;                         // S2: the `dgrad` argument carries the source filter size
        const int py = (chunk >> 1) & 1, px = chunk & 1;
        const float* g2 = w + ((size_t)co * Cin + ci) * K * K;
#pragma unroll
        for (int jy = 0; jy < R; ++jy)
#pragma unroll
            for (int jx = 0; jx < R; ++jx) {
                const int ky = 2 * jy + py - O, kx = 2 * jx + px - O;
                gv[jy * R + jx] = (ky >= 0 && ky < K && kx >= 0 && kx < K) ? (double)g2[ky * K + kx] : 0.0;
            }
    } else {
#pragma unroll
        for (int k = 0; k < R * R; ++k) gv[k] = (double)g[dgrad ? R * R - 1 - k : k];
    }
    const double scale = gamma ? (double)gamma[co] / sqrt((double)var[co] + (double)eps) : 1.0;
    constexpr double G3[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
    constexpr double G4[6][4] = {{1. / 4, 0, 0, 0}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6},
                                 {1. / 24, 1. / 12, 1. / 6, 1. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3}, {0, 0, 0, 1}};
    constexpr double G5[6][5] = {{1. / 4, 0, 0, 0, 0}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6},
                                 {1. / 24, 1. / 12, 1. / 6, 1. / 3, 2. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3, 2. / 3}, {0, 0, 0, 0, 1}};
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) {
        const int ai = xi / 6, bi = xi % 6;
        double s = 0;
#pragma unroll
        for (int p = 0; p < R; ++p)
#pragma unroll
            for (int q = 0; q < R; ++q) s += (R == 3 ? G3[ai][p] * G3[bi][q] : R == 4 ? G4[ai][p] * G4[bi][q] : G5[ai][p] * G5[bi][q]) * gv[p * R + q];
        if (gamma) s *= scale;
        out[xi * 256] = (float)s;
    }
}
template <int R, bool S2 = false>
__global__ __launch_bounds__(256) void pack_winograd36_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                                              float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up, int dgrad) {
    pack_winograd36_body<R, S2>(w, gamma, var, eps, Cout, Cin, rot, nchunks, up, dgrad, (int)blockIdx.x);
}

// [r6] MANY 3x3 filters in ONE launch (training: every filter is re-packed every step, forward and data-gradient form -- 63 launches of
// ~10 us each, 2 % of the step).  jobs[j] describes one cnm_pack_winograd4_bn_f32 (no BatchNorm fold) or cnm_pack_winograd4_dgrad_f32
// call; block b belongs to the job with the largest first_block <= b.  Same arithmetic per output float: bit-identical packed filters.
struct PackJob { const float* w; float* out; int Cout, Cin, rot, nchunks, dgrad, first_block; };
static_assert(sizeof(PackJob) == 40, "cnmnet_amd/autograd.py builds this table");
__global__ __launch_bounds__(256) void pack_winograd36_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                                          // wave-uniform binary search
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const PackJob j = jobs[lo];
    pack_winograd36_body<3, false>(j.w, nullptr, nullptr, 0.f, j.Cout, j.Cin, j.rot, j.nchunks, j.out, j.dgrad, (int)blockIdx.x - j.first_block);
}
extern "C" int cnm_pack_winograd4_batch_f32(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    CNM_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, CNM_ERR_BAD_ARG);
    pack_winograd36_batch_kernel<<<(unsigned)total_blocks, 256, 0, cnm_stream(stream)>>>(static_cast<const PackJob*>(jobs_dev), njobs);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" size_t cnm_packed_winograd4_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    return (size_t)nchunks * 36 * Cout * 16;
}

static int pack36(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps, int Cout, int Cin, int ksize, int rot,
                  float* u_packed, void* stream, int dgrad = 0, int s2 = 0) {
    CNM_REQUIRE(w_oihw && u_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16 * (s2 ? 4 : 1);
    const unsigned nb = (unsigned)(nchunks * (Cout / 16));
    if (s2 && ksize != 7) pack_winograd36_kernel<3, true><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed, ksize);
    else if (s2) pack_winograd36_kernel<4, true><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed, ksize);
    else if (ksize == 3) pack_winograd36_kernel<3><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed, dgrad);
    else if (ksize == 4) pack_winograd36_kernel<4><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed, dgrad);
    else pack_winograd36_kernel<5><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed, dgrad);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

static int conv36(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
                  float* out, int Gout_total, int gout0, int Cout, const float* u_packed, const float* b_packed,
                  int N, int H, int W, int ksize, int relu, void* stream, int ups = 0, int ring = 0, float* sync_ws = nullptr, size_t sync_floats = 0, int s2 = 0, int ups_zero = 0) {
    CNM_REQUIRE(in_a && out && u_packed && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!s2 || (!ups && (ksize == 3 || ksize == 5 || ksize == 7) && H % 2 == 0 && W % 2 == 0 && sync_ws), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!ups || ksize == 3 || (ksize == 4 && ups_zero), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    Wino4Args a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    const int m = s2 ? (ksize == 7 ? 3 : 4) : ksize == 3 ? 4 : ksize == 4 ? 3 : 2;   // outputs per tile side
    if (s2) { H /= 2; W /= 2; }                                          // from here on the output (= phase image) size
    a.N = N; a.H = H; a.W = W; a.TH = (H + m - 1) / m; a.TW = (W + m - 1) / m;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = ups ? 4 * Cout : Cout;     // fused upsampling: four phases of virtual output channels
    a.nchunks = (4 * a.Gin + 15) / 16 * (s2 ? 4 : 1); a.T = N * a.TH * a.TW; a.relu = relu; a.ring = ring; a.ups_zero = ups_zero;
    a.sync_ws = sync_ws; a.sync_floats = sync_floats;
    {
        const int e = cnm_wino36s_try_launch(a, m, ups, cnm_stream(stream), s2);   // LDS-staged persistent variant where eligible
        if (e <= 0) return e;
        CNM_REQUIRE(!s2 && m != 3, CNM_ERR_BAD_ARG);                     // the stride-2 form and F(3x3,4x4) exist on the staged kernel only (cnm_conv_s2_winograd4_ok)
    }
    const int nblocks = (a.Cout / 64) * cnm_ceil_div(a.T, 16);
    if (ups) conv_winograd36_f32_kernel<4, 3, true><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    else if (ksize == 3) conv_winograd36_f32_kernel<4, 3><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    else conv_winograd36_f32_kernel<2, 5><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_pack_winograd4_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                         int Cout, int Cin, int rot, float* u_packed, void* stream) {
    return pack36(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, 3, rot, u_packed, stream);
}

// The packed filter of the DATA GRADIENT of a stride-1 convolution with weight w [Cw_out][Cw_in][k][k] (k = 3 or 5): the filter
// w'[ci][co] = w[co][ci] rotated by 180 degrees has Cw_in output and Cw_out input channels (Cw_in % 64 == 0); same layout as
// cnm_pack_winograd4_bn_f32 / cnm_pack_winograd5x5_bn_f32 of the flipped, transposed tensor, without materialising it.
extern "C" int cnm_pack_winograd4_dgrad_f32(const float* w_oihw, int Cw_out, int Cw_in, int ksize, float* u_packed, void* stream) {
    CNM_REQUIRE(ksize == 3 || ksize == 5, CNM_ERR_BAD_ARG);
    return pack36(w_oihw, nullptr, nullptr, 0.f, Cw_in, Cw_out, ksize, 0, u_packed, stream, 1);
}

extern "C" int cnm_conv3x3_winograd4_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                            const float* in_b, int Gb_total, int gb0, int Gb,
                                            float* out, int Gout_total, int gout0, int Cout,
                                            const float* u_packed, const float* b_packed,
                                            int N, int H, int W, int relu, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream);
}

// The same with a sync workspace (cnm_wino36_sync_floats() floats, its first 4096 bytes zero before the first use and left
// zero by every call, not shared by launches that may run concurrently): the LDS-staged kernel then splits its work into
// equal phase ranges, one per CU, and adds the partial outputs of a cut unit in a fixed order (bit-reproducible; not bit-equal
// to the unsplit evaluation, whose accumulation chain is not cut).
extern "C" int cnm_conv3x3_winograd4_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                                 const float* in_b, int Gb_total, int gb0, int Gb,
                                                 float* out, int Gout_total, int gout0, int Cout,
                                                 const float* u_packed, const float* b_packed,
                                                 int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream, 0, 0, sync_ws, sync_floats);
}

// 3x3 convolution of the bilinear 2x upsampling of `in` ([N][G][H][W][4] -> out [N][Gout][2H][2W][4]) on the
// low-resolution input: u_packed / b_packed are packed from the four composed phase filters as 4*Cout output channels
// (phase major).  with_ring = 0: complete result with REPLICATE padding of the upsampled image (differs from the
// reference's zero padding on the one-pixel output ring); with_ring = 1: ring pixels are left un-biased / un-activated
// for cnm_conv3x3_upsampled_ring_c4_f32, which turns them into the zero-padding result.
extern "C" int cnm_conv3x3_upsampled_winograd4_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                      float* out, int Gout_total, int gout0, int Cout,
                                                      const float* u_packed, const float* b_packed,
                                                      int N, int H, int W, int relu, int with_ring, void* stream) {
    return conv36(in, Gin_total, gin0, Gin, nullptr, 0, 0, 0, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream, 1, with_ring);
}

extern "C" int cnm_conv3x3_upsampled_winograd4_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                           float* out, int Gout_total, int gout0, int Cout,
                                                           const float* u_packed, const float* b_packed,
                                                           int N, int H, int W, int relu, int with_ring, float* sync_ws, size_t sync_floats, void* stream) {
    return conv36(in, Gin_total, gin0, Gin, nullptr, 0, 0, 0, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream, 1, with_ring, sync_ws, sync_floats);
}

// Phase-scatter form: four 3x3 stride-1 convolutions of `in` [N][G][H][W][4], one per pixel phase (a, b) of the output
// [N][Gout][2H][2W][4] -- out[2y + a][2x + b] = conv3x3(in, w_phase[2a + b])[y][x], zero padding -- in ONE launch: the four filters
// packed as 4*Cout output channels, phase major (cnm_pack_winograd4_bn_f32 of the [4*Cout, Cin, 3, 3] tensor), on the
// kernel of the fused up_conv layers with its interleaving store path.  This is the data gradient of a stride-2
// convolution (train.py:164-310 through autograd: dX phases are stride-1 convolutions of dY), written without the four
// strided scatter copies.  Same sync-workspace contract as cnm_conv3x3_winograd4_sync_c4_f32 (NULL allowed).
extern "C" int cnm_conv3x3_phase_scatter_winograd4_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                               float* out, int Gout_total, int gout0, int Cout,
                                                               const float* u_packed, const float* b_packed,
                                                               int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    return conv36(in, Gin_total, gin0, Gin, nullptr, 0, 0, 0, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream, 1, 0, sync_ws, sync_floats, 0, 1);
}

// The same with 4x4 phase filters (taps at offsets -1 .. 2 of either axis) on F(3x3,4x4): the data gradient of a stride-2 7x7
// convolution.  u_packed: cnm_pack_winograd36_f32(ksize 4) of the [4*Cout, Cin, 4, 4] tensor.  Staged kernel only: 4*Cout % 128
// == 0 and at least 6 x 2 tiles of 3 x 3 low-resolution pixels per image, else CNM_ERR_BAD_ARG.
extern "C" int cnm_conv4x4_phase_scatter_winograd_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                              float* out, int Gout_total, int gout0, int Cout,
                                                              const float* u_packed, const float* b_packed,
                                                              int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    CNM_REQUIRE(Cout > 0 && Cout % 32 == 0 && (W + 2) / 3 >= 6 && (H + 2) / 3 >= 2, CNM_ERR_BAD_ARG);
    return conv36(in, Gin_total, gin0, Gin, nullptr, 0, 0, 0, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 4, relu, stream, 1, 0, sync_ws, sync_floats, 0, 1);
}

// Plain 36-point pack of a [Cout, Cin, k, k] filter (k = 3, 4, 5: F(4x4,3x3), F(3x3,4x4), F(2x2,5x5)), no BatchNorm fold, no rotation
extern "C" int cnm_pack_winograd36_f32(const float* w_oihw, int Cout, int Cin, int ksize, float* u_packed, void* stream) {
    CNM_REQUIRE(ksize >= 3 && ksize <= 5, CNM_ERR_BAD_ARG);
    return pack36(w_oihw, nullptr, nullptr, 0.f, Cout, Cin, ksize, 0, u_packed, stream);
}

// F(2x2,5x5): the same 36-point machine with 2x2 output tiles (25 -> 9 multiplies per output; the row-wise kernel needs 15)
extern "C" int cnm_pack_winograd5x5_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                           int Cout, int Cin, int rot, float* u_packed, void* stream) {
    return pack36(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, 5, rot, u_packed, stream);
}

extern "C" int cnm_conv5x5_winograd_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                                const float* in_b, int Gb_total, int gb0, int Gb,
                                                float* out, int Gout_total, int gout0, int Cout,
                                                const float* u_packed, const float* b_packed,
                                                int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 5, relu, stream, 0, 0, sync_ws, sync_floats);
}

extern "C" int cnm_conv5x5_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                           const float* in_b, int Gb_total, int gb0, int Gb,
                                           float* out, int Gout_total, int gout0, int Cout,
                                           const float* u_packed, const float* b_packed,
                                           int N, int H, int W, int relu, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 5, relu, stream);
}

// Stride-2 5x5 (pad 2) / 7x7 (pad 3) convolution as a stride-1 convolution of the four pixel phases of the input on the
// LDS-staged 36-point kernel: 5x5 -> four 3x3 phase filters, F(4x4,3x3), 9 multiplies per output instead of 25 (the row-wise
// phase kernel: 15); 7x7 -> four 4x4 phase filters, F(3x3,4x4), 16 instead of 49 (22.75).  H, W (even) are the INPUT size,
// the output is [N][Gout][H/2][W/2][4].  Needs the sync workspace of cnm_conv3x3_winograd4_sync_c4_f32 and a shape
// cnm_conv_s2_winograd4_ok() accepts (there is no gather-fed twin to fall back to): CNM_ERR_BAD_ARG otherwise.
extern "C" size_t cnm_packed_winograd4_s2_floats(int Cout, int Cin) { return 4 * cnm_packed_winograd4_floats(Cout, Cin); }

extern "C" int cnm_pack_winograd4_s2_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                            int Cout, int Cin, int ksize, int rot, float* u_packed, void* stream) {
    CNM_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, CNM_ERR_BAD_ARG);
    return pack36(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, ksize, rot, u_packed, stream, 0, 1);
}

extern "C" int cnm_conv_s2_winograd4_ok(int Cout, int H, int W, int ksize) {
    if ((ksize != 3 && ksize != 5 && ksize != 7) || Cout <= 0 || Cout % 128 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    const int m = ksize == 7 ? 3 : 4, th = (H / 2 + m - 1) / m, tw = (W / 2 + m - 1) / m;
    return tw >= 12 || (tw >= 6 && th >= 2) || (tw >= 3 && th >= 3);
}

extern "C" int cnm_conv_s2_winograd4_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                                 const float* in_b, int Gb_total, int gb0, int Gb,
                                                 float* out, int Gout_total, int gout0, int Cout,
                                                 const float* u_packed, const float* b_packed,
                                                 int N, int H, int W, int ksize, int relu, float* sync_ws, size_t sync_floats, void* stream) {
    CNM_REQUIRE(cnm_conv_s2_winograd4_ok(Cout, H, W, ksize), CNM_ERR_BAD_ARG);
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, ksize, relu, stream, 0, 0, sync_ws, sync_floats, 1);
}

// ------------------------------------------------------------------ ring of the fused upsample + 3x3
// The composed phase filters convolve the upsampled image with replicate padding; the reference zero-pads it.  The two
// differ only on the one-pixel ring of the output, by the filter taps that land outside the image:
//     out[co][Y][X] = pre[co][Y][X] - sum_{(ky,kx): (Y+ky-1, X+kx-1) outside} sum_ci w[co][ci][ky][kx] up[ci][clamp(Y+ky-1)][clamp(X+kx-1)]
// (pre = what the main kernel stored for ring pixels: no bias, no ReLU).  Workgroup = 64 couts x 32 consecutive pixels of
// one side (top / bottom row, left / right column without the corners); the 34 upsampled border samples a block needs are
// interpolated from the low-resolution input into LDS once per 16-channel chunk and feed the three taps of the side as
// shifted B operands of v_mfma_f32_16x16x4_f32; the two extra taps of a corner pixel are added by the row block that
// owns it.  w_ring: [tap 9][chunk][cout/16][lane][4] in MFMA A-operand order (cnm_pack_upsampled_ring_f32).
#ifndef RING_NW
#define RING_NW 4        // waves per workgroup of the ring pass = chunks of the reduction in flight per workgroup.  [r6] eight were measured: SLOWER (23.4 against 20.5 us, 38.1 against 36.1: tools/ring_ab.sh, profiles/r6_ring_ab.txt) -- the pass scales with its work (13 us for 8 images, 21-23 for 16), it is not one latency chain per chunk
#endif
struct RingArgs {
    const float* in; float* out; const float* wr; const float* bias;
    int N, H, W, Gin_tot, gin0, Gin, Gout_tot, gout0, Cout, nchunks, relu;
    int nrow, ncol;                      // 32-pixel blocks per row side / per column side
};

// Four channels (virtual c4 group g4) of one low-resolution pixel: fp32 c4 tensors directly, fp16 c8 tensors (HALF) as one
// half of the 16-byte group g4 >> 1 widened to fp32 -- the ring pass then runs the same fp32 arithmetic on either.
typedef _Float16 ring_f16x4 __attribute__((ext_vector_type(4)));
template <bool HALF>
__device__ __forceinline__ float4 ring_ld4(const float* __restrict__ in, int img, int G_tot, int g0, int g4, int HW, int pix) {
    if constexpr (HALF) {
        const char* p = reinterpret_cast<const char*>(in + c4_offset(img, G_tot, g0 + (g4 >> 1), HW, pix)) + (g4 & 1) * 8;
        const ring_f16x4 h = *reinterpret_cast<const ring_f16x4*>(p);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
        return *reinterpret_cast<const float4*>(in + c4_offset(img, G_tot, g0 + g4, HW, pix));
    }
}

template <bool HALF>
__device__ __forceinline__ float4 ring_up_sample(const float* __restrict__ in, int img, int G_tot, int g0, int g4, int H, int W, int Y, int X) {   // upsample2x_c4_kernel's arithmetic
    const float sy = fmaxf((Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((X + 0.5f) * 0.5f - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const int HW = H * W;
    const float4 p00 = ring_ld4<HALF>(in, img, G_tot, g0, g4, HW, y0 * W + x0), p01 = ring_ld4<HALF>(in, img, G_tot, g0, g4, HW, y0 * W + x1);
    const float4 p10 = ring_ld4<HALF>(in, img, G_tot, g0, g4, HW, y1 * W + x0), p11 = ring_ld4<HALF>(in, img, G_tot, g0, g4, HW, y1 * W + x1);
    float4 v;
    v.x = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
    v.y = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
    v.z = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
    v.w = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
    return v;
}

template <bool HALF>                                                     // HALF: fp16 c8 input / output (a.Gin = virtual c4 groups = 2 x the c8 groups)
__global__ __launch_bounds__(64 * RING_NW) void conv_upsampled_ring_kernel(const RingArgs a) {
    // The reduction is short and latency-bound (a few hundred workgroups, 8-32 chunks each): the RING_NW waves split the
    // chunks (wave w takes chunks w, w + RING_NW, ...), each staging its own chunk in a private LDS region (no workgroup
    // barrier in the loop) for all 64 couts, and the partial sums meet in LDS at the end.
    constexpr int PB = 2, NPX = 16 * PB;                                 // 16-pixel MFMA blocks / pixels per workgroup
    constexpr int LD = 20;                                               // row pitch in floats: conflict-free ds_read_b128
    constexpr int LSZ = (NPX + 2) * LD + 4 * 16;                         // border samples j' = 0..NPX+1 (16 channels) + up to 4 corner samples
    __shared__ __attribute__((aligned(16))) float smem[RING_NW * LSZ > RING_NW * PB * 64 * 4 ? RING_NW * LSZ : RING_NW * PB * 64 * 4];   // 12 KB at four waves (fits next to two 72 KB convolution workgroups): per-wave staging, then the partial sums
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float* Ls = smem + wave * LSZ;
    float* Es = Ls + (NPX + 2) * LD;
    const int Ho = 2 * a.H, Wo = 2 * a.W, HW = a.H * a.W;
    const int tilesC = a.Cout / 64, segs = 2 * a.nrow + 2 * a.ncol;
    int b = blockIdx.x;
    const int cblk = b % tilesC; b /= tilesC;
    const int seg = b % segs, img = b / segs;
    // side 0 top, 1 bottom, 2 left, 3 right; p0 = first pixel of the block along the side
    const int side = seg < a.nrow ? 0 : seg < 2 * a.nrow ? 1 : seg < 2 * a.nrow + a.ncol ? 2 : 3;
    const int p0 = NPX * (side == 0 ? seg : side == 1 ? seg - a.nrow : side == 2 ? seg - 2 * a.nrow : seg - 2 * a.nrow - a.ncol);
    const bool rowside = side < 2;
    const int len = rowside ? Wo : Ho - 2;                               // pixels along the side
    // corner extras of a row block: e = 2 * (right corner) + k, k-th of the two taps on the column outside the image
    //   top:    taps (1,kx),(2,kx) read up[0][X], up[1][X];   bottom: taps (0,kx),(1,kx) read up[Ho-2][X], up[Ho-1][X]
    const bool hasL = rowside && p0 == 0, hasR = rowside && (Wo - 1 - p0) < NPX && (Wo - 1 - p0) >= 0;
    const int jR = Wo - 1 - p0;                                          // block-local index of the right corner pixel

    f32x4 acc[4][PB];                                                    // [cout group of 16][pixel block of 16]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < PB; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ncb16 = a.Cout / 16;
    const float4* wbase = reinterpret_cast<const float4*>(a.wr) + lane + (size_t)(cblk * 4) * 64;
    const size_t tapstride = (size_t)a.nchunks * ncb16 * 64, chunkstride = (size_t)ncb16 * 64;
    const int col = lane & 15, kg = lane >> 4;

    for (int c = wave; c < a.nchunks; c += RING_NW) {
        // all global loads of the chunk first (weights of the three taps, then the four low-resolution texels of each
        // border sample), arithmetic after: the loop is a latency chain, not a throughput problem
        float4 af[3][4];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = side == 0 ? s : side == 1 ? 6 + s : side == 2 ? 3 * s : 3 * s + 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) af[s][g] = wbase[(size_t)tap * tapstride + (size_t)c * chunkstride + (size_t)g * 64];
        }
        constexpr int NIT = ((NPX + 2) * 4 + 63) / 64;
        float4 tx[NIT][4]; float lyv[NIT], lxv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = lane + 64 * i, j = min(it >> 2, NPX + 1), q = it & 3, g = c * 4 + q;
            int Y, X;
            if (rowside) { Y = side == 0 ? 0 : Ho - 1; X = min(max(p0 + j - 1, 0), Wo - 1); }
            else { X = side == 2 ? 0 : Wo - 1; Y = min(p0 + j, Ho - 1); }
            const float sy = fmaxf((Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((X + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
            lyv[i] = sy - y0; lxv[i] = sx - x0;
            const int gl = min(g, a.Gin - 1);
            tx[i][0] = ring_ld4<HALF>(a.in, img, a.Gin_tot, a.gin0, gl, HW, y0 * a.W + x0); tx[i][1] = ring_ld4<HALF>(a.in, img, a.Gin_tot, a.gin0, gl, HW, y0 * a.W + x1);
            tx[i][2] = ring_ld4<HALF>(a.in, img, a.Gin_tot, a.gin0, gl, HW, y1 * a.W + x0); tx[i][3] = ring_ld4<HALF>(a.in, img, a.Gin_tot, a.gin0, gl, HW, y1 * a.W + x1);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = lane + 64 * i, j = it >> 2, q = it & 3, g = c * 4 + q;
            const float ly = lyv[i], lx = lxv[i], hy = 1.f - ly, hx = 1.f - lx;
            const float4 p00 = tx[i][0], p01 = tx[i][1], p10 = tx[i][2], p11 = tx[i][3];
            float4 v;                                                    // upsample2x_c4_kernel's arithmetic
            v.x = hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x);
            v.y = hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y);
            v.z = hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z);
            v.w = hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w);
            if (g >= a.Gin) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < (NPX + 2) * 4) *reinterpret_cast<float4*>(Ls + j * LD + q * 4) = v;
        }
        if ((hasL || hasR) && lane < 16) {                               // lane = (extra e = lane >> 2, quad q = lane & 3)
            const int e = lane >> 2, q = lane & 3, g = c * 4 + q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool right = e >> 1;
            if (g < a.Gin && (right ? hasR : hasL)) {
                v = ring_up_sample<HALF>(a.in, img, a.Gin_tot, a.gin0, g, a.H, a.W, side == 0 ? (e & 1) : Ho - 2 + (e & 1), right ? Wo - 1 : 0);
            }
            *reinterpret_cast<float4*>(Es + e * 16 + q * 4) = v;
        }
        __builtin_amdgcn_wave_barrier();                                 // wave-private region: LDS operations of one wave complete in order
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float4 bw[PB];
#pragma unroll
            for (int blk = 0; blk < PB; ++blk) bw[blk] = *reinterpret_cast<const float4*>(Ls + (blk * 16 + col + s) * LD + kg * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 aw = af[s][g];
#pragma unroll
                for (int blk = 0; blk < PB; ++blk) {
                    acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.x, bw[blk].x, acc[g][blk], 0, 0, 0);
                    acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.y, bw[blk].y, acc[g][blk], 0, 0, 0);
                    acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.z, bw[blk].z, acc[g][blk], 0, 0, 0);
                    acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.w, bw[blk].w, acc[g][blk], 0, 0, 0);
                }
            }
        }
        if (hasL || hasR) {                                              // workgroup-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool right = e >> 1;
                if (!(right ? hasR : hasL)) continue;
                const int k = e & 1, jc = right ? jR : 0;
                const int ky = side == 0 ? 1 + k : k, kx = right ? 2 : 0;
                float4 bw = *reinterpret_cast<const float4*>(Es + e * 16 + kg * 4);
                if (col != (jc & 15)) bw = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 aw = wbase[(size_t)(ky * 3 + kx) * tapstride + (size_t)c * chunkstride + (size_t)g * 64];
#pragma unroll
                    for (int blk = 0; blk < PB; ++blk) {
                        if (blk != (jc >> 4)) continue;
                        acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.x, bw.x, acc[g][blk], 0, 0, 0);
                        acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.y, bw.y, acc[g][blk], 0, 0, 0);
                        acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.z, bw.z, acc[g][blk], 0, 0, 0);
                        acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.w, bw.w, acc[g][blk], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // reads done before the next chunk overwrites the region
    }

    // ---- the four partial sums meet, one cout group per round (8 KB): wave w finishes cout group w
    f32x4* red = reinterpret_cast<f32x4*>(smem);                         // [src wave][pixel block][lane]
    f32x4 sum[PB];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __syncthreads();                                                 // staging reads / previous round's sums done
#pragma unroll
        for (int blk = 0; blk < PB; ++blk) red[(wave * PB + blk) * 64 + lane] = acc[g][blk];
        __syncthreads();
        if (wave == g) {
#pragma unroll
            for (int blk = 0; blk < PB; ++blk) {
                sum[blk] = red[blk * 64 + lane];
#pragma unroll
                for (int src = 1; src < RING_NW; ++src) sum[blk] += red[(src * PB + blk) * 64 + lane];
            }
        }
    }

    // ---- epilogue: row = cout 4*kg + r of group `wave`, col = pixel blk*16 + col: out = act(pre - correction + bias)
    if (wave >= 4) return;                                               // waves 0-3 each finish one 16-channel group
    const int co = cblk * 64 + wave * 16 + 4 * kg;
    const float4 bias = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int blk = 0; blk < PB; ++blk) {
        const int j = p0 + blk * 16 + col;
        if (j >= len) continue;
        const int Y = side == 0 ? 0 : side == 1 ? Ho - 1 : 1 + j, X = side == 2 ? 0 : side == 3 ? Wo - 1 : j;
        float4 v;
        ring_f16x4* oh = nullptr; float4* o = nullptr;
        if constexpr (HALF) {
            oh = reinterpret_cast<ring_f16x4*>(reinterpret_cast<char*>(a.out + c4_offset(img, a.Gout_tot, a.gout0 + (co >> 3), Ho * Wo, Y * Wo + X)) + (co & 4) * 2);
            const ring_f16x4 h = *oh;
            v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        } else {
            o = reinterpret_cast<float4*>(a.out + c4_offset(img, a.Gout_tot, a.gout0 + (co >> 2), Ho * Wo, Y * Wo + X));
            v = *o;
        }
        v.x = v.x - sum[blk][0] + bias.x; v.y = v.y - sum[blk][1] + bias.y; v.z = v.z - sum[blk][2] + bias.z; v.w = v.w - sum[blk][3] + bias.w;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if constexpr (HALF) { const ring_f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; *oh = h; }
        else *o = v;
    }
}

// w_ring[tap][chunk][cout/16][lane][4] = w[co = 16 cb + (lane & 15)][ci = 16 chunk + 4 (lane >> 4) + e][tap] * BatchNorm scale
__global__ void pack_upsampled_ring_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                           float eps, int Cout, int Cin, int nchunks, float* __restrict__ wr) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb16 = Cout / 16;
    const long long total = 9ll * nchunks * ncb16 * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int cb = (int)(r % ncb16); r /= ncb16;
    const int chunk = (int)(r % nchunks), tap = (int)(r / nchunks);
    const int co = cb * 16 + (lane & 15), ci = chunk * 16 + 4 * (lane >> 4) + e;
    float v = 0.f;
    if (ci < Cin) {
        const double sc = gamma ? (double)gamma[co] / sqrt((double)var[co] + (double)eps) : 1.0;
        v = (float)((double)w[((size_t)co * Cin + ci) * 9 + tap] * sc);
    }
    wr[idx] = v;
}

extern "C" size_t cnm_packed_upsampled_ring_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
    return (size_t)9 * ((4 * ((Cin + 3) / 4) + 15) / 16) * Cout * 16;
}

extern "C" int cnm_pack_upsampled_ring_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                           int Cout, int Cin, float* w_ring, void* stream) {
    CNM_REQUIRE(w_oihw && w_ring && Cout > 0 && Cout % 64 == 0 && Cin > 0 && !bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    const long long total = 9ll * nchunks * Cout * 16;
    pack_upsampled_ring_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, nchunks, w_ring);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv3x3_upsampled_ring_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                 float* out, int Gout_total, int gout0, int Cout,
                                                 const float* w_ring, const float* b_packed,
                                                 int N, int H, int W, int relu, void* stream) {
    CNM_REQUIRE(in && out && w_ring && N > 0 && H > 1 && W > 1 && Gin > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total && gin0 >= 0 && gin0 + Gin <= Gin_total, CNM_ERR_BAD_ARG);
    RingArgs a;
    a.in = in; a.out = out; a.wr = w_ring; a.bias = b_packed;
    a.N = N; a.H = H; a.W = W; a.Gin_tot = Gin_total; a.gin0 = gin0; a.Gin = Gin; a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (4 * Gin + 15) / 16; a.relu = relu;
    a.nrow = cnm_ceil_div(2 * W, 32); a.ncol = cnm_ceil_div(2 * H - 2, 32);              // 32-pixel blocks (conv_upsampled_ring_kernel NPX)
    const int nblocks = N * (2 * a.nrow + 2 * a.ncol) * (Cout / 64);
    conv_upsampled_ring_kernel<false><<<nblocks, 64 * RING_NW, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// fp16 twin for cnm_conv3x3_upsampled_c8_f16: `in` / `out` are c8 half tensors (Gin = 16-byte groups of 8 channels), w_ring and
// b_packed the same fp32 buffers as above (the ring's arithmetic stays fp32; only the loads and the read-modify-write differ).
extern "C" int cnm_conv3x3_upsampled_ring_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                                                 void* out, int Gout_total, int gout0, int Cout,
                                                 const float* w_ring, const float* b_packed,
                                                 int N, int H, int W, int relu, void* stream) {
    CNM_REQUIRE(in && out && w_ring && N > 0 && H > 1 && W > 1 && Gin > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 8 <= Gout_total && gin0 >= 0 && gin0 + Gin <= Gin_total, CNM_ERR_BAD_ARG);
    RingArgs a;
    a.in = static_cast<const float*>(in); a.out = static_cast<float*>(out); a.wr = w_ring; a.bias = b_packed;
    a.N = N; a.H = H; a.W = W; a.Gin_tot = Gin_total; a.gin0 = gin0; a.Gin = 2 * Gin; a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (8 * Gin + 15) / 16; a.relu = relu;
    a.nrow = cnm_ceil_div(2 * W, 32); a.ncol = cnm_ceil_div(2 * H - 2, 32);
    const int nblocks = N * (2 * a.nrow + 2 * a.ncol) * (Cout / 64);
    conv_upsampled_ring_kernel<true><<<nblocks, 64 * RING_NW, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
