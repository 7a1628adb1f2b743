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

#ifndef WINO4_ABL
#define WINO4_ABL 0       // ablation bit mask for timing studies (results are wrong when set): 1 no gather, 2 no weight refill, 4 no transform, 8 windows from one line, 16 weights from one fragment
#endif
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

struct Wino4Args {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, TH, TW;                 // TH = ceil(H/4), TW = ceil(W/4) tiles
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*TH*TW tiles
};

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

template <int M, int R>                                                  // M x M outputs per tile, R x R filter, M + R - 1 == 6
__global__ __launch_bounds__(256, 2) void conv_winograd36_f32_kernel(const Wino4Args a) {
    static_assert(M + R - 1 == 6, "36-point kernel");
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
            const int iy = py + i, ix = px + i;
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
        d[ij] = __uint_as_float((WINO4_ABL & 8) ? __builtin_amdgcn_raw_buffer_load_b32(grsrc, (unsigned)lane * 4u, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b32(grsrc, sat_add(roff[ij / 6], coff[ij % 6]), gsoff, 0));
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
            if (WINO4_ABL & 16) { af[x0 % WD] = ubase[0]; af[x1 % WD] = ubase[0]; }
            else if (!(WINO4_ABL & 2)) {
                af[x0 % WD] = x0 + WD < NXI ? uc[(size_t)(x0 + WD) * 64] : un[(size_t)(x0 + WD - NXI) * 64];
                af[x1 % WD] = x1 + WD < NXI ? uc[(size_t)(x1 + WD) * 64] : un[(size_t)(x1 + WD - NXI) * 64];
            }
            // between the MFMAs: the transform of chunk c+1 (double steps 0..5), then the window of chunk c+2 (WINO4_LPS
            // loads per double step; past the last chunk all out of range = 0, written to the idle buffer)
            if (xp < 3) { if (!(WINO4_ABL & 4)) { column_pass(2 * xp); column_pass(2 * xp + 1); } }
            else if (xp < 6) { if (!(WINO4_ABL & 4)) { row_pass(2 * (xp - 3), Vn); row_pass(2 * (xp - 3) + 1, Vn); } }
            else if (xp < 6 + 36 / WINO4_LPS && !(WINO4_ABL & 1)) {      // row-major (column-major issue order measured 5 % slower: worse line locality)
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
template <int R>
__global__ void pack_winograd36_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                      float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb16 = Cout / 16;
    const long long total = (long long)nchunks * ncb16 * 36 * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int xi = (int)(r % 36); r /= 36;
    const int cb = (int)(r % ncb16), chunk = (int)(r / ncb16);
    const int co = cb * 16 + (lane & 15), cp = chunk * 16 + 4 * (lane >> 4) + e;
    float v = 0.f;
    if (cp < Cin) {
        const int ci = (cp + rot) % Cin;
        const float* g = w + ((size_t)co * Cin + ci) * R * R;
        const double G3[6][3] = {{1. / 4, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
        const double G5[6][5] = {{1. / 4, 0, 0, 0, 0}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6},
                                 {1. / 24, 1. / 12, 1. / 6, 1. / 3, 2. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3, 2. / 3}, {0, 0, 0, 0, 1}};
        const int ai = xi / 6, bi = xi % 6;
        double s = 0;
        for (int p = 0; p < R; ++p) for (int q = 0; q < R; ++q) s += (R == 3 ? G3[ai][p] * G3[bi][q] : G5[ai][p] * G5[bi][q]) * (double)g[p * R + q];
        if (gamma) s *= (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)s;
    }
    up[idx] = v;
}

extern "C" size_t cnm_packed_winograd4_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    return (size_t)nchunks * 36 * Cout * 16;
}

static int pack36(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps, int Cout, int Cin, int ksize, int rot,
                  float* u_packed, void* stream) {
    CNM_REQUIRE(w_oihw && u_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    const long long total = (long long)nchunks * 36 * Cout * 16;
    const unsigned nb = (unsigned)cnm_ceil_div_ll(total, 256);
    if (ksize == 3) pack_winograd36_kernel<3><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed);
    else pack_winograd36_kernel<5><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

static int conv36(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
                  float* out, int Gout_total, int gout0, int Cout, const float* u_packed, const float* b_packed,
                  int N, int H, int W, int ksize, int relu, void* stream) {
    CNM_REQUIRE(in_a && out && u_packed && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    Wino4Args a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    const int m = ksize == 3 ? 4 : 2;                                    // outputs per tile side
    a.N = N; a.H = H; a.W = W; a.TH = (H + m - 1) / m; a.TW = (W + m - 1) / m;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (4 * a.Gin + 15) / 16; a.T = N * a.TH * a.TW; a.relu = relu;
    const int nblocks = (Cout / 64) * cnm_ceil_div(a.T, 16);
    if (ksize == 3) conv_winograd36_f32_kernel<4, 3><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    else conv_winograd36_f32_kernel<2, 5><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_pack_winograd4_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                         int Cout, int Cin, int rot, float* u_packed, void* stream) {
    return pack36(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, 3, rot, u_packed, stream);
}

extern "C" int cnm_conv3x3_winograd4_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                            const float* in_b, int Gb_total, int gb0, int Gb,
                                            float* out, int Gout_total, int gout0, int Cout,
                                            const float* u_packed, const float* b_packed,
                                            int N, int H, int W, int relu, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 3, relu, stream);
}

// F(2x2,5x5): the same 36-point machine with 2x2 output tiles (25 -> 9 multiplies per output; the row-wise kernel needs 15)
extern "C" int cnm_pack_winograd5x5_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                           int Cout, int Cin, int rot, float* u_packed, void* stream) {
    return pack36(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, 5, rot, u_packed, stream);
}

extern "C" int cnm_conv5x5_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                           const float* in_b, int Gb_total, int gb0, int Gb,
                                           float* out, int Gout_total, int gout0, int Cout,
                                           const float* u_packed, const float* b_packed,
                                           int N, int H, int W, int relu, void* stream) {
    return conv36(in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, 5, relu, stream);
}
