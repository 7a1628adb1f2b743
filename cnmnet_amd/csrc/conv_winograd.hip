// Winograd F(2x2,3x3) convolution for the 3x3 stride-1 layers (53 % of a frame's conv FLOPs), fp32 MFMA.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 2x2 output tile, 4x4 input window d, 3x3 filter g:
//   16 multiplies per 4 outputs instead of 36 -> 2.25x fewer MFMA flops than the direct implicit GEMM, same fp32
//   data, transform coefficients in {0, +-1, +-1/2} (error growth ~1e-7 relative: far inside the 1e-3 parity bar).
//
// One fused kernel (no transformed tensors in HBM):
//   workgroup = 4 waves = 64 couts x 32 tiles (2x2 outputs each), TWO workgroups per CU; per 16-channel chunk
//     - every thread gathers half (2 of 4 channels) of the 4x4 window of one (tile, channel-quad) with 16 buffer
//       loads (out-of-range = 0: zero padding and ragged tails for free), transforms it in registers (B^T d B, adds
//       only) and writes the 16 frequency points to LDS  V[xi][tile][ci]  (double buffered, XOR-swizzled rows);
//     - every wave owns 16 couts x 32 tiles for ALL 16 frequency points on v_mfma_f32_16x16x4_f32: per point
//       2 ds_read_b128 of V + ONE 16-byte weight fragment straight from L2 (MFMA operand order, private to the
//       wave: no duplicate fetches) -> 8 MFMAs; 128 accumulator registers per lane, so two workgroups share a CU
//       (2 waves per SIMD from independent workgroups: one's prologue, output transform and load waits run under
//       the other's MFMAs -- VALU work itself never overlaps MFMAs on a SIMD);
//   epilogue: the 16 frequency values of each (cout, tile) sit in ONE lane -> A^T M A in registers, bias, ReLU,
//   float4 stores (c4 layout).
// Software pipeline inside a wave: while the MFMAs of chunk c issue, the same wave transforms chunk c+1 into the
// other V buffer, gathers the windows of chunk c+2 (two loads per step) and keeps 8 weight fragments in flight;
// one LDS-only barrier per chunk.  (The first version of this round used 32x32x2 MFMAs, 64 x 64 tiles and 256
// accumulators per wave = one wave per SIMD: 8-70 % slower, most on the short-K and low-resolution layers.)
#include "cnm_common.h"

#ifndef WINO_WD
#define WINO_WD 8        // weight fragments in flight per wave
#endif
#ifndef WINO_GSTART
#define WINO_GSTART 6    // first step of the window loads of the chunk two ahead (8 steps, two loads each)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt -- here that would wait, at
// every chunk, for the weight fragments and windows deliberately left in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct WinoArgs {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, TH, TW;                 // TH = ceil(H/2), TW = ceil(W/2) tiles
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*TH*TW tiles
    int s2;                              // stride 2: keep only output (0,0) of every 2x2 tile -> out is TH x TW
};

// Weight fragments: [chunk][cout/16][xi][lane][4], lane (i = l&15, kg = l>>4) holds U[xi][co = 16 cb + i][ci = 16 chunk + 4 kg + e].
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define F2OP(r, a, op, b) do { (r).x = (a).x op (b).x; (r).y = (a).y op (b).y; } while (0)

__global__ __launch_bounds__(256, 2) void conv3x3_winograd_f32_kernel(const WinoArgs a) {
    constexpr int TT = 32, VBUF = 16 * TT * 16;
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];           // 64 KB
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tilesC = a.Cout / 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = tile % tilesC, t0 = (tile / tilesC) * TT;
    const int HW = a.H * a.W, THW = a.TH * a.TW;

    // ---- loader: thread = (tile tl, half h of the channel quad qd = wave)
    const int tl = t & 31, hh = (t >> 5) & 1, qd = wave;
    const int tg = t0 + tl;
    const bool tvalid = tg < a.T;
    // Window offsets are loop-invariant per thread: off[ij] = byte offset of window pixel (i, j) in the first input view
    // (image term included), 0xFFFFFFFF outside the image / past the last tile (buffer loads return 0 there).  Per
    // chunk only SCALAR state changes: the channel-group offset (the load's soffset), the view (buffer resource) and a
    // zero-length resource for the ragged channel tail -- no vector instruction per load (VALU time is MFMA time).
    unsigned off[16];
    unsigned imgdelta;                                                   // image term of view 2 minus view 1
    {
        const int tt = tvalid ? tg : 0; const int img = tt / THW; const int rem = tt - img * THW; const int ty = rem / a.TW;
        const int py = 2 * ty - 1, px = 2 * (rem - ty * a.TW) - 1;
        const unsigned imgterm = (unsigned)img * (unsigned)a.Gin_tot * (unsigned)HW * 16u;
        imgdelta = (unsigned)img * (unsigned)a.Gin2_tot * (unsigned)HW * 16u - imgterm;
#pragma unroll
        for (int ij = 0; ij < 16; ++ij) {
            const int iy = py + (ij >> 2), ix = px + (ij & 3);
            const bool ok = tvalid & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            off[ij] = ok ? imgterm + (unsigned)(iy * a.W + ix) * 16u + hh * 8u : 0xFFFFFFFFu;
        }
    }
    float2 d[16];
    bool view2 = false;                                                  // wave-uniform: off[] already rebased to the second view
    __amdgpu_buffer_rsrc_t grsrc; unsigned gsoff;
    auto gather_begin = [&](int chunk) {
        const int g = chunk * 4 + qd;                                   // channel group of the (possibly concatenated) input
        const bool s1 = g < a.Gsplit;
        if (!s1 && !view2) {                                            // once per wave, when its quad crosses into the second view
            view2 = true;
#pragma unroll
            for (int ij = 0; ij < 16; ++ij) off[ij] = off[ij] == 0xFFFFFFFFu ? off[ij] : off[ij] + imgdelta;
        }
        const unsigned bytes = g < a.Gin ? (s1 ? a.in_bytes : a.in2_bytes) : 0u;
        grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 ? a.in : a.in2), 0, bytes, 0x00020000);
        gsoff = (unsigned)(s1 ? a.gin0 + g : a.gin2_0 + g - a.Gsplit) * (unsigned)HW * 16u;
    };
    auto gather_load = [&](int ij) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(grsrc, off[ij], gsoff, 0);
        d[ij] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    };
    float2 m[16];
    const int wofs = tl * 16 + (qd ^ ((tl >> 1) & 3)) * 4 + hh * 2;
    auto column_pass = [&](int j) {
        F2OP(m[0 * 4 + j], d[0 * 4 + j], -, d[2 * 4 + j]);
        F2OP(m[1 * 4 + j], d[1 * 4 + j], +, d[2 * 4 + j]);
        F2OP(m[2 * 4 + j], d[2 * 4 + j], -, d[1 * 4 + j]);
        F2OP(m[3 * 4 + j], d[1 * 4 + j], -, d[3 * 4 + j]);
    };
    auto row_pass = [&](int i, float* Vdst) {
        float2 v0, v1, v2, v3;
        F2OP(v0, m[i * 4 + 0], -, m[i * 4 + 2]);
        F2OP(v1, m[i * 4 + 1], +, m[i * 4 + 2]);
        F2OP(v2, m[i * 4 + 2], -, m[i * 4 + 1]);
        F2OP(v3, m[i * 4 + 1], -, m[i * 4 + 3]);
        float* dst = Vdst + (size_t)(i * 4) * TT * 16 + wofs;
        *reinterpret_cast<float2*>(dst) = v0;
        *reinterpret_cast<float2*>(dst + TT * 16) = v1;
        *reinterpret_cast<float2*>(dst + 2 * TT * 16) = v2;
        *reinterpret_cast<float2*>(dst + 3 * TT * 16) = v3;
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cb16 = cblk * 4 + wave, ncb16 = a.Cout / 16;
    const float4* ubase = reinterpret_cast<const float4*>(a.u) + lane + (size_t)cb16 * 16 * 64;
    const size_t ustride = (size_t)ncb16 * 16 * 64;                      // float4 per chunk
    const int rtile = lane & 15, kg = lane >> 4;
    const int voffA = rtile * 16 + (kg ^ ((rtile >> 1) & 3)) * 4;        // tile block 0: tiles 0..15
    const int voffB = (16 + rtile) * 16 + (kg ^ (((16 + rtile) >> 1) & 3)) * 4;

    constexpr int WD = WINO_WD;                                          // weight fragments in flight (steps of 8 MFMAs)
    float4 af[WD];
    {
#pragma unroll
        for (int s = 0; s < WD; ++s) af[s] = ubase[(size_t)s * 64];
    }
    gather_begin(0);
#pragma unroll
    for (int ij = 0; ij < 16; ++ij) gather_load(ij);
#pragma unroll
    for (int j = 0; j < 4; ++j) column_pass(j);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_pass(i, V);
    gather_begin(1);
#pragma unroll
    for (int ij = 0; ij < 16; ++ij) gather_load(ij);
    lds_barrier();
    for (int c = 0; c < a.nchunks; ++c) {
        const float* Vc = V + (c & 1) * VBUF;
        float* Vn = V + ((c + 1) & 1) * VBUF;
        const float4* uc = ubase + (size_t)c * ustride;
        const float4* un = ubase + (size_t)(c + 1 < a.nchunks ? c + 1 : c) * ustride;
        float4 bf0 = *reinterpret_cast<const float4*>(Vc + voffA);
        float4 bf1 = *reinterpret_cast<const float4*>(Vc + voffB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < 16; ++x) {                                   // one frequency point per step: 8 MFMAs
            const float4 aw = af[x % WD];
            const float4 b0 = bf0, b1 = bf1;
            if (x + 1 < 16) {
                bf0 = *reinterpret_cast<const float4*>(Vc + (size_t)(x + 1) * TT * 16 + voffA);
                bf1 = *reinterpret_cast<const float4*>(Vc + (size_t)(x + 1) * TT * 16 + voffB);
            }
            acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.x, b0.x, acc[x][0], 0, 0, 0);
            acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.x, b1.x, acc[x][1], 0, 0, 0);
            af[x % WD] = x + WD < 16 ? uc[(size_t)(x + WD) * 64] : un[(size_t)(x + WD - 16) * 64];
            if (x < 2) { column_pass(2 * x); column_pass(2 * x + 1); }
            else if (x < 4) { row_pass(2 * (x - 2), Vn); row_pass(2 * (x - 2) + 1, Vn); }
            if (x >= WINO_GSTART && x < WINO_GSTART + 8) {
                if (x == WINO_GSTART) gather_begin(c + 2);
                gather_load(2 * (x - WINO_GSTART)); gather_load(2 * (x - WINO_GSTART) + 1);
            }
            acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.y, b0.y, acc[x][0], 0, 0, 0);
            acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.y, b1.y, acc[x][1], 0, 0, 0);
            acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.z, b0.z, acc[x][0], 0, 0, 0);
            acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.z, b1.z, acc[x][1], 0, 0, 0);
            acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.w, b0.w, acc[x][0], 0, 0, 0);
            acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw.w, b1.w, acc[x][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    }

    // ---- epilogue: acc row = cout 4*(lane>>4)+r (one c4 group), col = tile lane&15 (+16 for the second block)
    const int co = cblk * 64 + wave * 16 + 4 * kg;
    const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int to = t0 + tb * 16 + rtile;
        if (to >= a.T) continue;
        const int oimg = to / THW, orem = to - oimg * THW, oty = orem / a.TW, otx = orem - oty * a.TW;
        const int opix = (2 * oty) * a.W + 2 * otx;
        float y[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[0 * 4 + j][tb][r] + acc[1 * 4 + j][tb][r] + acc[2 * 4 + j][tb][r];
                s1[j] = acc[1 * 4 + j][tb][r] - acc[2 * 4 + j][tb][r] - acc[3 * 4 + j][tb][r];
            }
            y[0][r] = s0[0] + s0[1] + s0[2]; y[1][r] = s0[1] - s0[2] - s0[3];
            y[2][r] = s1[0] + s1[1] + s1[2]; y[3][r] = s1[1] - s1[2] - s1[3];
        }
        if (a.s2) {                                                      // conv(stride 2, pad 1)[y][x] = conv(stride 1)[2y][2x] = element (0,0) of tile (y, x)
            float4 v = make_float4(y[0][0] + bb[0], y[0][1] + bb[1], y[0][2] + bb[2], y[0][3] + bb[3]);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), THW, orem)) = v;
            continue;
        }
        float* obase = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), HW, 0);
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            float4 v = make_float4(y[pq][0] + bb[0], y[pq][1] + bb[1], y[pq][2] + bb[2], y[pq][3] + bb[3]);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (2 * oty + (pq >> 1) < a.H && 2 * otx + (pq & 1) < a.W)
                *reinterpret_cast<float4*>(obase + (size_t)(opix + (pq >> 1) * a.W + (pq & 1)) * 4) = v;
        }
    }
}

// U = G g G^T (with the folded BatchNorm scale), packed in MFMA A-operand order [chunk][cout/16][xi][lane][4].
__global__ void pack_winograd_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                       float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb16 = Cout / 16;
    const long long total = (long long)nchunks * ncb16 * 16 * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), xi = (int)((idx >> 8) & 15);
    long long r = idx >> 12;
    const int cb = (int)(r % ncb16), chunk = (int)(r / ncb16);
    const int co = cb * 16 + (lane & 15), cp = chunk * 16 + 4 * (lane >> 4) + e;
    float v = 0.f;
    if (cp < Cin) {
        const int ci = (cp + rot) % Cin;
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        const int ai = xi >> 2, bi = xi & 3;
        double s = 0;
        for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) s += G[ai][p] * (double)g[p * 3 + q] * G[bi][q];
        if (gamma) s *= (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)s;
    }
    up[idx] = v;
}

extern "C" size_t cnm_packed_winograd_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    return (size_t)nchunks * 16 * Cout * 16;
}

extern "C" int cnm_pack_winograd_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                        int Cout, int Cin, int rot, float* u_packed, void* stream) {
    CNM_REQUIRE(w_oihw && u_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(!bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = (4 * ((Cin + 3) / 4) + 15) / 16;
    const long long total = (long long)nchunks * 16 * Cout * 16;
    pack_winograd_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

static int conv3x3_winograd_launch(int s2, const float* in_a, int Ga_total, int ga0, int Ga,
                                           const float* in_b, int Gb_total, int gb0, int Gb,
                                           float* out, int Gout_total, int gout0, int Cout,
                                           const float* u_packed, const float* b_packed,
                                           int N, int H, int W, int relu, void* stream) {
    CNM_REQUIRE(in_a && out && u_packed && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    WinoArgs a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    a.N = N; a.H = H; a.W = W; a.TH = (H + 1) / 2; a.TW = (W + 1) / 2;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (4 * a.Gin + 15) / 16; a.T = N * a.TH * a.TW; a.relu = relu; a.s2 = s2;
    const int nblocks = (Cout / 64) * cnm_ceil_div(a.T, 32);
    conv3x3_winograd_f32_kernel<<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv3x3_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                           const float* in_b, int Gb_total, int gb0, int Gb,
                                           float* out, int Gout_total, int gout0, int Cout,
                                           const float* u_packed, const float* b_packed,
                                           int N, int H, int W, int relu, void* stream) {
    return conv3x3_winograd_launch(0, in_a, Ga_total, ga0, Ga, in_b, Gb_total, gb0, Gb, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, relu, stream);
}

// Stride 2 (pad 1) through the same kernel: out[y][x] = element (0,0) of the F(2x2,3x3) tile (y, x); out is
// ceil(H/2) x ceil(W/2).  16 multiplies per kept output instead of 9, but 64 couts x 32 output pixels per workgroup and
// a 16-deep reduction step: for the 12x16 -> 6x8 layers the implicit-GEMM kernel has under a hundred workgroups.
extern "C" int cnm_conv3x3_s2_winograd_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                              float* out, int Gout_total, int gout0, int Cout,
                                              const float* u_packed, const float* b_packed,
                                              int N, int H, int W, int relu, void* stream) {
    return conv3x3_winograd_launch(1, in, Gin_total, gin0, Gin, nullptr, 0, 0, 0, out, Gout_total, gout0, Cout, u_packed, b_packed, N, H, W, relu, stream);
}
