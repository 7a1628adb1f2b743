// Winograd F(2x2,3x3) convolution for the 3x3 stride-1 layers (53 % of a frame's conv FLOPs), fp32 MFMA.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 2x2 output tile, 4x4 input window d, 3x3 filter g:
//   16 multiplies per 4 outputs instead of 36 -> 2.25x fewer MFMA flops than the direct implicit GEMM, same fp32
//   data, transform coefficients in {0, +-1, +-1/2} (error growth ~1e-7 relative: far inside the 1e-3 parity bar).
//
// One fused kernel (no transformed tensors in HBM):
//   workgroup = 4 waves = 64 couts x 64 tiles (2x2 outputs each); per 16-channel chunk
//     - every thread gathers the 4x4 window of ONE (tile, channel-quad) with 16 buffer loads (out-of-range = 0:
//       zero padding and ragged tails for free), transforms it in registers (B^T d B, adds only) and writes the
//       16 frequency points to LDS  V[xi][tile][ci]  (double buffered, XOR-swizzled rows);
//     - every wave owns 32 couts x 32 tiles for ALL 16 frequency points: per point 2 x (ds_read_b128 of V +
//       16-byte weight fragment straight from L2 in MFMA operand order) -> 8 v_mfma_f32_32x32x2_f32;
//       256 accumulator registers per lane (one wave per SIMD, 512-register budget);
//   epilogue: the 16 frequency values of each (cout, tile) sit in ONE lane -> A^T M A in registers, bias, ReLU,
//   four float4 stores (c4 layout).
// Software pipeline (one wave per SIMD, so the overlap is inside the wave): while the MFMAs of chunk c run, the
// same wave transforms chunk c+1 into the other V buffer, gathers the windows of chunk c+2 and keeps 16 weight
// fragments in flight; one barrier per chunk.
#include "cnm_common.h"

#ifndef WINO_ABL_GATHER
#define WINO_ABL_GATHER 0   // ablation switches (timing studies only; results are wrong when set)
#endif
#ifndef WINO_ABL_REFILL
#define WINO_ABL_REFILL 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt -- here that would wait, at
// every chunk, for the weight fragments and windows deliberately left in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, TH, TW;                 // TH = ceil(H/2), TW = ceil(W/2) tiles
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*TH*TW tiles
};

__device__ __forceinline__ float4 wino_load(const float* base, unsigned bytes, unsigned voff) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

#define F4OP(r, a, op, b) do { (r).x = (a).x op (b).x; (r).y = (a).y op (b).y; (r).z = (a).z op (b).z; (r).w = (a).w op (b).w; } while (0)

__global__ __launch_bounds__(256, 1) void conv3x3_winograd_f32_kernel(const WinoArgs a) {
    // V[buf][xi][tile][16 ci]: rows of 64 B, the four 16-byte slots of a row XOR-swizzled with ((tile >> 2) & 3) so that
    // 16 consecutive rows cover all 64 banks:
    // both the b128 writes (lanes = tiles, one slot) and the b128 operand reads are bank-conflict free without padding;
    // two buffers (128 KB): chunk c+1 is transformed while chunk c feeds the MFMAs.
    constexpr int VBUF = 16 * 64 * 16;
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wc = wave >> 1, wt = wave & 1;
    const int tilesC = a.Cout / 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = tile % tilesC, t0 = (tile / tilesC) * 64;
    const int HW = a.H * a.W, THW = a.TH * a.TW;

    // ---- loader: thread = (tile tl, channel quad qd of the chunk)
    const int tl = t & 63, qd = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tg = t0 + tl;
    const bool tvalid = tg < a.T;
    int img, py, px;
    { const int tt = tvalid ? tg : 0; img = tt / THW; const int rem = tt - img * THW; const int ty = rem / a.TW; py = 2 * ty - 1; px = 2 * (rem - ty * a.TW) - 1; }
    float4 d[16];
    // window loads of one chunk, issuable one at a time (spread over the MFMA steps: a burst of 16 would hold the
    // wave -- and the other three -- at the texture addresser while the matrix pipe drains)
    const float* gbase; unsigned gbytes, gofs; bool gok;
    auto gather_begin = [&](int chunk) {
        const int g = chunk * 4 + qd;                                   // channel group of the (possibly concatenated) input
        const bool s1 = g < a.Gsplit;
        gbase = s1 ? a.in : a.in2;
        gbytes = s1 ? a.in_bytes : a.in2_bytes;
        gofs = s1 ? (unsigned)((img * a.Gin_tot + a.gin0 + g) * HW) : (unsigned)((img * a.Gin2_tot + a.gin2_0 + g - a.Gsplit) * HW);
        gok = tvalid & (g < a.Gin);
    };
    auto gather_load = [&](int ij) {
        const int iy = py + (ij >> 2), ix = px + (ij & 3);
        const bool ok = gok & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        d[ij] = wino_load(gbase, gbytes, ok ? (gofs + (unsigned)(iy * a.W + ix)) * 16u : 0xFFFFFFFFu);
    };
    auto gather = [&](int chunk) {
        gather_begin(chunk);
#pragma unroll
        for (int ij = 0; ij < 16; ++ij) gather_load(ij);
    };
    float4 m[16];
    const int wslot = (qd ^ ((tl >> 2) & 3)) * 4;
    auto column_pass = [&](int j) {                                      // m = B^T d, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        F4OP(m[0 * 4 + j], d[0 * 4 + j], -, d[2 * 4 + j]);
        F4OP(m[1 * 4 + j], d[1 * 4 + j], +, d[2 * 4 + j]);
        F4OP(m[2 * 4 + j], d[2 * 4 + j], -, d[1 * 4 + j]);
        F4OP(m[3 * 4 + j], d[1 * 4 + j], -, d[3 * 4 + j]);
    };
    auto row_pass = [&](int i, float* Vdst) {                            // V = m B, four frequency points of row i to LDS
        float4 v0, v1, v2, v3;
        F4OP(v0, m[i * 4 + 0], -, m[i * 4 + 2]);
        F4OP(v1, m[i * 4 + 1], +, m[i * 4 + 2]);
        F4OP(v2, m[i * 4 + 2], -, m[i * 4 + 1]);
        F4OP(v3, m[i * 4 + 1], -, m[i * 4 + 3]);
        float* dst = Vdst + ((size_t)(i * 4) * 64 + tl) * 16 + wslot;
        *reinterpret_cast<float4*>(dst) = v0;
        *reinterpret_cast<float4*>(dst + 64 * 16) = v1;
        *reinterpret_cast<float4*>(dst + 2 * 64 * 16) = v2;
        *reinterpret_cast<float4*>(dst + 3 * 64 * 16) = v3;
    };

    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // weights in MFMA operand order: [chunk][xi][cout block of 32][kq][lane][4]
    const int cb = cblk * 2 + wc, ncb = a.Cout / 32;
    const float4* ubase = reinterpret_cast<const float4*>(a.u) + lane;
    const int rrow = wt * 32 + (lane & 31);
    const int voff0 = rrow * 16 + ((lane >> 5) ^ ((rrow >> 2) & 3)) * 4;      // kq = 0: slot (lane>>5)
    const int voff1 = rrow * 16 + ((2 + (lane >> 5)) ^ ((rrow >> 2) & 3)) * 4;  // kq = 1: slot 2 + (lane>>5)

    constexpr int WD = 16;                                               // weight fragments in flight (steps of 4 MFMAs): half a chunk
    float4 af[WD];
    // VMEM issue schedule of double step ds in the phase of chunk c: refill the two fragment slots just consumed
    // (they wrap into chunk c+1 from ds = 8 on), and from ds = 2 (column passes done, d free) two window loads of
    // chunk c+2.  The prologue replays the SAME order as a phase "c = -1" without MFMAs, so the loads in flight at
    // the loop head are ordered identically on both paths into the loop and the compiler's vmcnt counts are exact
    // instead of a conservative minimum.
    auto vmem_ds = [&](int ds, int c, const float4* uc, const float4* un) {
        const int sa = (ds >> 1) * 4 + (ds & 1), sb = sa + 2;
        const int na = sa + WD, nb = sb + WD;
        if (!WINO_ABL_REFILL && (uc || na >= 32)) {
            af[sa % WD] = na < 32 ? uc[((size_t)(na >> 1) * ncb * 2 + (na & 1)) * 64] : un[((size_t)((na - 32) >> 1) * ncb * 2 + (na & 1)) * 64];
            af[sb % WD] = nb < 32 ? uc[((size_t)(nb >> 1) * ncb * 2 + (nb & 1)) * 64] : un[((size_t)((nb - 32) >> 1) * ncb * 2 + (nb & 1)) * 64];
        }
        if (!WINO_ABL_GATHER && ds >= 2 && ds < 10) {
            if (ds == 2) gather_begin(c + 2);
            gather_load(2 * (ds - 2)); gather_load(2 * (ds - 2) + 1);
        }
    };
    {                                                                    // prologue: everything that does not depend on LDS goes out first
        const float4* u0 = ubase + ((size_t)0 * ncb + cb) * 2 * 64;
#pragma unroll
        for (int s = 0; s < WD; ++s) af[s] = u0[((size_t)(s >> 1) * ncb * 2 + (s & 1)) * 64];   // first half of chunk 0's weights
    }
    gather(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) column_pass(j);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_pass(i, V);
    gather(1);
    __syncthreads();
    for (int c = 0; c < a.nchunks; ++c) {
        const float* Vc = V + (c & 1) * VBUF;
        float* Vn = V + ((c + 1) & 1) * VBUF;
        const float4* uc = ubase + ((size_t)(c * 16) * ncb + cb) * 2 * 64;
        const float4* un = ubase + ((size_t)((c + 1 < a.nchunks ? c + 1 : c) * 16) * ncb + cb) * 2 * 64;
        // double step ds = (frequency pair p, k-quad kq): the two points' MFMAs alternate, so consecutive MFMAs never
        // chain on the same accumulator
        float4 bf0 = *reinterpret_cast<const float4*>(Vc + voff0);
        float4 bf1 = *reinterpret_cast<const float4*>(Vc + 64 * 16 + voff0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ds = 0; ds < 16; ++ds) {
            const int p2 = (ds >> 1) * 2, kq = ds & 1;
            const int sa = p2 * 2 + kq, sb = sa + 2;                     // fragment slots (step = xi*2 + kq)
            const float4 a0 = af[sa % WD], a1 = af[sb % WD];
            const float4 b0 = bf0, b1 = bf1;
            if (ds + 1 < 16) {
                const int np2 = ((ds + 1) >> 1) * 2, nkq = (ds + 1) & 1;
                bf0 = *reinterpret_cast<const float4*>(Vc + (size_t)np2 * 64 * 16 + (nkq ? voff1 : voff0));
                bf1 = *reinterpret_cast<const float4*>(Vc + (size_t)(np2 + 1) * 64 * 16 + (nkq ? voff1 : voff0));
            }
            acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[p2], 0, 0, 0);
            acc[p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[p2 + 1], 0, 0, 0);
            // the input transform of chunk c+1 rides in the shadow of the MFMAs (windows gathered one phase earlier;
            // past the last chunk they are all out of range = 0, written to the idle buffer, never read)
            if (ds < 2) { column_pass(2 * ds); column_pass(2 * ds + 1); }
            else if (ds < 4) { row_pass(2 * (ds - 2), Vn); row_pass(2 * (ds - 2) + 1, Vn); }
            vmem_ds(ds, c, uc, un);
            acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[p2], 0, 0, 0);
            acc[p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[p2 + 1], 0, 0, 0);
            acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[p2], 0, 0, 0);
            acc[p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[p2 + 1], 0, 0, 0);
            acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[p2], 0, 0, 0);
            acc[p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[p2 + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                                   // V[c+1] complete, V[c] free for chunk c+2
    }

    // ---- epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; acc row = cout (r&3)+8*(r>>2)+4*(lane>>5), col = tile lane&31
    const int to = t0 + wt * 32 + (lane & 31);
    if (to >= a.T) return;
    const int oimg = to / THW, orem = to - oimg * THW, oty = orem / a.TW, otx = orem - oty * a.TW;
    const int opix = (2 * oty) * a.W + 2 * otx;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = cblk * 64 + wc * 32 + 8 * q + 4 * (lane >> 5);
        const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        float y[4][4];                                                   // [pixel 2a+b][channel e]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
                s1[j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
            }
            y[0][e] = s0[0] + s0[1] + s0[2]; y[1][e] = s0[1] - s0[2] - s0[3];
            y[2][e] = s1[0] + s1[1] + s1[2]; y[3][e] = s1[1] - s1[2] - s1[3];
        }
        const float bb[4] = {b.x, b.y, b.z, b.w};
        float* obase = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), HW, 0);
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            float4 v = make_float4(y[pq][0] + bb[0], y[pq][1] + bb[1], y[pq][2] + bb[2], y[pq][3] + bb[3]);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (2 * oty + (pq >> 1) < a.H && 2 * otx + (pq & 1) < a.W)    // odd H / W: the last tile row / column is half outside
                *reinterpret_cast<float4*>(obase + (size_t)(opix + (pq >> 1) * a.W + (pq & 1)) * 4) = v;
        }
    }
}

// U = G g G^T (with the folded BatchNorm scale), packed in MFMA A-operand order
// [chunk][xi][cout/32][kq][lane][4]:  value(co = cb*32 + (lane&31), ci = chunk*16 + kq*8 + 4*(lane>>5) + e).
__global__ void pack_winograd_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                     float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb = Cout / 32;
    const long long total = (long long)nchunks * 16 * ncb * 2 * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), kq = (int)((idx >> 8) & 1);
    long long r = idx >> 9;
    const int cb = (int)(r % ncb); r /= ncb;
    const int xi = (int)(r % 16), chunk = (int)(r / 16);
    const int co = cb * 32 + (lane & 31), cp = chunk * 16 + kq * 8 + 4 * (lane >> 5) + e;
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

extern "C" int cnm_conv3x3_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
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
    a.nchunks = (4 * a.Gin + 15) / 16; a.T = N * a.TH * a.TW; a.relu = relu;
    const int nblocks = (Cout / 64) * cnm_ceil_div(a.T, 64);
    conv3x3_winograd_f32_kernel<<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
