// Row-wise Winograd F(2,R) convolution for the RxR stride-1 layers with R = 5, 7 (depthNet conv2.0 and conv1.0:
// 34 % of a frame's conv FLOPs), fp32 MFMA.
//
// A 2-D F(2x2,RxR) needs (R+1)^2 = 36 / 64 frequency points -- more accumulators than a wave has registers -- so the
// transform runs along image rows only and the R kernel rows stay in the GEMM reduction:
//     out[y, 2t..2t+1] = AT  sum_{ky, ci} [ (G w[co, ci, ky, :]) (.) (BT in[ci, y+ky-R/2, 2t-R/2 .. 2t+R/2+1]) ]
//   (R+1)/2 multiplies per output and kernel row instead of R: 1.75x (R=7) / 1.67x (R=5) fewer MFMA flops, fp32 data
//   and accumulation; Toom-Cook points {0,+-1,+-2,+-1/2,inf} (R=7), {0,+-1,+-2,inf} (R=5), tables from
//   tools/wino1d_matrices.py (exact in rationals; measured fp32 error ~4e-6 of the output scale).
//
// Same machine as conv_winograd.hip: workgroup = 4 waves = 64 couts x 128 row-tiles, every wave 32 couts x 64 tiles
// for ALL R+1 frequency points (<= 256 accumulator registers, one wave per SIMD); per 16-deep chunk of the
// (ky, ci) reduction every thread gathers the R+1 pixel windows of TWO (tile, channel-quad) items with buffer loads
// (out of range = 0), transforms them in registers and writes V[xi][tile][k] to LDS (double buffered, swizzled);
// weight fragments come straight from L2 in MFMA operand order, one whole chunk ahead; the transform of chunk c+1
// and the gather of chunk c+2 ride in the shadow of the MFMAs of chunk c; one barrier per chunk.
#include "cnm_common.h"

#ifndef ROWS_ABL
#define ROWS_ABL 0   // ablation bit mask for timing studies (results are wrong when set): 1 no gather, 2 no weight refill, 4 no transform
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

struct RowArgs {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, TW;                     // TW = ceil(W/2) tiles per row
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*H*TW tiles
};

__device__ __forceinline__ float4 rw_load(const float* base, unsigned bytes, unsigned voff) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int R>
__global__ __launch_bounds__(256, 1) void conv_rows_winograd_f32_kernel(const RowArgs a) {
    using WM = RowWino<R>;
    constexpr int NX = R + 1, TT = 128, NSTEP = NX * 2;
    constexpr int VBUF = NX * TT * 16;                                   // V[buf][xi][tile][16 k], slots XOR-swizzled with ((tile >> 2) & 3)
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wc = wave >> 1, wt = wave & 1;
    const int tilesC = a.Cout / 64;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = tile % tilesC, t0 = (tile / tilesC) * TT;
    const int HW = a.H * a.W, THW = a.H * a.TW;

    // ---- loader: thread = tile tl, channel quads qd0 and qd0 + 2 of the chunk
    const int tl = t & 127, qd0 = __builtin_amdgcn_readfirstlane(t >> 7);
    const int tg = t0 + tl;
    const bool tvalid = tg < a.T;
    int img, py, px;
    { const int tt = tvalid ? tg : 0; img = tt / THW; const int rem = tt - img * THW; py = rem / a.TW; px = 2 * (rem - py * a.TW) - R / 2; }
    float4 d[2][NX];
    int ky[2], gq[2];                                                    // (kernel row, channel group) of the next chunk's two quads
#pragma unroll
    for (int it = 0; it < 2; ++it) { const int q = qd0 + 2 * it; ky[it] = q / a.Gin; gq[it] = q - ky[it] * a.Gin; }
    // window loads of the next chunk in sequence, issuable one at a time (spread over the MFMA steps: a burst of
    // 16 would hold all four waves at the texture addresser while the matrix pipe drains)
    const float* gbase[2]; unsigned gbytes[2], gofs[2]; bool gok[2];
    auto gather_begin = [&](int it) {
        const int g = gq[it];
        const bool s1 = g < a.Gsplit;
        gbase[it] = s1 ? a.in : a.in2;
        gbytes[it] = s1 ? a.in_bytes : a.in2_bytes;
        const int iy = py + ky[it] - R / 2;
        gofs[it] = (s1 ? (unsigned)((img * a.Gin_tot + a.gin0 + g) * HW) : (unsigned)((img * a.Gin2_tot + a.gin2_0 + g - a.Gsplit) * HW)) + (unsigned)(iy * a.W);
        gok[it] = tvalid & (ky[it] < R) & ((unsigned)iy < (unsigned)a.H);
        gq[it] += 4;                                                     // advance to the following chunk's quad
#pragma unroll
        for (int w = 0; w < 4; ++w) { const bool wrap = gq[it] >= a.Gin; gq[it] -= wrap ? a.Gin : 0; ky[it] += wrap; }   // Gin >= 1: at most 4 wraps
    };
    auto gather_load = [&](int it, int j) {
        const int ix = px + j;
        const bool ok = gok[it] & ((unsigned)ix < (unsigned)a.W);
        d[it][j] = rw_load(gbase[it], gbytes[it], ok ? (gofs[it] + (unsigned)ix) * 16u : 0xFFFFFFFFu);
    };
    auto gather = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            gather_begin(it);
#pragma unroll
            for (int j = 0; j < NX; ++j) gather_load(it, j);
        }
    };
    // V = BT d in NG = (R+1)/2 groups of two frequency points (even/odd factorisation of the +-p point pairs, packed
    // fp32 math): group 0 = (V0, V_R) [points 0, inf], group i = (V_{2i-1}, V_{2i}) [points +p_i, -p_i].
    constexpr int NG = NX / 2;
    auto transform_group = [&](int it, int grp, float* Vdst) {
        const int wslot = ((qd0 + 2 * it) ^ ((tl >> 2) & 3)) * 4;
        const f4p* x = reinterpret_cast<const f4p*>(&d[it][0]);
        f4p va, vb; int ka, kb;
        if constexpr (R == 7) {
            if (grp == 0) { va = f4_fma(5.25f, f4_sub(x[2], x[4]), f4_sub(x[6], x[0])); vb = f4_fma(5.25f, f4_sub(x[3], x[5]), f4_sub(x[7], x[1])); ka = 0; kb = 7; }
            else {
                f4p e, o;
                if (grp == 1) { e = f4_fma(-4.25f, x[4], f4_add(x[2], x[6])); o = f4_fma(-4.25f, x[3], f4_add(x[1], x[5])); }
                else if (grp == 2) { e = f4_fma(0.25f, x[2], f4_fma(-1.25f, x[4], x[6])); o = f4_fma(0.5f, x[1], f4_fma(-2.5f, x[3], f4_mul(2.f, x[5]))); }
                else { e = f4_fma(4.f, x[2], f4_fma(-5.f, x[4], x[6])); o = f4_fma(2.f, x[1], f4_fma(-2.5f, x[3], f4_mul(0.5f, x[5]))); }
                va = f4_add(e, o); vb = f4_sub(e, o); ka = 2 * grp - 1; kb = 2 * grp;
            }
        } else {
            if (grp == 0) { va = f4_fma(4.f, x[0], f4_fma(-5.f, x[2], x[4])); vb = f4_fma(4.f, x[1], f4_fma(-5.f, x[3], x[5])); ka = 0; kb = 5; }
            else {
                f4p e, o;
                if (grp == 1) { e = f4_fma(-4.f, x[2], x[4]); o = f4_fma(-4.f, x[1], x[3]); }
                else { e = f4_sub(x[4], x[2]); o = f4_mul(2.f, f4_sub(x[3], x[1])); }
                va = f4_add(e, o); vb = f4_sub(e, o); ka = 2 * grp - 1; kb = 2 * grp;
            }
        }
        *reinterpret_cast<f4p*>(Vdst + ((size_t)ka * TT + tl) * 16 + wslot) = va;
        *reinterpret_cast<f4p*>(Vdst + ((size_t)kb * TT + tl) * 16 + wslot) = vb;
    };

    f32x16 acc[NX * 2];
#pragma unroll
    for (int x = 0; x < NX * 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // weights in MFMA operand order: [chunk][xi][cout block of 32][kq][lane][4]
    const int cb = cblk * 2 + wc, ncb = a.Cout / 32;
    const float4* ubase = reinterpret_cast<const float4*>(a.u) + lane;
    const int rrow = wt * 64 + (lane & 31);                              // + 32 for the second tile block (same swizzle: 32 % 16 == 0)
    const int voff0 = rrow * 16 + ((lane >> 5) ^ ((lane >> 2) & 3)) * 4;
    const int voff1 = rrow * 16 + ((2 + (lane >> 5)) ^ ((lane >> 2) & 3)) * 4;

    float4 af[NSTEP];                                                    // one whole chunk of weight fragments in flight
    // VMEM issue schedule of step s (weight fragment s of chunk `uchunk`, then two window loads of the item whose
    // registers the transform has just released).  The prologue replays the SAME order without MFMAs, so the loads
    // in flight at the loop head are ordered identically on both paths into the loop and the compiler's vmcnt
    // counts are exact instead of a conservative minimum.
    auto vmem_step = [&](int s, const float4* uchunk) {
        if (!(ROWS_ABL & 2)) af[s] = uchunk[((size_t)(s >> 1) * ncb * 2 + (s & 1)) * 64];
        if (!(ROWS_ABL & 1) && s >= NG && s < NG + NX) {
            const int l0 = (s - NG) * 2;
            if (l0 == 0) gather_begin(0);
            if (l0 == NX) gather_begin(1);
            gather_load(l0 / NX, l0 % NX);
            gather_load((l0 + 1) / NX, (l0 + 1) % NX);
        }
    };
    {                                                                    // prologue: everything that does not depend on LDS goes out first
        const float4* u0 = ubase + ((size_t)0 * ncb + cb) * 2 * 64;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) af[s] = u0[((size_t)(s >> 1) * ncb * 2 + (s & 1)) * 64];   // weights of chunk 0
    }
    gather();                                                            // chunk 0
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) transform_group(it, grp, V);
    gather();                                                            // chunk 1
    __syncthreads();
    for (int c = 0; c < a.nchunks; ++c) {
        const float* Vc = V + (c & 1) * VBUF;
        float* Vn = V + ((c + 1) & 1) * VBUF;
        const float4* un = ubase + ((size_t)((c + 1 < a.nchunks ? c + 1 : c) * NX) * ncb + cb) * 2 * 64;
        float4 bf0 = *reinterpret_cast<const float4*>(Vc + voff0);
        float4 bf1 = *reinterpret_cast<const float4*>(Vc + 32 * 16 + voff0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int x = s >> 1;
            const float4 aw = af[s];
            const float4 b0 = bf0, b1 = bf1;
            if (s + 1 < NSTEP) {
                const float* vp = Vc + (size_t)((s + 1) >> 1) * TT * 16 + (((s + 1) & 1) ? voff1 : voff0);
                bf0 = *reinterpret_cast<const float4*>(vp);
                bf1 = *reinterpret_cast<const float4*>(vp + 32 * 16);
            }
            acc[2 * x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.x, b0.x, acc[2 * x], 0, 0, 0);
            acc[2 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.x, b1.x, acc[2 * x + 1], 0, 0, 0);
            // in the shadow of the MFMAs: one transform group of chunk c+1 per step (item 0 in steps [0, NG), item 1 in
            // [NG, 2 NG)), then this step's loads: the same fragment slot of chunk c+1 and two windows of chunk c+2
            // (past the last chunk the windows are all out of range = 0 and land in the idle buffer)
            if (!(ROWS_ABL & 4) && s < 2 * NG) transform_group(s / NG, s % NG, Vn);
            vmem_step(s, un);
            acc[2 * x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.y, b0.y, acc[2 * x], 0, 0, 0);
            acc[2 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.y, b1.y, acc[2 * x + 1], 0, 0, 0);
            acc[2 * x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.z, b0.z, acc[2 * x], 0, 0, 0);
            acc[2 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.z, b1.z, acc[2 * x + 1], 0, 0, 0);
            acc[2 * x] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.w, b0.w, acc[2 * x], 0, 0, 0);
            acc[2 * x + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.w, b1.w, acc[2 * x + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                                   // V[c+1] complete, V[c] free for chunk c+2
    }

    // ---- epilogue: y0 = sum_{k<NX-1} M_k, y1 = sum_k AT1[k] M_k; acc row = cout (r&3)+8*(r>>2)+4*(lane>>5), col = tile lane&31
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int to = t0 + wt * 64 + nb * 32 + (lane & 31);
        if (to >= a.T) continue;
        const int oimg = to / THW, orem = to - oimg * THW, oy = orem / a.TW, otx = orem - oy * a.TW;
        const int opix = oy * a.W + 2 * otx;
        const bool two = 2 * otx + 1 < a.W;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = cblk * 64 + wc * 32 + 8 * q + 4 * (lane >> 5);
            const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            float y0[4], y1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const float m = acc[2 * k + nb][r];
                    if (k < NX - 1) s0 += m;
                    if (WM::AT1[k] != 0.f) s1 = fmaf(WM::AT1[k], m, s1);
                }
                y0[e] = s0; y1[e] = s1;
            }
            float4 v0 = make_float4(y0[0] + b.x, y0[1] + b.y, y0[2] + b.z, y0[3] + b.w);
            float4 v1 = make_float4(y1[0] + b.x, y1[1] + b.y, y1[2] + b.z, y1[3] + b.w);
            if (a.relu) {
                v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
            }
            float* op = a.out + c4_offset(oimg, a.Gout_tot, a.gout0 + (co >> 2), HW, opix);
            *reinterpret_cast<float4*>(op) = v0;
            if (two) *reinterpret_cast<float4*>(op + 4) = v1;
        }
    }
}

// U[xi][co][k = 4*(ky*Gin4 + g) + e] = (sum_j G[xi][j] w[co][ci][ky][j]) * BN scale, in MFMA A-operand order
// [chunk][xi][cout/32][kq][lane][4]:  co = cb*32 + (lane&31), k = chunk*16 + kq*8 + 4*(lane>>5) + e.
template <int R>
__global__ void pack_rows_winograd_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                          float eps, int Cout, int Cin, int rot, int nchunks, float* __restrict__ up) {
    constexpr int NX = R + 1;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncb = Cout / 32, Gin4 = (Cin + 3) / 4;
    const long long total = (long long)nchunks * NX * ncb * 2 * 64 * 4;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), kq = (int)((idx >> 8) & 1);
    long long r = idx >> 9;
    const int cb = (int)(r % ncb); r /= ncb;
    const int xi = (int)(r % NX), chunk = (int)(r / NX);
    const int co = cb * 32 + (lane & 31);
    const int q = chunk * 4 + kq * 2 + (lane >> 5);
    const int ky = q / Gin4, cp = 4 * (q - ky * Gin4) + e;
    float v = 0.f;
    if (ky < R && cp < Cin) {
        const int ci = (cp + rot) % Cin;
        const float* g = w + (((size_t)co * Cin + ci) * R + ky) * R;
        double s = 0;
        for (int j = 0; j < R; ++j) s += RowWino<R>::G[xi][j] * (double)g[j];
        if (gamma) s *= (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)s;
    }
    up[idx] = v;
}

static int rows_chunks(int Cin, int ksize) { return (ksize * ((Cin + 3) / 4) + 3) / 4; }

extern "C" size_t cnm_packed_winograd_rows_floats(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || (ksize != 5 && ksize != 7)) return 0;
    return (size_t)rows_chunks(Cin, ksize) * (ksize + 1) * Cout * 16;
}

extern "C" int cnm_pack_winograd_rows_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                             int Cout, int Cin, int ksize, int rot, float* u_packed, void* stream) {
    CNM_REQUIRE(w_oihw && u_packed && Cout > 0 && Cout % 64 == 0 && Cin > 0 && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((ksize == 5 || ksize == 7) && !bn_gamma == !bn_var, CNM_ERR_BAD_ARG);
    const int nchunks = rows_chunks(Cin, ksize);
    const long long total = (long long)nchunks * (ksize + 1) * Cout * 16;
    const unsigned nb = (unsigned)cnm_ceil_div_ll(total, 256);
    if (ksize == 5) pack_rows_winograd_kernel<5><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed);
    else pack_rows_winograd_kernel<7><<<nb, 256, 0, cnm_stream(stream)>>>(w_oihw, bn_gamma, bn_var, eps, Cout, Cin, rot, nchunks, u_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv_rows_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                             const float* in_b, int Gb_total, int gb0, int Gb,
                                             float* out, int Gout_total, int gout0, int Cout,
                                             const float* u_packed, const float* b_packed,
                                             int N, int H, int W, int ksize, int relu, void* stream) {
    CNM_REQUIRE(in_a && out && u_packed && N > 0 && H > 0 && W > 0 && Ga > 0 && Gb >= 0 && (ksize == 5 || ksize == 7), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 64 == 0 && gout0 >= 0 && gout0 + Cout / 4 <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ga0 >= 0 && ga0 + Ga <= Ga_total && (Gb == 0 || (in_b && gb0 >= 0 && gb0 + Gb <= Gb_total)), CNM_ERR_BAD_ARG);
    RowArgs a;
    a.in = in_a; a.in2 = Gb ? in_b : in_a; a.out = out; a.u = u_packed; a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Ga_total * H * W * 16ull;
    const unsigned long long b2 = Gb ? (unsigned long long)N * Gb_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    a.N = N; a.H = H; a.W = W; a.TW = (W + 1) / 2;
    a.Gin_tot = Ga_total; a.gin0 = ga0; a.Gin2_tot = Gb ? Gb_total : Ga_total; a.gin2_0 = Gb ? gb0 : ga0; a.Gsplit = Ga; a.Gin = Ga + Gb;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout;
    a.nchunks = (ksize * a.Gin + 3) / 4; a.T = N * H * a.TW; a.relu = relu;
    const int nblocks = (Cout / 64) * cnm_ceil_div(a.T, 128);
    if (ksize == 5) conv_rows_winograd_f32_kernel<5><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    else conv_rows_winograd_f32_kernel<7><<<nblocks, 256, 0, cnm_stream(stream)>>>(a);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
