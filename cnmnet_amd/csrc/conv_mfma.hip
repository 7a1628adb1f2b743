// K2: Conv2d(+folded BatchNorm)(+ReLU) as an fp32 MFMA implicit GEMM on c4 activations.
//
// Replaces the nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU chains built by
// down_conv_layer / conv_layer / up_conv_layer (reference depthnet/depthNet_model.py:19-112)
// in eval mode.  GEMM view:  D[cout][pixel] = sum_k Wp[cout][k] * X[k][pixel],
//   k = (ky*ks + kx) * 4*Gin + c,  X[k][pixel] = in[img][c/4][oy*s+ky-p][ox*s+kx-p][c%4] (0 outside).
// MFMA: v_mfma_f32_32x32x2_f32, A operand = weights (rows = cout), B operand = pixels, so
// every lane ends up holding 4 consecutive output channels (= one c4 group) of one pixel and
// consecutive lanes hold consecutive pixels: the epilogue is a coalesced float4 store.
// Numerics: exact fp32 FMA chains (the MFMA is bitwise an fmaf chain), no reduced precision.
//
// Tiling (64-lane wavefronts): block = 4 waves (2x2), block tile TC couts x TP pixels x 16 k,
// wave tile (TC/2)x(TP/2) built from 32x32 MFMA tiles; LDS double buffered, rows padded to
// 20 floats so the ds_read_b128 fragment reads are bank-conflict free; global->register
// prefetch of k-step t+1 overlaps the 32..128 MFMAs of k-step t.
#include "cnm_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CONV_BK
#define CONV_BK 16            // k-depth of one LDS tile (16 or 32)
#endif
#ifndef GLDS_PINGPONG
#define GLDS_PINGPONG 0       // conv_glds_kernel: two wave groups half a k-step apart (experiment; see the kernel)
#endif
#ifndef CONV_MINW
#define CONV_MINW 1           // __launch_bounds__ min waves per SIMD (register cap)
#endif

struct ConvArgs {
    const float* in; float* out; const float* w; const float* bias;
    const float* in2;            // optional second source: channel groups [Gsplit, Gin) (torch.cat without the copy)
    unsigned in_bytes, in2_bytes; // extents for the buffer descriptors (< 4 GB each)
    unsigned w_bytes;             // packed filter bytes (LDS-DMA kernel: k-steps past the end read zeros)
    int Gin2_tot, gin2_0, Gsplit;
    int N, H, W, Ho, Wo;
    int Gin_tot, gin0, Gin;
    int Gout_tot, gout0, Cout;   // Cout: real output channels (multiple of 4); weights/tiles use Cout_pad
    int Cout_pad;                // Cout rounded up to 64
    int ks, stride, pad;
    int nk;          // Kpad / CONV_BK
    int M;           // N*Ho*Wo
    int relu;
    int ring;        // fused upsampling (conv_glds_kernel<..., UPS>): 1 = leave the one-pixel output ring without bias / ReLU for the ring pass
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: an out-of-range byte offset (we pass 0xFFFFFFFF for the
// zero-padding taps and the tile's pixel tail) returns zeros in hardware -- the im2col gather needs
// no branches and no selects, so the whole k-step stays one basic block the scheduler can interleave
// with the MFMAs.
__device__ __forceinline__ float4 buffer_load_f4(const float* base, unsigned bytes, unsigned voff, unsigned soff) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// TS = 1: ordinary convolution (any stride, via a.stride).  TS = 2: transposed stride-2 gather used by the
// data gradient of a stride-2 layer: tap (a,b) of output pixel (y,x) reads source pixel ((y+a-pad)/2, (x+b-pad)/2)
// when both are even and in range (the zero-upsampled view of dy, never materialised).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int TC, int TP, int TS = 1>
__global__ __launch_bounds__(256, CONV_MINW) void conv_mfma_f32_kernel(const ConvArgs a) {
    constexpr int NT = 256, WP = 2;
    constexpr int CI = TC / 64, PI = TP / 64;          // 32x32 MFMA tiles per wave
    constexpr int BK = CONV_BK, KQ = BK / 4;           // k-quads (float4) per LDS row
    constexpr int LDK = BK + 4;                        // +4 pad floats per LDS row: conflict-free ds_read_b128
    constexpr int A_LOADS = TC * KQ / NT, B_LOADS = TP * KQ / NT, QSTEP = NT / TP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (TC + TP) * LDK];
    float* As = smem;                                  // [2][TC][LDK]
    float* Bs = smem + 2 * TC * LDK;                   // [2][TP][LDK]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int tilesC = a.Cout_pad / TC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int c0 = (tile % tilesC) * TC, m0 = (tile / tilesC) * TP;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    // ---- pixel (B operand) loader state: one output pixel per thread
    const int prow = t % TP;
    const int m = m0 + prow;
    const bool mvalid = m < a.M;
    int iy0, ix0;
    unsigned vb1, vb2;                                  // byte offset of tap (0,0), group 0 in source 1 / 2 (mod 2^32)
    {
        const int mm = mvalid ? m : 0;
        const int img = mm / HoWo, rem = mm - img * HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0 = oy * a.stride - a.pad; ix0 = ox * a.stride - a.pad;
        const unsigned pix0 = TS == 1 ? (unsigned)(iy0 * a.W + ix0) : 0u;
        vb1 = ((unsigned)(img * a.Gin_tot + a.gin0) * (unsigned)HW + pix0) * 16u;
        vb2 = ((unsigned)(img * a.Gin2_tot + a.gin2_0) * (unsigned)HW + pix0) * 16u;
    }
    // (tap, channel-group) of the k-quads this wave loads is wave-uniform: SGPR state, SALU updates
    const int wq = __builtin_amdgcn_readfirstlane(t / TP);
    int bg[B_LOADS], bky[B_LOADS], bkx[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int kq = wq + i * QSTEP;                 // k-quad index within the k-step
        const int tap = kq / a.Gin;
        bg[i] = kq - tap * a.Gin; bky[i] = tap / a.ks; bkx[i] = tap - bky[i] * a.ks;
    }
    const float* wtile = a.w + (size_t)c0 * BK;        // [nk][Cout][BK]
    const unsigned HW16 = (unsigned)HW * 16u;

    float4 ra[A_LOADS], rb[B_LOADS];
#define CONV_LOAD_GLOBAL(KT)                                                                                       \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                                        \
            ra[i] = *reinterpret_cast<const float4*>(wtile + (size_t)(KT) * a.Cout_pad * BK + (size_t)(t + i * NT) * 4); \
        _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) {                                                      \
            const int iy = iy0 + bky[i], ix = ix0 + bkx[i];                                                        \
            bool ok = mvalid && bky[i] < a.ks;                                                                      \
            unsigned tapoff;                                                                                       \
            if (TS == 1) {                                                                                         \
                ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;                           \
                tapoff = (unsigned)(bky[i] * a.W + bkx[i]) * 16u;                                                  \
            } else {                                                                                               \
                ok = ok && !((iy | ix) & 1) && (unsigned)(iy >> 1) < (unsigned)a.H && (unsigned)(ix >> 1) < (unsigned)a.W; \
                tapoff = (unsigned)((iy >> 1) * a.W + (ix >> 1)) * 16u;                                            \
            }                                                                                                      \
            const bool s1 = bg[i] < a.Gsplit;                                            /* wave-uniform */         \
            const unsigned voff = ok ? (s1 ? vb1 : vb2) + tapoff : 0xFFFFFFFFu;                                     \
            rb[i] = buffer_load_f4(s1 ? a.in : a.in2, s1 ? a.in_bytes : a.in2_bytes, voff,                          \
                                   (unsigned)(s1 ? bg[i] : bg[i] - a.Gsplit) * HW16);                              \
            bg[i] += KQ;                                                 /* advance BK floats of flat k */           \
            const bool wrap = bg[i] >= a.Gin;                            /* Gin >= KQ: at most one wrap */           \
            bg[i] -= wrap ? a.Gin : 0; bkx[i] += wrap ? 1 : 0;                                                      \
            const bool wrapx = bkx[i] == a.ks;                                                                      \
            bkx[i] = wrapx ? 0 : bkx[i]; bky[i] += wrapx ? 1 : 0;                                                   \
        }                                                                                                          \
    } while (0)
#define CONV_STORE_LDS(BUF)                                                                                        \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                                      \
            const int f = t + i * NT;                                                                              \
            *reinterpret_cast<float4*>(As + ((size_t)(BUF) * TC + f / KQ) * LDK + (f % KQ) * 4) = ra[i];           \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                                        \
            *reinterpret_cast<float4*>(Bs + ((size_t)(BUF) * TP + prow) * LDK + (wq + i * QSTEP) * 4) = rb[i];     \
    } while (0)

    f32x16 acc[CI][PI];
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int j = 0; j < PI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 4;
#define CONV_COMPUTE(BUF)                                                                                          \
    do {                                                                                                           \
        const float* Ab = As + ((size_t)(BUF) * TC + wc * (CI * 32) + frow) * LDK + fk;                            \
        const float* Bb = Bs + ((size_t)(BUF) * TP + wp * (PI * 32) + frow) * LDK + fk;                            \
        _Pragma("unroll") for (int kg = 0; kg < BK / 8; ++kg) {                                                    \
            float4 af[CI], bf[PI];                                                                                 \
            _Pragma("unroll") for (int i = 0; i < CI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + kg * 8); \
            _Pragma("unroll") for (int j = 0; j < PI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + kg * 8); \
            _Pragma("unroll") for (int i = 0; i < CI; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < PI; ++j) {                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);        \
                }                                                                                                  \
        }                                                                                                          \
    } while (0)

    CONV_LOAD_GLOBAL(0);
    CONV_STORE_LDS(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < a.nk; ++kt) {            // steady state: one basic block per k-step
        const int buf = kt & 1;
        CONV_LOAD_GLOBAL(kt + 1);                      // issue first: a whole k-step of MFMAs hides the latency
        CONV_COMPUTE(buf);
        CONV_STORE_LDS(buf ^ 1);
        __syncthreads();
    }
    CONV_COMPUTE((a.nk - 1) & 1);
#undef CONV_LOAD_GLOBAL
#undef CONV_STORE_LDS
#undef CONV_COMPUTE

    // ---- epilogue: acc row = cout (r&3)+8*(r>>2)+4*(lane>>5), col = pixel lane&31
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int mm = m0 + (wp * PI + j) * 32 + (lane & 31);
        if (mm >= a.M) continue;
        const int img = mm / HoWo, pix = mm - img * HoWo;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int c = c0 + (wc * CI + i) * 32 + 8 * qd + 4 * (lane >> 5);
                if (c >= a.Cout) continue;
                const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v = make_float4(acc[i][j][4 * qd + 0] + b.x, acc[i][j][4 * qd + 1] + b.y,
                                       acc[i][j][4 * qd + 2] + b.z, acc[i][j][4 * qd + 3] + b.w);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(a.out + c4_offset(img, a.Gout_tot, a.gout0 + (c >> 2), HoWo, pix)) = v;
            }
        }
    }
}

// ------------------------------------------------------------------ fp16 path: the order of the reduction [r5]
// A k-step is 64 halfs = 8 slots of one 16-byte (8-channel) group each.  G = channel groups of the input, NGB = G / 8 full
// blocks of 8 groups, r = G % 8 left-over groups.  k-steps run
//   main:      for ky: for gb < NGB: for kx:   slot w = group 8 gb + w at tap (ky, kx)
//   left-over: for ky: for c < spk:            slot w = group 8 NGB + w % r at tap (ky, c tps + w / r),  tps = 8 / r taps per step,
//                                              spk = ceil(ks / tps) steps per filter row (slots past the row's taps: zeros)
// i.e. kx is the FASTEST index of a group block: the ks k-steps of one (ky, gb) read the SAME input rows shifted by one pixel
// per step -- conv_gldsx_kernel stages that row block once and shifts its fragment addresses, conv_glds_kernel (stride 2, fused
// upsampling, images narrower than 32 pixels) walks the same order tap by tap.  Packed filters follow it: [k-step][8 slots][Cout_pad][8 halfs].
struct F16Walk {
    int ks, NGB, r, tps, spk, nmain, nk;
    __host__ __device__ F16Walk(int G, int ks_) : ks(ks_), NGB(G >> 3), r(G & 7) {
        tps = r ? 8 / r : 0; spk = r ? (ks + tps - 1) / tps : 0;
        nmain = ks * NGB * ks; nk = nmain + ks * spk;
    }
    // slot w of k-step kt -> tap (ky, kx), group g; false: the slot is padding (zero filter rows, no pixels)
    __host__ __device__ bool slot(int kt, int w, int& ky, int& kx, int& g) const {
        if (kt < nmain) { ky = kt / (NGB * ks); const int rem = kt - ky * NGB * ks, gb = rem / ks; kx = rem - gb * ks; g = 8 * gb + w; return true; }
        const int q = kt - nmain; ky = q / spk; const int c = q - ky * spk, t = w / r;
        kx = c * tps + t; g = 8 * NGB + w % r;
        return t < tps && kx < ks && ky < ks;
    }
};

// The same walk one k-step at a time, for a kernel that issues its k-steps in order: slot() costs two divisions per call, and in the
// launches whose k-steps are latency-bound (the 12 x 16 / 6 x 8 levels: 4 MFMAs per wave and barrier) that scalar arithmetic was a
// tenth of a k-step (40 -> 37 us per layer, same box).
struct F16WalkIter {
    int ks, NGB, spk, ky, gb, kx, c; bool left;
    __device__ explicit F16WalkIter(const F16Walk& w) : ks(w.ks), NGB(w.NGB), spk(w.spk), ky(0), gb(0), kx(0), c(0), left(w.NGB == 0) {}
    __device__ void next() {
        if (!left) { if (++kx == ks) { kx = 0; if (++gb == NGB) { gb = 0; if (++ky == ks) { ky = 0; left = true; } } } }
        else if (++c == spk) { c = 0; ++ky; }
    }
};

// ------------------------------------------------------------------ fp16 implicit GEMM fed by LDS-DMA
// The fp16 matrix pipe is 16x faster than the fp32 one, so the loader above (global -> registers -> ds_write_b128, one
// barrier per 64 bytes of depth) would be LDS-store bound at ~30 % of the pipe.  This kernel moves both operands with
// buffer_load_dwordx4 ... lds (no VGPRs, no ds_write; an out-of-range offset -- zero padding, pixel tail, ragged K --
// lands as zeros in LDS: tools/glds_probe.hip), 64 halfs of depth per barrier, 8 waves of WTC couts x WTP pixels each:
//   LDS image of a k-step, per operand: [8 channel groups][rows][16 B]  (rows = couts / pixels)
//     - a DMA piece is one group x 64 consecutive rows = 1 KB, lane-linear, and its (tap, channel group) is wave-uniform:
//       wave w moves group w of every k-step (all its A and B pieces), the tap walk lives in SGPRs;
//     - the MFMA fragment of lane (row = l & 31, k-half = l >> 5) for the k16 step s is the 16 bytes at
//       [(2 s + k-half)][row]: 32 consecutive rows per half wave, conflict-free ds_read_b128.  The reads are typed
//       f16x8 on purpose: with float4 reads hipcc orders every first ds_read of a k-step behind s_waitcnt vmcnt(0), i.e.
//       behind the DMA just issued for the NEXT steps (measured: 25-30 % slower).
//   Weights are packed in exactly that order ([k-step][8 groups][Cout_pad][8 halfs]), so A pieces are contiguous.
// Three LDS buffers when they fit: the DMA of steps t+1 and t+2 is in flight under the MFMAs of step t, one of them
// across the barrier (counted s_waitcnt vmcnt + raw s_barrier).  Tiles: 128 x 256 and 64 x 512 (waves of 64 x 64), the
// larger 256 x 256 (waves of 128 x 64) and 128 x 512 (64 x 128) where they fill whole rounds of workgroups, and 64 x 128
// (waves of 32 x 32, two workgroups per CU) where the others would leave CUs without a workgroup.
// (The same kernel on fp32 data -- four v_mfma_f32_32x32x2_f32 per fragment pair -- was measured on the fp32 engine's
// implicit-GEMM layers: bit-identical, +21 % on conv4.3, -8 ... -20 % on the larger stride-2 layers, which are
// matrix-pipe bound with the register-staged loader already; not kept.)
// UPS: the fp32 engine's fused up_conv (conv_winograd4.hip) on this kernel -- a 3x3 convolution of the LOW-resolution input
// with the four composed phase filters as 4 x Cr "virtual" output channels (phase major), window samples outside the image
// clamped (replicate padding of the composition), the epilogue storing virtual channel (2a+b) Cr + c of low-resolution
// pixel (y, x) to channel c of (2y+a, 2x+b); the one-pixel output ring is finished by cnm_conv3x3_upsampled_ring_c8_f16.
template <int TC, int TP, int WTC, int WTP, bool UPS = false>
__global__ __launch_bounds__(512) void conv_glds_kernel(const ConvArgs a) {
    constexpr int WC = TC / WTC, WP = TP / WTP, CI = WTC / 32, PI = WTP / 32;
    static_assert(WC * WP == 8 && TC % 64 == 0 && TP % 64 == 0 && WTC % 32 == 0 && WTP % 32 == 0, "8 waves");
    constexpr int AB = 8 * TC * 16, BB = 8 * TP * 16, BUF = AB + BB;   // bytes per operand image / per buffer
    constexpr int NPA = TC / 64, NPB = TP / 64;                         // DMA pieces of one group
    constexpr int NBUF = 3 * BUF <= 160 * 1024 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF];      // 144 KB: 3 x 48 KB (128 x 256) / 2 x 72 KB (64 x 512); 72 KB for 64 x 128; 2 x 64 KB (256 x 256), 2 x 80 KB (128 x 512)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int tilesC = a.Cout_pad / TC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int c0 = (tile % tilesC) * TC, m0 = (tile / tilesC) * TP;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    // ---- B pieces: lane = pixel m0 + 64 pb + lane of piece pb
    int iy0[NPB], ix0[NPB]; unsigned vb1[NPB], vb2[NPB]; bool mval[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int m = m0 + 64 * pb + lane;
        mval[pb] = m < a.M;
        const int mm = mval[pb] ? m : 0;
        const int img = mm / HoWo, rem = mm - img * HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[pb] = oy * a.stride - a.pad; ix0[pb] = ox * a.stride - a.pad;
        const unsigned pix0 = UPS ? 0u : (unsigned)(iy0[pb] * a.W + ix0[pb]);   // UPS: the clamped tap position is added per k-step
        vb1[pb] = ((unsigned)(img * a.Gin_tot + a.gin0) * (unsigned)HW + pix0) * 16u;
        vb2[pb] = ((unsigned)(img * a.Gin2_tot + a.gin2_0) * (unsigned)HW + pix0) * 16u;
    }
    // (tap, channel group) of this wave's slot in a k-step: the walk above, scalar arithmetic per issue
    const F16Walk walkk(a.Gin, a.ks);
    F16WalkIter wit(walkk);                                             // issue() is called for k-steps 0, 1, 2, ... in order
    const int wtl = walkk.r ? wave / walkk.r : 0, wgl = 8 * walkk.NGB + (walkk.r ? wave % walkk.r : 0);   // this wave's slot in a left-over step: tap within the step, group
    const unsigned HW16 = (unsigned)HW * 16u;
    // filter [k-step][8 groups][Cout_pad][16 B]: a piece = 64 consecutive rows of one group
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
    const unsigned wlane = ((unsigned)c0 + (unsigned)lane) * 16u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
#ifdef GLDS_PROBE
    int probe_last = 0;                                                 // DMA instructions of this wave's latest issue()
#endif
    auto issue = [&](int kt, int buf) {                                 // DMA of k-step kt into buffer buf (this wave: group `wave`)
        char* A = smem + buf * BUF + wave * TC * 16;
        char* B = smem + buf * BUF + AB + wave * TP * 16;
        const unsigned wsoff = (unsigned)(kt * 8 + wave) * (unsigned)a.Cout_pad * 16u;
#ifdef GLDS_PROBE
        const auto wrsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, ((GLDS_PROBE & 2) && kt > 0) ? 0u : a.w_bytes, 0x00020000);
#define wrsrc wrsrc_p
#endif
#pragma unroll
        for (int p = 0; p < NPA; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(A + p * 1024), 16, wlane + p * 1024u, wsoff, 0, 0);
#ifdef GLDS_PROBE
#undef wrsrc
#endif
        int g = 8 * wit.gb + wave, ky = wit.ky, kx = wit.kx;            // = walkk.slot(kt, wave, ky, kx, g)
        bool tapok = true;
        if (wit.left) { kx = wit.c * walkk.tps + wtl; g = wgl; tapok = (wtl < walkk.tps) & (kx < a.ks) & (ky < a.ks); }
        wit.next();
        const bool s1 = g < a.Gsplit;
#ifdef GLDS_PROBE   // traffic experiment (tools/f16_traffic_probe.sh; wrong results): 1 = pixel pieces fetched for the first tap only, 2 = filter pieces for the first k-step only, 3 = both
        if ((GLDS_PROBE & 1) && (ky | kx)) tapok = false;
#endif
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 ? a.in : a.in2), 0, tapok ? (s1 ? a.in_bytes : a.in2_bytes) : 0u, 0x00020000);
        const unsigned soff = (unsigned)(s1 ? g : g - a.Gsplit) * HW16, tapoff = (unsigned)(ky * a.W + kx) * 16u;
#ifdef GLDS_PROBE
        probe_last = NPA + NPB;
        if ((GLDS_PROBE & 4) && (ky | kx)) probe_last = NPA;            // 4: NO pixel-piece instructions past the first tap (the DMA issue count of a halo-staged kernel)
        if (probe_last == NPA + NPB)
#endif
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int iy = iy0[pb] + ky, ix = ix0[pb] + kx;
            unsigned voff;
            if constexpr (UPS) {
                const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
                voff = mval[pb] ? vb1[pb] + (unsigned)(cy * a.W + cx) * 16u : 0xFFFFFFFFu;
            } else {
                const bool ok = mval[pb] & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
                voff = ok ? (s1 ? vb1[pb] : vb2[pb]) + tapoff : 0xFFFFFFFFu;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(B + pb * 1024), 16, voff, soff, 0, 0);
        }
    };

    f32x16 acc[CI][PI];
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int j = 0; j < PI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, kh = lane >> 5;
    auto compute = [&](int buf) {
        const char* Ab = smem + buf * BUF + (kh * TC + wc * WTC + frow) * 16;
        const char* Bb = smem + buf * BUF + AB + (kh * TP + wp * WTP + frow) * 16;
        f16x8 af[2][CI], bf[2][PI];                                     // fragments of group pair s + 1 are read under the MFMAs of pair s
#pragma unroll
        for (int i = 0; i < CI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(Ab + 32 * i * 16);
#pragma unroll
        for (int j = 0; j < PI; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(Bb + 32 * j * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int i = 0; i < CI; ++i) af[(s + 1) & 1][i] = *reinterpret_cast<const f16x8*>(Ab + (2 * (s + 1) * TC + 32 * i) * 16);
#pragma unroll
                for (int j = 0; j < PI; ++j) bf[(s + 1) & 1][j] = *reinterpret_cast<const f16x8*>(Bb + (2 * (s + 1) * TP + 32 * j) * 16);
            }
#pragma unroll
            for (int i = 0; i < CI; ++i)
#pragma unroll
                for (int j = 0; j < PI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
        }
        // pin the order: the fragment reads of pair s + 1 are ISSUED before the MFMAs of pair s (left alone, the scheduler
        // reuses one register set and every step waits out the LDS latency with an idle matrix pipe)
        constexpr int NR = CI + PI, NM = CI * PI;
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
        for (int s = 0; s < 3; ++s) { __builtin_amdgcn_sched_group_barrier(0x100, NR, 0); __builtin_amdgcn_sched_group_barrier(0x008, NM, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    };

    // NBUF LDS buffers: the DMA of k-step kt + NBUF - 1 is issued before the MFMAs of step kt; with three buffers one
    // k-step stays in flight across the barrier (counted vmcnt: NPW DMA instructions per wave and step, in order)
    constexpr int NPW = NPA + NPB;
    auto wait_dma = [&](bool one_in_flight) {
        if (NBUF == 3 && one_in_flight) {
#ifdef GLDS_PROBE
            if (probe_last == NPA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPA) : "memory"); else
#endif
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence would drain the DMA left in flight
    };
    issue(0, 0);
    if (NBUF == 3 && a.nk > 1) issue(1, 1);
    wait_dma(a.nk > 1);
    int cur = 0, nxt = NBUF - 1;                                        // buffer of step kt / of the step issued in iteration kt
#if GLDS_PINGPONG
    // Two wave groups one segment apart (waves 0-3 / 4-7: one wave of each on every SIMD).  A k-step is 4 / SK sub-steps of SK k16 slices
    // (16 MFMAs per wave); per sub-step a group has a memory segment (fragment reads of the sub-step, and the step's DMA in the first
    // one) and a matrix segment (its MFMAs out of registers), and the groups swap roles at every barrier -- every SIMD has one wave in a
    // matrix segment and one in a memory segment instead of two waves in the same phase.
    constexpr bool PP = CI * PI >= 4;
    constexpr int SK = CI * PI >= 8 ? 2 : 4, NS = 4 / SK;
    f16x8 af[SK][CI], bf[SK][PI];
    auto load_frags = [&](int buf, int ss) {
        const char* Ab = smem + buf * BUF + (kh * TC + wc * WTC + frow) * 16;
        const char* Bb = smem + buf * BUF + AB + (kh * TP + wp * WTP + frow) * 16;
#pragma unroll
        for (int s = 0; s < SK; ++s) {
#pragma unroll
            for (int i = 0; i < CI; ++i) af[s][i] = *reinterpret_cast<const f16x8*>(Ab + (2 * (ss * SK + s) * TC + 32 * i) * 16);
#pragma unroll
            for (int j = 0; j < PI; ++j) bf[s][j] = *reinterpret_cast<const f16x8*>(Bb + (2 * (ss * SK + s) * TP + 32 * j) * 16);
        }
    };
    auto mfma_cluster = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < SK; ++s)
#pragma unroll
            for (int i = 0; i < CI; ++i)
#pragma unroll
                for (int j = 0; j < PI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (PP) {
        // both groups run the SAME code; waves 4-7 run it one barrier late (and waves 0-3 pass one more at the end), so the accumulators
        // live in one set of registers.  Buffer of step kt: last read in the trailing group's memory segment of the last sub-step, whose
        // barrier is the leading group's last matrix barrier of the step -- only behind it does the leading group issue DMA into that
        // buffer again.  Both barriers of the last sub-step carry the DMA wait (one of them is the "data of step kt + 1 has landed"
        // barrier for either group).
        const bool lead = wave < 4;
        if (!lead) asm volatile("s_barrier" ::: "memory");
        for (int kt = 0; kt < a.nk; ++kt) {
#pragma unroll
            for (int ss = 0; ss < NS; ++ss) {
                if (ss == 0 && kt + NBUF - 1 < a.nk) issue(kt + NBUF - 1, nxt);
                load_frags(cur, ss);
                if (ss == NS - 1) wait_dma(kt + 2 < a.nk);
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                mfma_cluster();
                if (ss == NS - 1) wait_dma(kt + 2 < a.nk);
                else asm volatile("s_barrier" ::: "memory");
            }
            cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
        }
        if (lead) asm volatile("s_barrier" ::: "memory");
    } else
#endif
    {
    for (int kt = 0; kt < a.nk; ++kt) {
        if (kt + NBUF - 1 < a.nk) issue(kt + NBUF - 1, nxt);
        compute(cur);
        wait_dma(kt + 2 < a.nk);
        cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }
    }

    // ---- epilogue: acc row = cout (r&3)+8*(r>>2)+4*(lane>>5), col = pixel lane&31
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int mm = m0 + wp * WTP + j * 32 + (lane & 31);
        if (mm >= a.M) continue;
        const int img = mm / HoWo, pix = mm - img * HoWo;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int c = c0 + wc * WTC + i * 32 + 8 * qd + 4 * (lane >> 5);
                if (c >= a.Cout) continue;
                const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v = make_float4(acc[i][j][4 * qd + 0], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
                int ogrp = c >> 3, opix = pix, oHW = HoWo, coff = (c & 4) * 2;
                bool finish = true;
                if constexpr (UPS) {                                    // virtual channel -> (phase, real channel); low-resolution pixel -> its phase pixel
                    const int Cr = a.Cout >> 2, ph = c / Cr, cr = c - ph * Cr, y = pix / a.W, x = pix - y * a.W;
                    const int Y = 2 * y + (ph >> 1), X = 2 * x + (ph & 1), Ho2 = 2 * a.H, Wo2 = 2 * a.W;
                    ogrp = cr >> 3; coff = (cr & 4) * 2; opix = Y * Wo2 + X; oHW = 4 * HoWo;
                    finish = !(a.ring && ((Y == 0) | (Y == Ho2 - 1) | (X == 0) | (X == Wo2 - 1)));   // ring pixels: bias and ReLU belong to the ring pass
                }
                if (finish) {
                    v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};   // c8: channels c..c+3 = half of the 16-byte group c/8
                char* o = reinterpret_cast<char*>(a.out + c4_offset(img, a.Gout_tot, a.gout0 + ogrp, oHW, opix)) + coff;
                *reinterpret_cast<f16x4*>(o) = h;
            }
        }
    }
}

// ------------------------------------------------------------------ fp16 implicit GEMM, row-extended pixel operand [r5]
// conv_glds_kernel fetches the pixel operand of EVERY k-step from L2 -- for a k x k filter the same input rows k times per
// filter row, shifted by one pixel.  profiles/r4_f16_traffic_probe.txt measured what that costs: with the pixel pieces of
// all taps but the first not fetched, the 7x7 / 5x5 layers run 22-30 % faster, the 3x3 layers 3-9 %.  This kernel does not
// fetch them: a tile is TP consecutive output pixels in raster order (whole image rows, or an aligned fraction of one), and for
// one filter row ky and one block of 8 channel groups the input rows it needs are staged ONCE as a row-extended image
//   [8 groups][segment = tile row][seglen + 8 slots of 16 B]   (slot j of a segment = input pixel x0 - pad + j; out-of-image = DMA zeros)
// which the ks k-steps (ky, gb, kx = 0 .. ks-1) of the walk above all read -- the MFMA B fragment of k-step kx is the same
// 32-consecutive-rows ds_read_b128 as in conv_glds_kernel, its base moved by kx * 16 bytes.  Filters arrive per k-step as
// before (three LDS buffers); the pixel images are double-buffered per (ky, gb) block and in flight a whole block early.
// A single left-over group (Cin = 8 n + 1 .. 8: the cost volume's RGB group, the decoder's disparity channel) takes one
// k-step per filter row: its 8 slots are the taps kx = 0 .. 7 of that ONE group, i.e. the same one-group image read at
// plane pitch 16 bytes -- slot s is the image moved by s pixels.
// Scope: stride 1, W a multiple of 32 and (H W) a multiple of TP, at most one left-over group; everything else (stride 2,
// the fused up_conv, the 12 x 16 and 6 x 8 levels) stays on conv_glds_kernel, which walks the same filter order.
template <int TC, int TP, int WTC, int WTP>
__global__ __launch_bounds__(512) void conv_gldsx_kernel(const ConvArgs a) {
    constexpr int WC = TC / WTC, WP = TP / WTP, CI = WTC / 32, PI = WTP / 32;
    static_assert(WC * WP == 8 && TC % 64 == 0 && TP % 64 == 0 && WTC % 32 == 0 && WTP % 32 == 0, "8 waves");
    constexpr int AB = 8 * TC * 16, NPA = TC / 64;                      // filter image of a k-step; its DMA pieces per group
    constexpr int NSL = TP + 64, NPX = NSL / 64;                        // slots of one group's row-extended image (8 halo slots per segment, <= 8 segments); its DMA pieces
    constexpr int BXG = NSL * 16, BX = 8 * BXG;                         // bytes of one group's image / of a block of 8 groups
    constexpr int NBUF = (3 * AB + 2 * BX + 2 * BXG) <= 160 * 1024 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * AB + 2 * BX + 2 * BXG];   // 128 x 256: 48 + 80 + 10 KB; 256 x 256: 64 + 80 + 10 KB
    char* const Abuf = smem;
    char* const Xbuf = smem + NBUF * AB;
    char* const Lbuf = smem + NBUF * AB + 2 * BX;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int tilesC = a.Cout_pad / TC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int c0 = (tile % tilesC) * TC, m0 = (tile / tilesC) * TP;
    const int HW = a.H * a.W, W = a.W, H = a.H;
    const F16Walk walkk(a.Gin, a.ks);
    const int ks = a.ks, pad = a.pad, NGB = walkk.NGB, nmain = walkk.nmain, nk = walkk.nk;

    // ---- the tile: TP consecutive pixels of ONE image (HW % TP == 0), segments = image rows (or one aligned row fraction)
    const int img = m0 / HW, rem0 = m0 - img * HW, oy0 = rem0 / W, ox0 = rem0 - oy0 * W;
    const int seglen = TP < W ? TP : W, nseg = TP / seglen, SEGW = seglen + 8;
    const int lgseg = 31 - __builtin_clz(seglen);                       // seglen is a power of two (W % 32 == 0 and TP | W or W | TP, checked by the launcher)
    // staging: lane = slot 64 p + lane of piece p of this wave's group
    int vbx[NPX], ybx[NPX]; bool xok[NPX];   // (vbx as int: with an unsigned array read inside the staging lambda hipcc silently emits no host stub for this template)
    const unsigned base1 = (unsigned)(img * a.Gin_tot + a.gin0) * (unsigned)HW, base2 = (unsigned)(img * a.Gin2_tot + a.gin2_0) * (unsigned)HW;
    {
        const float inv_segw = 1.0f / (float)SEGW;
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            const int slot = 64 * p + lane;
            const int seg = (int)(((float)slot + 0.5f) * inv_segw), j = slot - seg * SEGW;
            const int x = ox0 + j - pad;
            ybx[p] = oy0 + seg;
            xok[p] = seg < nseg && (unsigned)x < (unsigned)W;
            vbx[p] = (ybx[p] * W + x) * 16;
        }
    }
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
    const unsigned wlane = ((unsigned)c0 + (unsigned)lane) * 16u;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    auto issue_a = [&](int kt, int buf) {                               // filters of k-step kt (this wave: slot `wave`)
        char* A = Abuf + buf * AB + wave * TC * 16;
        const unsigned wsoff = (unsigned)(kt * 8 + wave) * (unsigned)a.Cout_pad * 16u;
#pragma unroll
        for (int p = 0; p < NPA; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(A + p * 1024), 16, wlane + p * 1024u, wsoff, 0, 0);
    };
    // one group's row-extended image for filter row ky: pieces [p0, p1) into `dst`
    auto issue_rows = [&](int g, int ky, char* dst, int p0, int p1) {
        const bool s1 = g < a.Gsplit;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 ? a.in : a.in2), 0, s1 ? a.in_bytes : a.in2_bytes, 0x00020000);
        const unsigned soff = ((s1 ? base1 : base2) + (unsigned)(s1 ? g : g - a.Gsplit) * (unsigned)HW) * 16u;
        const int dy = ky - pad;
        const unsigned tapoff = (unsigned)(dy * W) * 16u;
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            if (p < p0 || p >= p1) continue;
            const bool ok = xok[p] & ((unsigned)(ybx[p] + dy) < (unsigned)H);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(dst + p * 1024), 16, ok ? (unsigned)vbx[p] + tapoff : 0xFFFFFFFFu, soff, 0, 0);
        }
    };
    // block bi = (ky, gb): this wave stages group 8 gb + wave; left-over image of filter row q: wave w < NPX stages piece w
#define GLDSX_ISSUE_BLOCK(bi_) do { const int b_ = (bi_), ky_ = b_ / NGB, gb_ = b_ - ky_ * NGB; issue_rows(8 * gb_ + wave, ky_, Xbuf + (b_ & 1) * BX + wave * BXG, 0, NPX); } while (0)
#define GLDSX_ISSUE_LEFT(q_) do { const int l_ = (q_); if (wave < NPX) issue_rows(8 * NGB, l_, Lbuf + (l_ & 1) * BXG, wave, wave + 1); } while (0)

    f32x16 acc[CI][PI];
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int j = 0; j < PI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, kh = lane >> 5;
    unsigned rowoff[PI];
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int R = wp * WTP + 32 * j + frow, seg = R >> lgseg, within = R & (seglen - 1);
        rowoff[j] = (unsigned)(seg * SEGW + within) * 16u;
    }
    // B fragments of slot pair s: bbase + (2 s + kh) * pitch + rowoff[j]
    // (two lambdas from one macro, not a generic lambda: hipcc emits no host stub for a __global__ template whose body holds one)
#define GLDSX_COMPUTE(NAME, PITCH)                                                                                                   \
    auto NAME = [&](int abuf, const char* bbase) {     /* PITCH: a compile-time constant, so that every fragment address is one register + an immediate */ \
        constexpr unsigned pitch = (PITCH);                                                                                          \
        const char* Ab = Abuf + abuf * AB + (kh * TC + wc * WTC + frow) * 16;                                                        \
        const char* Bb = bbase + kh * pitch;                                                                                         \
        f16x8 af[2][CI], bf[2][PI];                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < CI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(Ab + 32 * i * 16);                 \
        _Pragma("unroll") for (int j = 0; j < PI; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(Bb + rowoff[j]);                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                              \
            if (s < 3) {                                                                                                             \
                _Pragma("unroll") for (int i = 0; i < CI; ++i) af[(s + 1) & 1][i] = *reinterpret_cast<const f16x8*>(Ab + (2 * (s + 1) * TC + 32 * i) * 16); \
                _Pragma("unroll") for (int j = 0; j < PI; ++j) bf[(s + 1) & 1][j] = *reinterpret_cast<const f16x8*>(Bb + 2 * (s + 1) * pitch + rowoff[j]); \
            }                                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < CI; ++i)                                                                           \
                _Pragma("unroll") for (int j = 0; j < PI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0); \
        }                                                                                                                            \
        constexpr int NR = CI + PI, NM = CI * PI;     /* fragment reads of pair s + 1 ISSUED before the MFMAs of pair s (as in conv_glds_kernel) */ \
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);                                                                          \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) { __builtin_amdgcn_sched_group_barrier(0x100, NR, 0); __builtin_amdgcn_sched_group_barrier(0x008, NM, 0); } \
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                                                                          \
    }
    GLDSX_COMPUTE(compute_main, (unsigned)BXG);
    GLDSX_COMPUTE(compute_left, 16u);
#undef GLDSX_COMPUTE
    // Every step issues its pixel images FIRST and its filter pieces LAST: vmcnt retires in order, so "all but the newest NPA
    // operations" = everything but the filters of the step after next.
    auto wait_dma = [&](bool filters_in_flight) {
        if (NBUF == 3 && filters_in_flight) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence would drain the DMA left in flight
    };
    const int nblocks = ks * NGB;
    GLDSX_ISSUE_BLOCK(0);
    issue_a(0, 0);
    if (NBUF == 3 && nk > 1) issue_a(1, 1);
    wait_dma(nk > 1);
    int cur = 0, nxt = NBUF - 1;
    int bi = 0, kx = 0;                                                 // main walk: block, tap within the block
    for (int kt = 0; kt < nmain; ++kt) {
        // images first: the next block at the first tap of this one; the first left-over image one step early
        if (kx == 0 && bi + 1 < nblocks) GLDSX_ISSUE_BLOCK(bi + 1);
        if (walkk.r && kt + 1 == nmain) GLDSX_ISSUE_LEFT(0);
        if (kt + NBUF - 1 < nk) issue_a(kt + NBUF - 1, nxt);
        compute_main(cur, Xbuf + (bi & 1) * BX + kx * 16);
        wait_dma(kt + 2 < nk);
        cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
        if (++kx == ks) { kx = 0; ++bi; }
    }
    for (int kt = nmain; kt < nk; ++kt) {                               // the left-over group: one k-step per filter row
        const int q = kt - nmain;
        if (q + 1 < ks) GLDSX_ISSUE_LEFT(q + 1);
        if (kt + NBUF - 1 < nk) issue_a(kt + NBUF - 1, nxt);
        compute_left(cur, Lbuf + (q & 1) * BXG);
        wait_dma(kt + 2 < nk);
        cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

#undef GLDSX_ISSUE_BLOCK
#undef GLDSX_ISSUE_LEFT

    // ---- epilogue: acc row = cout (r&3)+8*(r>>2)+4*(lane>>5), col = pixel lane&31
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int pix = rem0 + wp * WTP + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < CI; ++i) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int c = c0 + wc * WTC + i * 32 + 8 * qd + 4 * (lane >> 5);
                if (c >= a.Cout) continue;
                const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v = make_float4(acc[i][j][4 * qd + 0] + b.x, acc[i][j][4 * qd + 1] + b.y, acc[i][j][4 * qd + 2] + b.z, acc[i][j][4 * qd + 3] + b.w);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};   // c8: channels c..c+3 = half of the 16-byte group c/8
                char* o = reinterpret_cast<char*>(a.out + c4_offset(img, a.Gout_tot, a.gout0 + (c >> 3), HW, pix)) + (c & 4) * 2;
                *reinterpret_cast<f16x4*>(o) = h;
            }
        }
    }
}

// ------------------------------------------------------------------ weight packing
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                 const float* __restrict__ var, float eps, int Cout, int Cout_pad, int Cin, int ks, int rot,
                                 int Kpad, float* __restrict__ wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Kpad * Cout_pad) return;
    const int kk = (int)(idx % CONV_BK);
    const int co = (int)((idx / CONV_BK) % Cout_pad);
    const int kstep = (int)((idx / CONV_BK) / Cout_pad);
    const int k = kstep * CONV_BK + kk;
    const int Cp = 4 * ((Cin + 3) / 4);
    const int tap = k / Cp, cp = k - tap * Cp;
    float v = 0.f;
    if (tap < ks * ks && cp < Cin && co < Cout) {
        const int ci = (cp + rot) % Cin;
        double s = 1.0;
        if (gamma) s = (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)((double)w[((size_t)co * Cin + ci) * ks * ks + tap] * s);
    }
    wp[idx] = v;
}

__global__ void pack_bias_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var,
                                 const float* __restrict__ bias, float eps, int Cout, float* __restrict__ bp) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Cout) return;
    double b = bias ? (double)bias[co] : 0.0;
    if (gamma) {
        const double s = (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        b = (double)beta[co] + (b - (double)mean[co]) * s;
    }
    bp[co] = (float)b;
}

static inline int round64(int c) { return (c + 63) / 64 * 64; }
static inline int conv_kpad(int Cin, int ks) { return ((ks * ks * 4 * ((Cin + 3) / 4) + CONV_BK - 1) / CONV_BK) * CONV_BK; }

extern "C" size_t cnm_packed_conv_floats(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || ksize <= 0) return 0;
    return (size_t)conv_kpad(Cin, ksize) * (size_t)round64(Cout);
}

extern "C" int cnm_pack_conv_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_beta,
                                    const float* bn_mean, const float* bn_var, const float* bias, float eps,
                                    int Cout, int Cin, int ksize, int rot,
                                    float* w_packed, float* b_packed, void* stream) {
    CNM_REQUIRE(w_oihw && w_packed && b_packed, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cin > 0 && ksize > 0 && (ksize & 1) && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    const bool bn = bn_gamma || bn_beta || bn_mean || bn_var;
    CNM_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bn_var), CNM_ERR_BAD_ARG);
    const int Kpad = conv_kpad(Cin, ksize);
    const long long total = (long long)Kpad * round64(Cout);
    pack_conv_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(
        w_oihw, bn_gamma, bn_var, eps, Cout, round64(Cout), Cin, ksize, rot, Kpad, w_packed);
    pack_bias_kernel<<<cnm_ceil_div(Cout, 256), 256, 0, cnm_stream(stream)>>>(
        bn_gamma, bn_beta, bn_mean, bn_var, bias, eps, Cout, b_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// Tile choice of the LDS-DMA kernel (tools/f16_conv_probe.py).  Base tile 128 x 256 (64 x 512 when Cout is not a multiple
// of 128).  If that leaves most CUs without a workgroup (< 160 workgroups) or the reduction is so short (<= 12 k-steps, Cout =
// 64 layers) that prologue and epilogue dominate: 64 x 128 with 32 x 32 wave tiles, two workgroups per CU.  Otherwise the
// larger tiles 256 x 256 (waves of 128 x 64; Cout a multiple of 256) and 128 x 512 (waves of 64 x 128; stride 1) are taken
// when rounds-of-256-workgroups x tile work / measured efficiency (1.2 and 1.12 against 1.0: fewer operand bytes and
// fragment reads per MFMA) comes out lower -- i.e. when the larger tile does not end in a mostly empty last round.
// g_glds_tile != 0 forces a tile (tests, probes): 1 128x256, 2 64x512, 3 64x128, 4 128x512, 5 256x256.
static int lds_per_block() {                                            // bytes of LDS one workgroup may use on the current device (queried once per device)
    static int cache[64] = {0};
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64 && cache[dev]) return cache[dev];
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64) cache[dev] = v;
    return v;
}
static int g_glds_tile = 0;
extern "C" int cnm_tune_glds_tile(int n) { const int old = g_glds_tile; if (n >= 0 && n <= 5) g_glds_tile = n; return old; }
static int g_gldsx = 1;                                                 // 1: stride-1 layers that qualify run on conv_gldsx_kernel (row-extended pixel operand); 0: conv_glds_kernel everywhere (A/B)
extern "C" int cnm_tune_gldsx(int n) { const int old = g_gldsx; if (n == 0 || n == 1) g_gldsx = n; return old; }
// conv_gldsx_kernel's scope (see the kernel): returns false when the layer stays on conv_glds_kernel
static bool launch_gldsx(const ConvArgs& a, hipStream_t s) {
    const F16Walk wk(a.Gin, a.ks);
    if (!g_gldsx || g_glds_tile || a.stride != 1 || a.Ho != a.H || a.Wo != a.W || a.W % 32 || a.W > 256 || (a.W & (a.W - 1)) || (a.H * a.W) % 256 || wk.NGB < 1 || wk.r > 1) return false;
    if (lds_per_block() < 160 * 1024) return false;
    auto wgs = [&](int tc, int tp) { return (long long)(a.Cout_pad / tc) * (a.M / tp); };
    const bool c128 = a.Cout_pad % 128 == 0, c256 = a.Cout_pad % 256 == 0;
    if (c256 && (wgs(256, 256) + 255) / 256 * 2.0 / 1.15 < (double)((wgs(128, 256) + 255) / 256)) conv_gldsx_kernel<256, 256, 128, 64><<<(unsigned)wgs(256, 256), 512, 0, s>>>(a);
    else if (c128) conv_gldsx_kernel<128, 256, 64, 64><<<(unsigned)wgs(128, 256), 512, 0, s>>>(a);
    else conv_gldsx_kernel<64, 256, 32, 64><<<(unsigned)wgs(64, 256), 512, 0, s>>>(a);
    return true;
}
template <bool UPS = false>
static void launch_glds(const ConvArgs& a, hipStream_t s) {
    if (!UPS && launch_gldsx(a, s)) return;
    auto wgs = [&](int tc, int tp) { return (long long)(a.Cout_pad / tc) * cnm_ceil_div(a.M, tp); };
    const bool c128 = a.Cout_pad % 128 == 0, c256 = a.Cout_pad % 256 == 0;
    const long long big = c128 ? wgs(128, 256) : wgs(64, 512);
    int v = g_glds_tile;
    const int lds = lds_per_block();                                    // static LDS of the tiles: 128x512 160 KB, 128x256 / 64x512 144 KB, 256x256 128 KB, 64x128 72 KB
    if (v == 0) {
        v = (big < 160 || (!c128 && a.nk <= 12)) ? 3 : c128 ? 1 : 2;
        if (v == 1) {
            auto cost = [&](int tc, int tp, double work, double eff) { return (double)((wgs(tc, tp) + 255) / 256) * work / eff; };
            double best = cost(128, 256, 1.0, 1.0);
            if (c256 && cost(256, 256, 2.0, 1.2) < best) { best = cost(256, 256, 2.0, 1.2); v = 5; }
            if (a.stride == 1 && lds >= 160 * 1024 && cost(128, 512, 2.0, 1.12) < best) v = 4;   // that tile takes all 160 KB
        }
    }
    if (!c128 && (v == 1 || v == 4 || v == 5)) v = 2;
    if (v == 5 && !c256) v = 1;
    // a device (or partition) with less LDS per workgroup: every choice, forced ones included, falls back to a tile that fits
    if (v == 4 && lds < 160 * 1024) v = 1;
    if ((v == 1 || v == 2) && lds < 144 * 1024) v = (c256 && lds >= 128 * 1024) ? 5 : 3;
    if (v == 5 && lds < 128 * 1024) v = 3;
    if (v == 1) conv_glds_kernel<128, 256, 64, 64, UPS><<<(unsigned)wgs(128, 256), 512, 0, s>>>(a);
    else if (v == 2) conv_glds_kernel<64, 512, 64, 64, UPS><<<(unsigned)wgs(64, 512), 512, 0, s>>>(a);
    else if (v == 3) conv_glds_kernel<64, 128, 32, 32, UPS><<<(unsigned)wgs(64, 128), 512, 0, s>>>(a);
    else if (v == 4) conv_glds_kernel<128, 512, 64, 128, UPS><<<(unsigned)wgs(128, 512), 512, 0, s>>>(a);
    else conv_glds_kernel<256, 256, 128, 64, UPS><<<(unsigned)wgs(256, 256), 512, 0, s>>>(a);
}

template <int TC, int TP, int TS = 1>
static void launch_conv(const ConvArgs& a, hipStream_t s) {
    const int nblocks = (a.Cout_pad / TC) * cnm_ceil_div(a.M, TP);
    conv_mfma_f32_kernel<TC, TP, TS><<<nblocks, 256, 0, s>>>(a);
}

// out_h/out_w > 0 selects the transposed (data-gradient) gather: `in` is then dy [N,.,H,W] and the output is
// out_h x out_w with tstride in {1,2}; otherwise an ordinary convolution with `stride`.
static int conv_dispatch(const float* in, int Gin_total, int gin0, int Gin,
                         const float* in2, int Gin2_total, int gin2_0, int Gsplit,
                         float* out, int Gout_total, int gout0, int Cout,
                         const float* w_packed, const float* b_packed,
                         int N, int H, int W, int ksize, int stride, int relu, void* stream,
                         int out_h = 0, int out_w = 0, int tstride = 1, bool f16 = false) {
    CNM_REQUIRE(in && out && w_packed, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(N > 0 && H > 0 && W > 0 && Gin > 0 && Gsplit > 0 && Gsplit <= Gin, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(gin0 >= 0 && gin0 + Gsplit <= Gin_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Gsplit == Gin || (in2 && gin2_0 >= 0 && gin2_0 + (Gin - Gsplit) <= Gin2_total), CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % (f16 ? 8 : 4) == 0 && gout0 >= 0 && gout0 + Cout / (f16 ? 8 : 4) <= Gout_total, CNM_ERR_BAD_ARG);
    CNM_REQUIRE((ksize == 3 || ksize == 5 || ksize == 7) && (stride == 1 || stride == 2), CNM_ERR_BAD_ARG);
    ConvArgs a;
    a.in = in; a.out = out; a.w = w_packed; a.bias = b_packed;
    const bool two = Gsplit < Gin;
    const unsigned long long b1 = (unsigned long long)N * Gin_total * H * W * 16ull;
    const unsigned long long b2 = two ? (unsigned long long)N * Gin2_total * H * W * 16ull : b1;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && b2 < 0xFFFFFFFFull, CNM_ERR_BAD_ARG);   // 32-bit buffer offsets
    CNM_REQUIRE(Gin >= CONV_BK / 4, CNM_ERR_BAD_ARG);                          // loader: one tap wrap per k-step
    a.in2 = two ? in2 : in; a.Gin2_tot = two ? Gin2_total : Gin_total; a.gin2_0 = two ? gin2_0 : gin0; a.Gsplit = Gsplit;
    a.in_bytes = (unsigned)b1; a.in2_bytes = (unsigned)b2;
    a.N = N; a.H = H; a.W = W;
    a.ks = ksize; a.pad = (ksize - 1) / 2;
    const bool transposed = out_h > 0;
    if (transposed) { a.stride = 1; a.Ho = out_h; a.Wo = out_w; }             // pad' = ks-1-pad = pad for odd ks
    else { a.stride = stride; a.Ho = (H + 2 * a.pad - ksize) / stride + 1; a.Wo = (W + 2 * a.pad - ksize) / stride + 1; }
    a.Gin_tot = Gin_total; a.gin0 = gin0; a.Gin = Gin;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = Cout; a.Cout_pad = round64(Cout);
    a.nk = ((ksize * ksize * 4 * Gin + CONV_BK - 1) / CONV_BK);
    a.M = N * a.Ho * a.Wo; a.relu = relu;
    hipStream_t s = cnm_stream(stream);
    // Tile choice: largest tile that still gives every CU (256) a couple of workgroups.
    const long long t128 = (long long)(a.Cout_pad / 128) * cnm_ceil_div(a.M, 128);
    const long long t64x128 = (long long)(a.Cout_pad / 64) * cnm_ceil_div(a.M, 128);
    if (f16) {                                                                 // fp16: every layer on the LDS-DMA kernel (measured faster down to 6x8 images)
        CNM_REQUIRE(!transposed, CNM_ERR_BAD_ARG);
        a.nk = F16Walk(Gin, ksize).nk;                                         // k-steps of 64 halfs
        a.w_bytes = (unsigned)((size_t)a.nk * 64 * a.Cout_pad * 2);
        launch_glds(a, s);
    } else if (transposed && tstride == 2) {
        if (a.Cout_pad % 128 == 0 && t128 >= 512) launch_conv<128, 128, 2>(a, s);
        else if (t64x128 >= 512) launch_conv<64, 128, 2>(a, s);
        else launch_conv<64, 64, 2>(a, s);
    } else {
        if (a.Cout_pad % 128 == 0 && t128 >= 512) launch_conv<128, 128>(a, s);
        else if (t64x128 >= 512) launch_conv<64, 128>(a, s);
        else launch_conv<64, 64>(a, s);
    }
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv2d_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                 float* out, int Gout_total, int gout0, int Cout,
                                 const float* w_packed, const float* b_packed,
                                 int N, int H, int W, int ksize, int stride, int relu, void* stream) {
    return conv_dispatch(in, Gin_total, gin0, Gin, nullptr, 0, 0, Gin, out, Gout_total, gout0, Cout,
                         w_packed, b_packed, N, H, W, ksize, stride, relu, stream);
}

extern "C" int cnm_conv2d_cat2_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                      const float* in_b, int Gb_total, int gb0, int Gb,
                                      float* out, int Gout_total, int gout0, int Cout,
                                      const float* w_packed, const float* b_packed,
                                      int N, int H, int W, int ksize, int stride, int relu, void* stream) {
    CNM_REQUIRE(Ga > 0 && Gb > 0, CNM_ERR_BAD_ARG);
    return conv_dispatch(in_a, Ga_total, ga0, Ga + Gb, in_b, Gb_total, gb0, Ga, out, Gout_total, gout0, Cout,
                         w_packed, b_packed, N, H, W, ksize, stride, relu, stream);
}

// ------------------------------------------------------------------ data gradient (training)
// dX = conv_transpose(dY, W): the same implicit GEMM with the roles of Cin/Cout swapped and the taps
// flipped,  w'[ci][(a,b), co] = w[co][ci][ks-1-a][ks-1-b];  output channels follow the (rotated) channel
// order of the forward input, so dX has exactly the layout of X.
__global__ void pack_conv_dgrad_kernel(const float* __restrict__ w, int Cout, int Cin, int Cin_pad, int ks, int rot,
                                       int Kpad, float* __restrict__ wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Kpad * Cin_pad) return;
    const int kk = (int)(idx % CONV_BK);
    const int p = (int)((idx / CONV_BK) % Cin_pad);          // packed input-channel position = output row here
    const int kstep = (int)((idx / CONV_BK) / Cin_pad);
    const int k = kstep * CONV_BK + kk;
    const int Cp = 4 * ((Cout + 3) / 4);
    const int tap = k / Cp, co = k - tap * Cp;
    float v = 0.f;
    if (tap < ks * ks && co < Cout && p < Cin) {
        const int ci = (p + rot) % Cin;
        v = w[((size_t)co * Cin + ci) * ks * ks + (ks * ks - 1 - tap)];
    }
    wp[idx] = v;
}

extern "C" size_t cnm_packed_dgrad_floats(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || ksize <= 0) return 0;
    return (size_t)conv_kpad(Cout, ksize) * (size_t)round64(4 * ((Cin + 3) / 4));
}

extern "C" int cnm_pack_conv_dgrad_f32(const float* w_oihw, int Cout, int Cin, int ksize, int rot,
                                       float* w_packed, void* stream) {
    CNM_REQUIRE(w_oihw && w_packed && Cout > 0 && Cin > 0 && (ksize & 1) && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    const int Kpad = conv_kpad(Cout, ksize), Cin_pad = round64(4 * ((Cin + 3) / 4));
    const long long total = (long long)Kpad * Cin_pad;
    pack_conv_dgrad_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(
        w_oihw, Cout, Cin, Cin_pad, ksize, rot, Kpad, w_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv2d_dgrad_c4_f32(const float* dy, int Gy_total, int gy0, int Cout,
                                       float* dx, int Gx_total, int gx0, int Cin,
                                       const float* w_packed_dgrad, int N, int H, int W,
                                       int ksize, int stride, void* stream) {
    CNM_REQUIRE(Cout > 0 && Cin > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), CNM_ERR_BAD_ARG);
    const int pad = (ksize - 1) / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int Gy = (Cout + 3) / 4;
    return conv_dispatch(dy, Gy_total, gy0, Gy, nullptr, 0, 0, Gy, dx, Gx_total, gx0, 4 * ((Cin + 3) / 4),
                         w_packed_dgrad, nullptr, N, Ho, Wo, ksize, 1, 0, stream, H, W, stride);
}


// ------------------------------------------------------------------ fp16 path (BASELINE config 5)
__global__ void pack_conv_f16_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                     const float* __restrict__ var, float eps, int Cout, int Cout_pad, int Cin, int ks, int rot,
                                     int Kpad, _Float16* __restrict__ wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Kpad * Cout_pad) return;
    // [k-step of 64][8 channel groups][Cout_pad][8 halfs]: the LDS image of conv_glds_kernel, DMA piece by DMA piece
    const int e = (int)(idx % 8);
    const int co = (int)((idx / 8) % Cout_pad);
    const int kgrp = (int)((idx / 8) / Cout_pad);           // kstep * 8 + slot
    const F16Walk walkk((Cin + 7) / 8, ks);
    int ky, kx, g;
    const bool ok = walkk.slot(kgrp >> 3, kgrp & 7, ky, kx, g);
    const int cp = 8 * g + e;
    float v = 0.f;
    if (ok && cp < Cin && co < Cout) {
        const int ci = (cp + rot) % Cin;
        double s = 1.0;
        if (gamma) s = (double)gamma[co] / sqrt((double)var[co] + (double)eps);
        v = (float)((double)w[((size_t)co * Cin + ci) * ks * ks + ky * ks + kx] * s);
    }
    wp[idx] = (_Float16)v;
}

static inline int conv_kpad_f16(int Cin, int ks) { return F16Walk((Cin + 7) / 8, ks).nk * 64; }

extern "C" size_t cnm_packed_conv_halfs(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || ksize <= 0) return 0;
    return (size_t)conv_kpad_f16(Cin, ksize) * (size_t)round64(Cout);
}

extern "C" int cnm_pack_conv_bn_f16(const float* w_oihw, const float* bn_gamma, const float* bn_beta,
                                    const float* bn_mean, const float* bn_var, const float* bias, float eps,
                                    int Cout, int Cin, int ksize, int rot,
                                    void* w_packed_f16, float* b_packed, void* stream) {
    CNM_REQUIRE(w_oihw && w_packed_f16 && b_packed, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cin > 0 && ksize > 0 && (ksize & 1) && rot >= 0 && rot < Cin, CNM_ERR_BAD_ARG);
    const bool bn = bn_gamma || bn_beta || bn_mean || bn_var;
    CNM_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bn_var), CNM_ERR_BAD_ARG);
    const int Kpad = conv_kpad_f16(Cin, ksize);
    const long long total = (long long)Kpad * round64(Cout);
    pack_conv_f16_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(
        w_oihw, bn_gamma, bn_var, eps, Cout, round64(Cout), Cin, ksize, rot, Kpad, static_cast<_Float16*>(w_packed_f16));
    pack_bias_kernel<<<cnm_ceil_div(Cout, 256), 256, 0, cnm_stream(stream)>>>(
        bn_gamma, bn_beta, bn_mean, bn_var, bias, eps, Cout, b_packed);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_conv2d_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                                 void* out, int Gout_total, int gout0, int Cout,
                                 const void* w_packed_f16, const float* b_packed,
                                 int N, int H, int W, int ksize, int stride, int relu, void* stream) {
    return conv_dispatch(static_cast<const float*>(in), Gin_total, gin0, Gin, nullptr, 0, 0, Gin,
                         static_cast<float*>(out), Gout_total, gout0, Cout, static_cast<const float*>(w_packed_f16), b_packed,
                         N, H, W, ksize, stride, relu, stream, 0, 0, 1, true);
}

extern "C" int cnm_conv2d_cat2_c8_f16(const void* in_a, int Ga_total, int ga0, int Ga,
                                      const void* in_b, int Gb_total, int gb0, int Gb,
                                      void* out, int Gout_total, int gout0, int Cout,
                                      const void* w_packed_f16, const float* b_packed,
                                      int N, int H, int W, int ksize, int stride, int relu, void* stream) {
    CNM_REQUIRE(Ga > 0 && Gb > 0, CNM_ERR_BAD_ARG);
    return conv_dispatch(static_cast<const float*>(in_a), Ga_total, ga0, Ga + Gb, static_cast<const float*>(in_b), Gb_total, gb0, Ga,
                         static_cast<float*>(out), Gout_total, gout0, Cout, static_cast<const float*>(w_packed_f16), b_packed,
                         N, H, W, ksize, stride, relu, stream, 0, 0, 1, true);
}

// conv3x3(upsample2x(in)) + bias (+ ReLU) on fp16 c8 data without the upsampled tensor: in [N][Gin_total][H][W][8] ->
// out [N][Gout_total][2H][2W][8]; w_packed_f16 / b_packed are packed (cnm_pack_conv_bn_f16) from the four composed phase
// filters as 4 * Cout output channels, phase major (reference up_conv_layer, depthNet_model.py:89-112).  with_ring = 0:
// complete result with REPLICATE padding of the upsampled image; 1: the one-pixel output ring is left un-biased /
// un-activated for cnm_conv3x3_upsampled_ring_c8_f16, which turns it into the reference's zero-padding result.
extern "C" int cnm_conv3x3_upsampled_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                                            void* out, int Gout_total, int gout0, int Cout,
                                            const void* w_packed_f16, const float* b_packed,
                                            int N, int H, int W, int relu, int with_ring, void* stream) {
    CNM_REQUIRE(in && out && w_packed_f16 && N > 0 && H > 0 && W > 0 && Gin > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(Cout > 0 && Cout % 8 == 0 && gout0 >= 0 && gout0 + Cout / 8 <= Gout_total && gin0 >= 0 && gin0 + Gin <= Gin_total, CNM_ERR_BAD_ARG);
    ConvArgs a;
    a.in = a.in2 = static_cast<const float*>(in); a.out = static_cast<float*>(out); a.w = static_cast<const float*>(w_packed_f16); a.bias = b_packed;
    const unsigned long long b1 = (unsigned long long)N * Gin_total * H * W * 16ull;
    CNM_REQUIRE(b1 < 0xFFFFFFFFull && (unsigned long long)N * Gout_total * 4 * H * W * 16ull < (1ull << 40), CNM_ERR_BAD_ARG);
    a.in_bytes = a.in2_bytes = (unsigned)b1;
    a.Gin2_tot = Gin_total; a.gin2_0 = gin0; a.Gsplit = Gin;
    a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
    a.Gin_tot = Gin_total; a.gin0 = gin0; a.Gin = Gin;
    a.Gout_tot = Gout_total; a.gout0 = gout0; a.Cout = 4 * Cout; a.Cout_pad = round64(4 * Cout);
    a.ks = 3; a.stride = 1; a.pad = 1;
    a.nk = F16Walk(Gin, 3).nk; a.w_bytes = (unsigned)((size_t)a.nk * 64 * a.Cout_pad * 2);
    a.M = N * H * W; a.relu = relu; a.ring = with_ring;
    launch_glds<true>(a, cnm_stream(stream));
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
