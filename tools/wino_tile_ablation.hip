// Go / no-go probe (GPU box): would a per-wave tile with A-fragment reuse x2 lift the staged F(4x4,3x3)
// kernel's multiply loop?  The loop in isolation, as tools/wino36s_ablate.sh's ABL modes leave it -- MFMAs
// (v_mfma_f32_16x16x4_f32, exact fp32), the weight stream (one 16-byte fragment per lane, point and 16-channel chunk, straight
// from L2 / HBM, WD in flight) and the B-fragment reads (one ds_read_b128 per point and chunk from a static V buffer) -- no
// transform, no DMA, no epilogue, in two shapes:
//   NT = 1  the shipped tile: a wave = 16 output channels x 16 tiles x 36 points = 144 accumulator registers, two waves per SIMD
//           (8-wave workgroups, one per CU); a weight fragment feeds ONE MFMA.
//   NT = 2  the candidate: a wave = 16 output channels x 32 tiles = 288 accumulator registers, ONE wave per SIMD (4-wave
//           workgroups); a weight fragment feeds TWO MFMAs: half the weight stream per flop, the same LDS bytes per flop.
// Both run the same flops per CU and phase (a "phase" = one 16-channel chunk for 128 output channels x 16 tiles, resp.
// 64 x 32).  WEIGHTS = 0 replaces the stream by registers (the pure MFMA + LDS loop).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_tile_ablation.hip -o /tmp/wta && /tmp/wta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int WEIGHTS, int WD>
__global__ __launch_bounds__(NT == 1 ? 512 : 256, NT == 1 ? 2 : 1) void loop_kernel(const f32x4* __restrict__ wts, float* __restrict__ out, int phases, int nchunks, int blocks_of_weights) {
    __shared__ f32x4 V[36 * 16 * 4 * NT];                              // [point][tile][k group]: 36 KB per 16 tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 16 * 4 * NT; i += blockDim.x) V[i] = f32x4{(float)(i & 7) * 0.125f, 0.5f, -0.25f, (float)(i & 3)};
    __syncthreads();
    f32x4 acc[36 * NT];
#pragma unroll
    for (int x = 0; x < 36 * NT; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weights: [block][chunk][wave of 8][point][lane] fragments; a workgroup streams the slice of its waves
    const int nw = NT == 1 ? 8 : 4;
    const int blk = blockIdx.x % blocks_of_weights;
    const f32x4* wbase = wts + ((size_t)blk * nchunks * 8 + (NT == 1 ? wave : wave + 4 * (blockIdx.x & 1))) * 36 * 64 + lane;
    const size_t chunk_stride = (size_t)8 * 36 * 64;
    f32x4 afr[WD];
    int cq = 0;
    const f32x4* wp = wbase;
    if (WEIGHTS) {
#pragma unroll
        for (int i = 0; i < WD; ++i) afr[i] = wp[i * 64];
    } else {
#pragma unroll
        for (int i = 0; i < WD; ++i) afr[i] = f32x4{1.f + lane, 0.5f, 0.25f, 2.f};
    }
    f32x4 bcur[NT], bnxt[NT];
#pragma unroll
    for (int h = 0; h < NT; ++h) bcur[h] = V[(h * 16 + (lane & 15)) * 4 + (lane >> 4)];
    for (int p = 0; p < phases; ++p) {
        const f32x4* wn = wbase + (size_t)((cq + 1) % nchunks) * chunk_stride;
#pragma unroll
        for (int pt = 0; pt < 36; ++pt) {
            const f32x4 a = afr[pt % WD];
            if (WEIGHTS) {                                              // refill the slot: fragment pt + WD of this chunk, or the next chunk's first ones
                const int nx = pt + WD;
                afr[pt % WD] = nx < 36 ? wp[nx * 64] : wn[(nx - 36) * 64];
            }
            const int pn = (pt + 1) % 36;                               // the next point's B fragments travel during this point's MFMAs
#pragma unroll
            for (int h = 0; h < NT; ++h) bnxt[h] = V[(pn * 16 * NT + h * 16 + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
            for (int h = 0; h < NT; ++h) {
                f32x4 c = acc[pt * NT + h];
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bcur[h].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bcur[h].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bcur[h].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bcur[h].w, c, 0, 0, 0);
                acc[pt * NT + h] = c;
            }
#pragma unroll
            for (int h = 0; h < NT; ++h) bcur[h] = bnxt[h];
            __builtin_amdgcn_sched_barrier(0);
        }
        cq = (cq + 1) % nchunks; wp = wn;
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 36 * NT; ++x) s += acc[x];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    (void)nw;
}

template <int NT, int WEIGHTS, int WD>
static void run(const char* what, const f32x4* wts, float* out, int nchunks, int blocks_of_weights) {
    const int cus = 256, phases = 2000;
    const int grid = NT == 1 ? cus : 2 * cus;                          // NT = 2: two 4-wave workgroups per CU would be 2 waves per SIMD -- ONE per CU is the candidate
    const int g = NT == 1 ? cus : cus;
    (void)grid;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop_kernel<NT, WEIGHTS, WD><<<g, NT == 1 ? 512 : 256>>>(wts, out, 50, nchunks, blocks_of_weights); hipDeviceSynchronize();
    hipEventRecord(e0); loop_kernel<NT, WEIGHTS, WD><<<g, NT == 1 ? 512 : 256>>>(wts, out, phases, nchunks, blocks_of_weights); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (NT == 1 ? 8.0 : 4.0) * g, mfma = waves * phases * 144.0 * NT, flops = mfma * 16 * 16 * 4 * 2;
    printf("%-86s %7.3f ms  %6.1f TF executed = %.3f of 157.3\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
    const int nchunks = 16, nblocks = 4;                                 // Cin = 256: 16 chunks; four 128-channel blocks of filters (Cout = 512): 18.9 MB of fragments
    const size_t n = (size_t)nblocks * nchunks * 8 * 36 * 64;
    f32x4* wts; hipMalloc(&wts, n * 16);
    std::vector<float> h(n * 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-4f - 0.2f;
    hipMemcpy(wts, h.data(), n * 16, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 512 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 0, 4>("shipped tile (16 x 16 per wave, 2 waves / SIMD), no weight stream", wts, out, nchunks, nblocks);
        run<1, 1, 4>("shipped tile, weight stream (4 fragments in flight)", wts, out, nchunks, nblocks);
        run<1, 1, 8>("shipped tile, weight stream (8 fragments in flight)", wts, out, nchunks, nblocks);
        run<2, 0, 4>("reuse x2 (16 x 32 per wave, 1 wave / SIMD), no weight stream", wts, out, nchunks, nblocks);
        run<2, 1, 4>("reuse x2, weight stream (4 fragments in flight)", wts, out, nchunks, nblocks);
        run<2, 1, 8>("reuse x2, weight stream (8 fragments in flight)", wts, out, nchunks, nblocks);
        run<1, 1, 4>("shipped tile, weight stream, ONE filter block for every workgroup (L2-hot)", wts, out, nchunks, 1);
        run<2, 1, 4>("reuse x2, weight stream, ONE filter block (L2-hot)", wts, out, nchunks, 1);
    }
    return 0;
}
