// Probe: does an out-of-range lane of a buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds) write zeros to its LDS slot,
// or leave the slot untouched?  (the fp16 convolution's zero padding relies on the answer)   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, unsigned bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = -7.f;                       // sentinel
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const unsigned voff = (lane & 1) ? 0xFFFFFFFFu : (unsigned)lane * 16u;       // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    float *d, *o; (void)hipMalloc(&d, 1024); (void)hipMalloc(&o, 1024); (void)hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 1024, o);
    std::vector<float> r(256); (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int zeros = 0, kept = 0, good = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) { const float v = r[l * 4 + i];
        if (l & 1) { zeros += v == 0.f; kept += v == -7.f; } else good += v == h[l * 4 + i]; }
    printf("in-range lanes correct: %d/128   out-of-range lanes: %d zeroed, %d untouched (of 128)\n", good, zeros, kept);
    return 0;
}
