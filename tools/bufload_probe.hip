// Probe (GPU box): which raw-buffer out-of-range forms are safe on gfx950?
//   mode 0: voffset = 0xFFFFFFFF, soffset = 0        mode 1: voffset = 0xFFFFFFFF, soffset = 0x30000
//   mode 2: voffset = num_records + 64, soffset = 0   mode 3: voffset = 0xFFFFFFFF - 0x30000 (sum stays below 2^32), soffset = 0x30000
//   mode 4: voffset = 0x80000000, soffset = 0x30000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(const float* base, unsigned bytes, int mode, unsigned soff, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    unsigned v = 0xFFFFFFFFu;
    if (mode == 2) v = bytes + 64u;
    if (mode == 3) v = 0xFFFFFFFFu - 0x30000u;
    if (mode == 4) v = 0x80000000u;
    if (mode == 5) v = threadIdx.x * 4u;
    const unsigned s = (mode == 0 || mode == 2) ? 0u : soff;
    out[threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0)) + 1.0f;
}
int main(int argc, char** argv) {
    const int mode = atoi(argv[1]);
    float *buf, *out; const unsigned bytes = 0x90000;
    hipMalloc(&buf, bytes); hipMalloc(&out, 256); hipMemset(buf, 0x3f, bytes);
    probe<<<1, 64>>>(buf, bytes, mode, 0x30000u, out);
    const hipError_t e = hipDeviceSynchronize();
    float h[64]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
    printf("mode %d: %s, out[0] = %g (1 = the load returned zero)\n", mode, hipGetErrorString(e), h[0]);
    return 0;
}
