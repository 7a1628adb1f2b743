// Probe (GPU box): does gfx950 execute scalar atomics (s_atomic_add ... glc)?  Every wave of every workgroup draws a ticket
// from one counter through the scalar cache path; the host checks that the tickets are a permutation of 0 .. n-1.
//   hipcc --offload-arch=gfx950 -O3 tools/satomic_probe.hip -o /tmp/sap && /tmp/sap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* counter, unsigned* out) {
    unsigned t = 1u;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(counter) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t;
}
int main() {
    unsigned *c, *o; const int blocks = 1024, waves = 4, n = blocks * waves;
    hipMalloc(&c, 4); hipMalloc(&o, n * 4); hipMemset(c, 0, 4); hipMemset(o, 0xff, n * 4);
    k<<<blocks, waves * 64>>>(c, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<unsigned> h(n); unsigned hc; hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end()); int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != (unsigned)i;
    printf("scalar atomics: counter %u (want %d), tickets out of place %d, first %u last %u\n", hc, n, bad, h[0], h[n - 1]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipMemset(c, 0, 4);
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) k<<<blocks, waves * 64>>>(c, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("  %d tickets per launch: %.1f us per launch = %.0f tickets/us\n", n, ms / 20 * 1e3, n / (ms / 20 * 1e3));
    return 0;
}
