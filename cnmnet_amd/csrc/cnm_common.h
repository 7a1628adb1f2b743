// Shared helpers for the gfx950 kernels of the CNMNet depth engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cnm_engine.h"

#define CNM_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define CNM_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return CNM_ERR_LAUNCH; } while (0)

static inline hipStream_t cnm_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cnm_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long cnm_ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// Offset (in floats) of element (n, g, pix, 0) of a c4 view [N][G_total][HW][4].
__host__ __device__ static inline size_t c4_offset(int n, int G_total, int g, int HW, int pix) {
    return (((size_t)n * G_total + g) * (size_t)HW + pix) * 4;
}

// MI355X: 8 XCDs, workgroup b is observed on XCD b % 8 (speed only, never correctness).
// Bijective remap so each XCD walks a contiguous range of tile ids (guide T1).
__device__ static inline int xcd_remap(int bid, int nblocks) {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nblocks >> 3, r = nblocks & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
