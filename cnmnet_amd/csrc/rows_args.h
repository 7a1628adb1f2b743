// Kernel arguments of the row-wise Winograd convolutions, shared by the gather-fed kernels (conv_winograd_rows.hip) and the
// LDS-staged 7x7 kernel (conv_rows_staged.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

struct RowArgs {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, Ho, Wo, TW;             // input H x W, output Ho x Wo, TW = ceil(Wo/2) tiles per row
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*Ho*TW tiles
};

// 7x7 stride 1, four outputs per tile, Cout % 128 == 0: CNM_OK after launching the staged kernel, 1 when the shape is not
// eligible (the caller launches the gather-fed kernel), negative on a launch failure.  sync_ws: see cnm_wino36_sync_floats().
int cnm_rows7s_try_launch(const RowArgs& a, float* sync_ws, size_t sync_floats, hipStream_t stream);
