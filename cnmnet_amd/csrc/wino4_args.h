// Argument block shared by the two 36-point Winograd kernels (conv_winograd4.hip: gather-fed, conv_winograd4s.hip: LDS-staged).
#pragma once
#include "cnm_common.h"

struct Wino4Args {
    const float* in; const float* in2; float* out; const float* u; const float* bias;
    unsigned in_bytes, in2_bytes;
    int N, H, W, TH, TW;                 // TH = ceil(H/4), TW = ceil(W/4) tiles
    int Gin_tot, gin0, Gin2_tot, gin2_0, Gsplit, Gin;
    int Gout_tot, gout0, Cout;
    int nchunks, T, relu;                // T = N*TH*TW tiles
    int ring;                            // fused upsampling: leave the one-pixel output ring without bias / ReLU for the ring kernel
    int ups_zero;                        // UPS kernels: zero padding of the low-resolution input instead of edge replication (phase-scatter form)
    float* sync_ws; size_t sync_floats;  // LDS-staged kernel only: zeroed flag words + partial-output slots (cnm_wino36_sync_floats), or null
};

// LDS-staged variant (conv_winograd4s.hip): returns CNM_OK after launching, 1 when the shape is not eligible (the caller
// then launches the gather-fed kernel), a negative status on launch failure.
// s2: the stride-2 form on the four pixel phases of the input (a.H / a.W = output size, a.nchunks = 4 x the input's chunks,
// filter packed by cnm_pack_winograd4_s2_bn_f32): M = 4 for a 5x5, M = 3 for a 7x7 filter; there is no gather-fed twin.
int cnm_wino36s_try_launch(const Wino4Args& a, int M, int ups, hipStream_t stream, int s2 = 0);
// Four-wave variant (conv_winograd4q.hip): a.u = the quad-packed filter, a.nchunks = 8-channel chunks; same return convention.
int cnm_wino36q_try_launch(const Wino4Args& a, hipStream_t stream);
int cnm_wino36_quad_mode();                                              // cnm_tune_wino36_quad: 0 never, 1 where measured faster, 2 wherever eligible
