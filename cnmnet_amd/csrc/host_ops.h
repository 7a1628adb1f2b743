// Host twins of the engine's operators (host_twins.cpp): plain C++ on HOST pointers, same argument meaning and the same
// c4 activation layout as the HIP entry points they mirror, so that the whole-network executors of nets.hip run them through
// a third engine policy (EngHost) -- BASELINE configs[0] ("1 ref + 1 src, 256x192, 32 planes, batch=1 on CPU, plumbing")
// through the product, without touching oracle/.  Each returns CNM_OK or a negative cnm_status.
#pragma once
#include <stddef.h>

namespace cnmh {
// cnm_homography_terms_f32
int homography(const float* ref_cam, const float* src_cam, float* hmkt, int B, int S);
// cnm_planesweep_cat_c4_f32 / cnm_planesweep_volume_nchw_f32 (nchw != 0: volume [B*S,D,H,W] instead of the c4 conv input)
int sweep(const float* ref, const float* src, const float* hmkt, float* out, int B, int S, int H, int W, int D, double idmin, double idmax, int nchw);
// cnm_conv2d_c4_f32 / cnm_conv2d_cat2_c4_f32 with the host filter layout w [Cout][k*k][4*(Ga+Gb)] (cnm_pack_conv_bn_cpu)
int conv(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
         float* out, int Gout_total, int gout0, int Cout, const float* w, const float* bias, int N, int H, int W, int ksize, int stride, int relu);
// cnm_upsample2x_c4_f32
int upsample2x(const float* in, int Gin_total, int gin0, float* out, int Gout_total, int gout0, int N, int G, int H, int W);
// cnm_head_sigmoid_c4_f32 (w_head [9][C] as cnm_pack_head_f32 lays it out)
int head(const float* in, int Gin_total, int gin0, int C, const float* w_head, const float* bias, float scale,
         float* disp, float* up_out, int up_Gtotal, int up_g, int N, int H, int W);
// cnm_refine_assemble_c4_f32 / cnm_refine_assemble_multi_c4_f32
int assemble(const float* id1, const float* id2, long long id_stride, const float* f1, int G1_total, int g1,
             const float* f2, int G2_total, int g2, float* x, int N, int C, int H, int W);
int assemble_multi(const float* idepth_pairs, const float* feat_pairs, float* x, int B, int S, int C, int H, int W);
}
