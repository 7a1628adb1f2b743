// Whole-network executors: depthNet.forward and DepthRefineNet.forward
// (reference depthnet/depthNet_model.py:226-263 and :331-370) as fixed launch sequences
// over a caller-owned workspace.  No allocation, no synchronisation, no global state:
// everything is enqueued on the caller's stream, so a forward can be captured in a hipGraph.
//
// Concatenations are never copied: a producer writes straight into the channel-group slice
// of its consumer's input buffer (CAT* below), or the consumer reads two views (cat2 conv).
#include <string.h>
#include "cnm_common.h"

// ------------------------------------------------------------------ layer tables
static const cnm_layer_info kDepthLayers[] = {
    {"conv1.0", "conv1.1", 67, 128, 7, 1, 3, 0},   {"conv1.3", "conv1.4", 128, 128, 7, 2, 0, 0},
    {"conv2.0", "conv2.1", 128, 256, 5, 1, 0, 0},  {"conv2.3", "conv2.4", 256, 256, 5, 2, 0, 0},
    {"conv3.0", "conv3.1", 256, 512, 3, 1, 0, 0},  {"conv3.3", "conv3.4", 512, 512, 3, 2, 0, 0},
    {"conv4.0", "conv4.1", 512, 512, 3, 1, 0, 0},  {"conv4.3", "conv4.4", 512, 512, 3, 2, 0, 0},
    {"conv5.0", "conv5.1", 512, 512, 3, 1, 0, 0},  {"conv5.3", "conv5.4", 512, 512, 3, 2, 0, 0},
    {"upconv5.1", "upconv5.2", 512, 512, 3, 1, 0, 0}, {"iconv5.0", "iconv5.1", 1024, 512, 3, 1, 0, 0},
    {"upconv4.1", "upconv4.2", 512, 512, 3, 1, 0, 0}, {"iconv4.0", "iconv4.1", 1024, 512, 3, 1, 0, 0},
    {"upconv3.1", "upconv3.2", 512, 256, 3, 1, 0, 0}, {"iconv3.0", "iconv3.1", 513, 256, 3, 1, 0, 0},
    {"upconv2.1", "upconv2.2", 256, 128, 3, 1, 0, 0}, {"iconv2.0", "iconv2.1", 257, 128, 3, 1, 0, 0},
    {"upconv1.1", "upconv1.2", 128, 64, 3, 1, 0, 0},  {"iconv1.0", "iconv1.1", 65, 64, 3, 1, 0, 0},
    {"disp4.0", nullptr, 512, 1, 3, 1, 0, 1}, {"disp3.0", nullptr, 256, 1, 3, 1, 0, 1},
    {"disp2.0", nullptr, 128, 1, 3, 1, 0, 1}, {"disp1.0", nullptr, 64, 1, 3, 1, 0, 1},
};
enum { D_CONV1_0, D_CONV1_3, D_CONV2_0, D_CONV2_3, D_CONV3_0, D_CONV3_3, D_CONV4_0, D_CONV4_3, D_CONV5_0, D_CONV5_3,
       D_UPCONV5, D_ICONV5, D_UPCONV4, D_ICONV4, D_UPCONV3, D_ICONV3, D_UPCONV2, D_ICONV2, D_UPCONV1, D_ICONV1,
       D_DISP4, D_DISP3, D_DISP2, D_DISP1, D_NUM };

static const cnm_layer_info kRefineLayers[] = {
    {"conv1.0", "conv1.1", 67, 128, 3, 1, 3, 0},   {"conv1.3", "conv1.4", 128, 128, 3, 2, 0, 0},
    {"conv2.0", "conv2.1", 128, 256, 3, 1, 0, 0},  {"conv2.3", "conv2.4", 256, 256, 3, 2, 0, 0},
    {"conv3.0", "conv3.1", 256, 512, 3, 1, 0, 0},  {"conv3.3", "conv3.4", 512, 512, 3, 2, 0, 0},
    {"upconv3_depth.1", "upconv3_depth.2", 512, 256, 3, 1, 0, 0}, {"iconv3_depth.0", "iconv3_depth.1", 512, 256, 3, 1, 0, 0},
    {"upconv2_depth.1", "upconv2_depth.2", 256, 128, 3, 1, 0, 0}, {"iconv2_depth.0", "iconv2_depth.1", 256, 128, 3, 1, 0, 0},
    {"upconv1_depth.1", "upconv1_depth.2", 128, 64, 3, 1, 0, 0},  {"iconv1_depth.0", "iconv1_depth.1", 64, 64, 3, 1, 0, 0},
    {"upconv3_prob.1", "upconv3_prob.2", 512, 256, 3, 1, 0, 0},   {"iconv3_prob.0", "iconv3_prob.1", 512, 256, 3, 1, 0, 0},
    {"upconv2_prob.1", "upconv2_prob.2", 256, 128, 3, 1, 0, 0},   {"iconv2_prob.0", "iconv2_prob.1", 256, 128, 3, 1, 0, 0},
    {"upconv1_prob.1", "upconv1_prob.2", 128, 64, 3, 1, 0, 0},    {"iconv1_prob.0", "iconv1_prob.1", 64, 64, 3, 1, 0, 0},
    {"disp_refine.0", nullptr, 64, 1, 3, 1, 0, 1}, {"prob.0", nullptr, 64, 1, 3, 1, 0, 1},
};
enum { R_CONV1_0, R_CONV1_3, R_CONV2_0, R_CONV2_3, R_CONV3_0, R_CONV3_3, R_BRANCH0, R_HEAD0 = 18, R_NUM = 20 };

extern "C" int cnm_abi_version(void) { return CNM_ABI_VERSION; }

extern "C" const char* cnm_status_string(int status) {
    switch (status) {
        case CNM_OK: return "ok";
        case CNM_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or unsupported parameter)";
        case CNM_ERR_BAD_SHAPE: return "image height and width must be multiples of 32";
        case CNM_ERR_BAD_SCALE: return "idepth_scale must be 2.0 or 3.0";
        case CNM_ERR_LAUNCH: return "HIP kernel launch failed";
        case CNM_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int cnm_net_num_layers(int net) {
    return net == CNM_NET_DEPTH ? D_NUM : net == CNM_NET_REFINE ? R_NUM : CNM_ERR_BAD_ARG;
}

extern "C" int cnm_net_layer(int net, int index, cnm_layer_info* info) {
    CNM_REQUIRE(info, CNM_ERR_BAD_ARG);
    const int n = cnm_net_num_layers(net);
    CNM_REQUIRE(n > 0 && index >= 0 && index < n, CNM_ERR_BAD_ARG);
    *info = (net == CNM_NET_DEPTH ? kDepthLayers : kRefineLayers)[index];
    return CNM_OK;
}

// ------------------------------------------------------------------ workspace carving
struct Carver {
    float* base; size_t used;
    float* take(size_t n) { float* p = base ? base + used : nullptr; used += (n + 63) & ~(size_t)63; return p; }
};

struct DepthBufs {
    float *hmkt, *TEX, *X0, *A1, *CAT2, *A2, *CAT3, *A3, *CAT4, *A4, *CAT5, *A5, *C5, *U5, *I5, *U4, *I4, *U3, *I3, *U2, *I2, *U1, *CAT1;
};

static size_t carve_depth(float* ws, int P, int H, int W, int D, DepthBufs* b) {
    Carver c{ws, 0};
    const size_t q = (size_t)P * H * W * 4;     // floats of one channel group at full resolution
    b->hmkt = c.take((size_t)P * 12);
    b->TEX = c.take((size_t)P * (H + 4) * (W + 4) * 4);
    b->X0 = c.take(q * (D / 4 + 1));
    b->A1 = c.take(q * 32);        b->U1 = c.take(q * 32);       b->CAT1 = c.take(q * 17);
    b->CAT2 = c.take(q / 4 * 65);  b->A2 = c.take(q / 4 * 64);   b->U2 = c.take(q / 4 * 64);   b->I2 = c.take(q / 4 * 32);
    b->CAT3 = c.take(q / 16 * 129); b->A3 = c.take(q / 16 * 128); b->U3 = c.take(q / 16 * 128); b->I3 = c.take(q / 16 * 64);
    b->CAT4 = c.take(q / 64 * 256); b->A4 = c.take(q / 64 * 128); b->U4 = c.take(q / 64 * 128); b->I4 = c.take(q / 64 * 128);
    b->CAT5 = c.take(q / 256 * 256); b->A5 = c.take(q / 256 * 128); b->U5 = c.take(q / 256 * 128); b->I5 = c.take(q / 256 * 128);
    b->C5 = c.take(q / 1024 * 128);
    return c.used;
}

extern "C" size_t cnm_depthnet_workspace_floats(int P, int H, int W, int D) {
    if (P <= 0 || H <= 0 || W <= 0 || D < 4 || (H % 32) || (W % 32) || (D % 4)) return 0;
    DepthBufs b;
    return carve_depth(nullptr, P, H, W, D, &b);
}

#define CNM_TRY(expr) do { int _e = (expr); if (_e != CNM_OK) return _e; } while (0)

extern "C" int cnm_depthnet_forward_f32(const cnm_layer_weights* wt, float idepth_scale, int D,
                                        const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                                        float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                                        float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream) {
    CNM_REQUIRE(wt && ref && src && ref_cam && src_cam && disp1 && disp2 && disp3 && disp4 && iconv1_c4 && ws, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(B > 0 && S > 0 && D >= 4 && D % 4 == 0 && D <= 128, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, CNM_ERR_BAD_SHAPE);
    double idmin, idmax;
    CNM_TRY(cnm_idepth_range_host((double)idepth_scale, &idmin, &idmax));
    for (int i = 0; i < D_NUM; ++i) CNM_REQUIRE(wt[i].w && wt[i].b, CNM_ERR_BAD_ARG);
    const int P = B * S;
    DepthBufs b;
    CNM_REQUIRE(carve_depth(ws, P, H, W, D, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    const int G0 = D / 4 + 1;
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8, H4 = H / 16, W4 = W / 16, H5 = H / 32, W5 = W / 32;
    void* s = stream;
#define CONV(L, in, Gt, g0, Gin, out, Gto, go0, Cout, HH, WW) \
    CNM_TRY(cnm_conv2d_c4_f32(in, Gt, g0, Gin, out, Gto, go0, Cout, wt[L].w, wt[L].b, P, HH, WW, kDepthLayers[L].ksize, kDepthLayers[L].stride, 1, s))
    // geometry + cost volume                                                   depthNet_model.py:228-233
    CNM_TRY(cnm_homography_terms_f32(ref_cam, src_cam, b.hmkt, B, S, s));
    CNM_TRY(cnm_planesweep_cat_c4_f32(ref, src, b.hmkt, b.X0, b.TEX, (size_t)P * (H + 4) * (W + 4) * 4, B, S, H, W, D, idmin, idmax, s));
    // encoder                                                                  :235-239
    CONV(D_CONV1_0, b.X0, G0, 0, G0, b.A1, 32, 0, 128, H, W);
    CONV(D_CONV1_3, b.A1, 32, 0, 32, b.CAT2, 65, 32, 128, H, W);           // conv1 -> skip slot of iconv2
    CONV(D_CONV2_0, b.CAT2, 65, 32, 32, b.A2, 64, 0, 256, H1, W1);
    CONV(D_CONV2_3, b.A2, 64, 0, 64, b.CAT3, 129, 64, 256, H1, W1);        // conv2 -> skip slot of iconv3
    CONV(D_CONV3_0, b.CAT3, 129, 64, 64, b.A3, 128, 0, 512, H2, W2);
    CONV(D_CONV3_3, b.A3, 128, 0, 128, b.CAT4, 256, 128, 512, H2, W2);     // conv3 -> skip slot of iconv4
    CONV(D_CONV4_0, b.CAT4, 256, 128, 128, b.A4, 128, 0, 512, H3, W3);
    CONV(D_CONV4_3, b.A4, 128, 0, 128, b.CAT5, 256, 128, 512, H3, W3);     // conv4 -> skip slot of iconv5
    CONV(D_CONV5_0, b.CAT5, 256, 128, 128, b.A5, 128, 0, 512, H4, W4);
    CONV(D_CONV5_3, b.A5, 128, 0, 128, b.C5, 128, 0, 512, H4, W4);
    // decoder                                                                  :241-261
    CNM_TRY(cnm_upsample2x_c4_f32(b.C5, 128, 0, b.U5, 128, 0, P, 128, H5, W5, s));
    CONV(D_UPCONV5, b.U5, 128, 0, 128, b.CAT5, 256, 0, 512, H4, W4);
    CONV(D_ICONV5, b.CAT5, 256, 0, 256, b.I5, 128, 0, 512, H4, W4);
    CNM_TRY(cnm_upsample2x_c4_f32(b.I5, 128, 0, b.U4, 128, 0, P, 128, H4, W4, s));
    CONV(D_UPCONV4, b.U4, 128, 0, 128, b.CAT4, 256, 0, 512, H3, W3);
    CONV(D_ICONV4, b.CAT4, 256, 0, 256, b.I4, 128, 0, 512, H3, W3);
    CNM_TRY(cnm_head_sigmoid_c4_f32(b.I4, 128, 0, 512, wt[D_DISP4].w, wt[D_DISP4].b, idepth_scale, disp4, b.CAT3, 129, 128, P, H3, W3, s));
    CNM_TRY(cnm_upsample2x_c4_f32(b.I4, 128, 0, b.U3, 128, 0, P, 128, H3, W3, s));
    CONV(D_UPCONV3, b.U3, 128, 0, 128, b.CAT3, 129, 0, 256, H2, W2);
    CONV(D_ICONV3, b.CAT3, 129, 0, 129, b.I3, 64, 0, 256, H2, W2);
    CNM_TRY(cnm_head_sigmoid_c4_f32(b.I3, 64, 0, 256, wt[D_DISP3].w, wt[D_DISP3].b, idepth_scale, disp3, b.CAT2, 65, 64, P, H2, W2, s));
    CNM_TRY(cnm_upsample2x_c4_f32(b.I3, 64, 0, b.U2, 64, 0, P, 64, H2, W2, s));
    CONV(D_UPCONV2, b.U2, 64, 0, 64, b.CAT2, 65, 0, 128, H1, W1);
    CONV(D_ICONV2, b.CAT2, 65, 0, 65, b.I2, 32, 0, 128, H1, W1);
    CNM_TRY(cnm_head_sigmoid_c4_f32(b.I2, 32, 0, 128, wt[D_DISP2].w, wt[D_DISP2].b, idepth_scale, disp2, b.CAT1, 17, 16, P, H1, W1, s));
    CNM_TRY(cnm_upsample2x_c4_f32(b.I2, 32, 0, b.U1, 32, 0, P, 32, H1, W1, s));
    CONV(D_UPCONV1, b.U1, 32, 0, 32, b.CAT1, 17, 0, 64, H, W);
    CONV(D_ICONV1, b.CAT1, 17, 0, 17, iconv1_c4, 16, 0, 64, H, W);
    CNM_TRY(cnm_head_sigmoid_c4_f32(iconv1_c4, 16, 0, 64, wt[D_DISP1].w, wt[D_DISP1].b, idepth_scale, disp1, nullptr, 0, 0, P, H, W, s));
#undef CONV
    return CNM_OK;
}

// ------------------------------------------------------------------ refine net
struct RefineBufs { float *X, *A1, *C1, *A2, *C2, *A3, *C3, *U3, *UC3, *I3, *U2, *UC2, *I2, *U1, *UC1, *I1; };

static size_t carve_refine(float* ws, int N, int H, int W, RefineBufs* b) {
    Carver c{ws, 0};
    const size_t q = (size_t)N * H * W * 4;
    b->X = c.take(q * 17); b->A1 = c.take(q * 32); b->U1 = c.take(q * 32); b->UC1 = c.take(q * 16); b->I1 = c.take(q * 16);
    b->C1 = c.take(q / 4 * 32); b->A2 = c.take(q / 4 * 64); b->U2 = c.take(q / 4 * 64); b->UC2 = c.take(q / 4 * 32); b->I2 = c.take(q / 4 * 32);
    b->C2 = c.take(q / 16 * 64); b->A3 = c.take(q / 16 * 128); b->U3 = c.take(q / 16 * 128); b->UC3 = c.take(q / 16 * 64); b->I3 = c.take(q / 16 * 64);
    b->C3 = c.take(q / 64 * 128);
    return c.used;
}

extern "C" size_t cnm_refinenet_workspace_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
    RefineBufs b;
    return carve_refine(nullptr, N, H, W, &b);
}

static int refinenet_body(const cnm_layer_weights* wt, float idepth_scale, const RefineBufs& b,
                          float* disp_refined, float* prob_map, float* iconv1_depth_c4, int N, int H, int W, void* s) {
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8;
#define CONV(L, in, Gt, g0, Gin, out, Gto, go0, Cout, HH, WW) \
    CNM_TRY(cnm_conv2d_c4_f32(in, Gt, g0, Gin, out, Gto, go0, Cout, wt[L].w, wt[L].b, N, HH, WW, 3, kRefineLayers[L].stride, 1, s))
#define CONV2(L, ina, Ga, inb, Gb, out, Gto, Cout, HH, WW) \
    CNM_TRY(cnm_conv2d_cat2_c4_f32(ina, Ga, 0, Ga, inb, Gb, 0, Gb, out, Gto, 0, Cout, wt[L].w, wt[L].b, N, HH, WW, 3, 1, 1, s))
    CONV(R_CONV1_0, b.X, 17, 0, 17, b.A1, 32, 0, 128, H, W);
    CONV(R_CONV1_3, b.A1, 32, 0, 32, b.C1, 32, 0, 128, H, W);
    CONV(R_CONV2_0, b.C1, 32, 0, 32, b.A2, 64, 0, 256, H1, W1);
    CONV(R_CONV2_3, b.A2, 64, 0, 64, b.C2, 64, 0, 256, H1, W1);
    CONV(R_CONV3_0, b.C2, 64, 0, 64, b.A3, 128, 0, 512, H2, W2);
    CONV(R_CONV3_3, b.A3, 128, 0, 128, b.C3, 128, 0, 512, H2, W2);
    CNM_TRY(cnm_upsample2x_c4_f32(b.C3, 128, 0, b.U3, 128, 0, N, 128, H3, W3, s));     // shared by both decoders
    for (int br = 0; br < 2; ++br) {                                                   // 0: depth (:341-351), 1: prob (:357-365)
        const int L = R_BRANCH0 + 6 * br;
        float* feat = (br == 0 && iconv1_depth_c4) ? iconv1_depth_c4 : b.I1;
        CONV(L + 0, b.U3, 128, 0, 128, b.UC3, 64, 0, 256, H2, W2);
        CONV2(L + 1, b.UC3, 64, b.C2, 64, b.I3, 64, 256, H2, W2);
        CNM_TRY(cnm_upsample2x_c4_f32(b.I3, 64, 0, b.U2, 64, 0, N, 64, H2, W2, s));
        CONV(L + 2, b.U2, 64, 0, 64, b.UC2, 32, 0, 128, H1, W1);
        CONV2(L + 3, b.UC2, 32, b.C1, 32, b.I2, 32, 128, H1, W1);
        CNM_TRY(cnm_upsample2x_c4_f32(b.I2, 32, 0, b.U1, 32, 0, N, 32, H1, W1, s));
        CONV(L + 4, b.U1, 32, 0, 32, b.UC1, 16, 0, 64, H, W);
        CONV(L + 5, b.UC1, 16, 0, 16, feat, 16, 0, 64, H, W);
        CNM_TRY(cnm_head_sigmoid_c4_f32(feat, 16, 0, 64, wt[R_HEAD0 + br].w, wt[R_HEAD0 + br].b,
                                        br == 0 ? idepth_scale : 1.0f, br == 0 ? disp_refined : prob_map,
                                        nullptr, 0, 0, N, H, W, s));
    }
#undef CONV
#undef CONV2
    return CNM_OK;
}

extern "C" int cnm_refinenet_forward_f32(const cnm_layer_weights* wt, float idepth_scale,
                                         const float* idepth01, const float* idepth02, long long idepth_stride,
                                         const float* iconv01, int G1_total, int g1,
                                         const float* iconv02, int G2_total, int g2,
                                         float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                         float* ws, size_t ws_floats, int N, int H, int W, void* stream) {
    CNM_REQUIRE(wt && idepth01 && idepth02 && iconv01 && iconv02 && disp_refined && prob_map && ws && N > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, CNM_ERR_BAD_SHAPE);
    for (int i = 0; i < R_NUM; ++i) CNM_REQUIRE(wt[i].w && wt[i].b, CNM_ERR_BAD_ARG);
    RefineBufs b;
    CNM_REQUIRE(carve_refine(ws, N, H, W, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    CNM_TRY(cnm_refine_assemble_c4_f32(idepth01, idepth02, idepth_stride, iconv01, G1_total, g1, iconv02, G2_total, g2, b.X, N, 64, H, W, stream));  // :332-333
    return refinenet_body(wt, idepth_scale, b, disp_refined, prob_map, iconv1_depth_c4, N, H, W, stream);
}

extern "C" int cnm_refinenet_forward_multi_f32(const cnm_layer_weights* wt, float idepth_scale,
                                               const float* idepth_pairs, const float* iconv_pairs_c4, int S,
                                               float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                               float* ws, size_t ws_floats, int B, int H, int W, void* stream) {
    CNM_REQUIRE(wt && idepth_pairs && iconv_pairs_c4 && disp_refined && prob_map && ws && B > 0 && S >= 2 && S % 2 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, CNM_ERR_BAD_SHAPE);
    for (int i = 0; i < R_NUM; ++i) CNM_REQUIRE(wt[i].w && wt[i].b, CNM_ERR_BAD_ARG);
    RefineBufs b;
    CNM_REQUIRE(carve_refine(ws, B, H, W, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    CNM_TRY(cnm_refine_assemble_multi_c4_f32(idepth_pairs, iconv_pairs_c4, b.X, B, S, 64, H, W, stream));
    return refinenet_body(wt, idepth_scale, b, disp_refined, prob_map, iconv1_depth_c4, B, H, W, stream);
}
