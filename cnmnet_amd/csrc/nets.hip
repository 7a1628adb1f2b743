// Whole-network executors: depthNet.forward and DepthRefineNet.forward
// (reference depthnet/depthNet_model.py:226-263 and :331-370) as fixed launch sequences
// over a caller-owned workspace.  No allocation, no host synchronisation: everything is ordered on the caller's stream
// (the refine net's second decoder forks onto a side stream and joins back with events), so a forward can be captured
// in a hipGraph.  Process-wide state is limited to tuning knobs and the per-thread side streams.
//
// Concatenations are never copied: a producer writes straight into the channel-group slice
// of its consumer's input buffer (CAT* below), or the consumer reads two views (cat2 conv).
#include <string.h>
#include "cnm_common.h"
#include "host_ops.h"
#include "wino4_args.h"

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
        case CNM_ERR_LAUNCH: return "HIP kernel launch failed, or a stream-K hand-off timed out inside an earlier launch (cnm_engine_status)";
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

// ------------------------------------------------------------------ engines (fp32 c4 / fp16 c8)
// Both activation layouts use 16 bytes per (pixel, channel group); a policy supplies the group width and the
// operator entry points, the launch sequences below are shared.
extern "C" {
int cnm_conv2d_c8_f16(const void*, int, int, int, void*, int, int, int, const void*, const float*, int, int, int, int, int, int, void*);
int cnm_conv2d_cat2_c8_f16(const void*, int, int, int, const void*, int, int, int, void*, int, int, int, const void*, const float*, int, int, int, int, int, int, void*);
int cnm_upsample2x_c8_f16(const void*, int, int, void*, int, int, int, int, int, int, void*);
int cnm_head_sigmoid_c8_f16(const void*, int, int, int, const float*, const float*, float, float*, void*, int, int, int, int, int, void*);
int cnm_refine_assemble_multi_c8_f16(const float*, const void*, void*, int, int, int, int, int, void*);
int cnm_planesweep_cat_c8_f16(const float*, const float*, const float*, void*, float*, size_t, int, int, int, int, int, double, double, void*);
int cnm_conv3x3_upsampled_c8_f16(const void*, int, int, int, void*, int, int, int, const void*, const float*, int, int, int, int, int, void*);
int cnm_conv3x3_upsampled_ring_c8_f16(const void*, int, int, int, void*, int, int, int, const float*, const float*, int, int, int, int, void*);
}

// F(4x4,3x3) workgroups cover 64 couts x 16 tiles of 4x4 outputs: worth it only when they fill the chip twice over
static int g_wino4_min_workgroups = CNM_WINO4_MIN_WORKGROUPS;         // the engine's only process-wide state: a tuning knob
extern "C" int cnm_tune_wino4_min_workgroups(int n) { const int old = g_wino4_min_workgroups; if (n > 0) g_wino4_min_workgroups = n; return old; }
static inline bool wino4_fills_chip(int Cout, int N, int H, int W, int m = 4) {   // m x m outputs per tile
    const long long tiles = (long long)N * ((H + m - 1) / m) * ((W + m - 1) / m);
    return (Cout / 64) * ((tiles + 15) / 16) >= g_wino4_min_workgroups;
}

static int g_wino4_s2 = 1;                                               // tuning knob: 5x5 / 7x7 stride-2 layers on the staged 36-point kernel (pixel phases) instead of the row-wise phase kernel
extern "C" int cnm_tune_wino4_s2(int on) { const int old = g_wino4_s2; if (on == 0 || on == 1) g_wino4_s2 = on; return old; }
static int g_wino4_small = 1;                                            // tuning knob: small layers on the staged F(4x4,3x3) kernel (A/B against F(2x2,3x3))
extern "C" int cnm_tune_wino4_small(int on) { const int old = g_wino4_small; if (on == 0 || on == 1) g_wino4_small = on; return old; }

// The depth and the probability decoder (depthNet_model.py:341-351 / :357-365) share only their inputs, and their
// layers launch 384-768 workgroups on 512 slots; run side by side they backfill each other's tail and the bandwidth-
// bound upsample / head launches hide under the other decoder's MFMA work.  The second decoder runs on a side stream
// forked from and joined back into the caller's stream with events, so callers still see one ordered stream (and a
// hipGraph capture of the caller's stream records both).  One side stream per (host thread, device), created on
// first use and kept.  (Tried and dropped: depthNet's three intermediate heads on the side stream, -0.05 ms.)
struct SideStream { hipStream_t stream; hipEvent_t fork, join; bool ok; };
static int g_refine_side_stream = 1;
extern "C" int cnm_tune_refine_side_stream(int on) { const int old = g_refine_side_stream; if (on >= 0) g_refine_side_stream = on ? 1 : 0; return old; }
static SideStream* side_stream() {
    constexpr int kMaxDev = 64;
    thread_local SideStream tl[kMaxDev] = {};
    thread_local bool tried[kMaxDev] = {};
    int dev = 0;
    if (!g_refine_side_stream || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    SideStream& x = tl[dev];
    if (!tried[dev]) {
        tried[dev] = true;
        x.ok = hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
        if (!x.ok) (void)hipGetLastError();
    }
    return x.ok ? &x : nullptr;
}

// up_conv layers: one fused pass over the low-resolution input when the upsampled tensor is big and the ring is cheap
static int g_upsampled_min_pixels = CNM_UPSAMPLED_MIN_PIXELS;
extern "C" int cnm_tune_upsampled_min_pixels(int n) { const int old = g_upsampled_min_pixels; if (n > 0) g_upsampled_min_pixels = n; return old; }
// [r6] the fp16 engine's own threshold: depthNet's upconv2 (256 -> 128 channels, 16 x 96 x 128 output pixels) is faster UNFUSED there (upsample + the
// row-extended implicit GEMM: 0.166 against 0.199 ms; bench step 1863-1867 against 1843-1855 frames/s, tools/ups_threshold_ab.sh); the fp32 engine is indifferent
static int g_upsampled_min_pixels_f16 = CNM_UPSAMPLED_MIN_PIXELS_F16;
extern "C" int cnm_tune_upsampled_min_pixels_f16(int n) { const int old = g_upsampled_min_pixels_f16; if (n > 0) g_upsampled_min_pixels_f16 = n; return old; }

// [r6] floats between consecutive frames of the four inputs (0 = dense): ref = frames[:, 0], src = frames[:, 1:] of one tensor are read where they lie
struct InStrides { long long ref, src, ref_cam, src_cam; };
extern "C" {
int cnm_homography_terms_strided_f32(const float*, long long, const float*, long long, float*, int, int, void*);
int cnm_planesweep_cat_strided_c4_f32(const float*, long long, const float*, long long, const float*, float*, float*, size_t, int, int, int, int, int, double, double, void*);
int cnm_planesweep_cat_strided_c8_f16(const float*, long long, const float*, long long, const float*, void*, float*, size_t, int, int, int, int, int, double, double, void*);
}

struct EngF32 {
    static constexpr int GD = 4;
    static constexpr bool HOST = false;
    static int homography(const float* rc, const float* sc, float* hmkt, int B, int S, void* s, const InStrides& st) { return cnm_homography_terms_strided_f32(rc, st.ref_cam, sc, st.src_cam, hmkt, B, S, s); }
    static int assemble(const float* a, const float* b, long long st, const float* f1, int G1, int g1, const float* f2, int G2, int g2, float* x, int N, int H, int W, void* s) {
        return cnm_refine_assemble_c4_f32(a, b, st, f1, G1, g1, f2, G2, g2, x, N, 64, H, W, s); }
    // nn.Upsample(2, bilinear) + conv3x3 + BN + ReLU (up_conv_layer, depthNet_model.py:89-112): in [N][G][H][W] -> out at 2H x 2W
    // sync: the stream's sync workspace for the LDS-staged F(4x4,3x3) kernel (cnm_wino36_sync_floats() floats), or null
    static int upconv(const float* in, int G, float* up_tmp, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, float* sync, void* s) {
        if (w.uu && w.bu && w.wr && G * 4 <= 256 && (long long)N * 4 * H * W >= g_upsampled_min_pixels) {
            const int e = cnm_conv3x3_upsampled_winograd4_sync_c4_f32(in, G, 0, G, out, Gto, go0, Cout, w.uu, w.bu, N, H, W, 1, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
            return e != CNM_OK ? e : cnm_conv3x3_upsampled_ring_c4_f32(in, G, 0, G, out, Gto, go0, Cout, w.wr, w.b, N, H, W, 1, s);
        }
        const int e = up(in, G, up_tmp, N, H, W, s);
        return e != CNM_OK ? e : conv(up_tmp, G, 0, G, out, Gto, go0, Cout, w, N, 2 * H, 2 * W, 3, 1, sync, s);
    }
    // with a sync workspace the staged F(4x4,3x3) kernel spreads ANY number of units evenly over the CUs, so it also takes the
    // small layers (down to 3 x 3 tiles per image) that the gather-fed kernel could not fill the chip with
    static bool wino4_staged_small(int Cout, int H, int W, const float* sync) { return sync && g_wino4_small && Cout % 128 == 0 && (W + 3) / 4 >= 3 && (H + 3) / 4 >= 3; }
    // [r6] the four-wave F(4x4,3x3) kernel (64 output channels x 32 tiles per workgroup): where the eight-wave kernel cannot go (Cout = 64: the three
    // iconv1 layers, which ran the gather-fed kernel at 0.42-0.46 of the matrix roof); on the Cout % 128 == 0 layers it was measured 10-25 % slower
    // than the eight-wave kernel (profiles/r6_wino36q_probe*.txt), so mode 1 leaves those alone
    static bool wino4_quad(const cnm_layer_weights& w, int Cout, int H, int W) {
        const int mode = cnm_wino36_quad_mode();
        return w.u4q && mode && (mode == 2 || Cout % 128 != 0) && cnm_conv3x3_winograd4q_ok(Cout, H, W);
    }
    static int conv(const float* in, int Gt, int g0, int Gin, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, int k, int st, float* sync, void* s) {
        if (k == 3 && st == 1 && wino4_quad(w, Cout, H, W))
            return cnm_conv3x3_winograd4q_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u4q, w.b, N, H, W, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        if (w.u4 && k == 3 && st == 1 && (wino4_fills_chip(Cout, N, H, W) || wino4_staged_small(Cout, H, W, sync)))
            return cnm_conv3x3_winograd4_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u4, w.b, N, H, W, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        if (w.u && k == 3 && st == 2 && (Cout / 64) * (((long long)N * ((H + 1) / 2) * ((W + 1) / 2) + 63) / 64) < 256)   // too few implicit-GEMM tiles
            return cnm_conv3x3_s2_winograd_c4_f32(in, Gt, g0, Gin, out, Gto, go0, Cout, w.u, w.b, N, H, W, 1, s);
        if (w.u4 && k == 3 && st == 2 && (Cout / 64) * (((long long)N * ((H + 1) / 2) * (((W + 1) / 2 + 3) / 4) + 47) / 48) >= 384)   // F(4,2) column phases along rows: 1.2x fewer multiplies; pays from ~1.5 workgroups per CU slot pair
            return cnm_conv_rows_winograd_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u4, w.b, N, H, W, 3, 2, 4, 1, s);
        if (w.uu && !w.bu && k == 3 && st == 2 && sync && g_wino4_s2 && cnm_conv_s2_winograd4_ok(Cout, H, W, 3))   // what is left of the 3x3 stride-2 layers: the pixel phases on the staged kernel (the direct count, at its efficiency) instead of the implicit GEMM
            return cnm_conv_s2_winograd4_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.uu, w.b, N, H, W, 3, 1, sync, cnm_wino36_sync_floats(), s);
        if (w.u && k == 3 && st == 1) return cnm_conv3x3_winograd_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u, w.b, N, H, W, 1, s);
        if (w.u4 && k == 5 && st == 1 && wino4_fills_chip(Cout, N, H, W, 2))
            return cnm_conv5x5_winograd_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u4, w.b, N, H, W, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        if (w.u4 && (k == 5 || k == 7) && st == 2 && sync && g_wino4_s2 && cnm_conv_s2_winograd4_ok(Cout, H, W, k))   // four pixel phases on the staged 36-point kernel
            return cnm_conv_s2_winograd4_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u4, w.b, N, H, W, k, 1, sync, cnm_wino36_sync_floats(), s);
        if (w.u && (k == 5 || k == 7)) return cnm_conv_rows_winograd_sync_c4_f32(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.u, w.b, N, H, W, k, st, (k == 5 && st == 1) ? 2 : 4, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        return cnm_conv2d_c4_f32(in, Gt, g0, Gin, out, Gto, go0, Cout, w.w, w.b, N, H, W, k, st, 1, s); }
    static int conv2(const float* a, int Ga, const float* b, int Gb, float* out, int Gto, int Cout, const cnm_layer_weights& w, int N, int H, int W, float* sync, void* s) {
        if (wino4_quad(w, Cout, H, W))
            return cnm_conv3x3_winograd4q_sync_c4_f32(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.u4q, w.b, N, H, W, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        if (w.u4 && (wino4_fills_chip(Cout, N, H, W) || wino4_staged_small(Cout, H, W, sync))) return cnm_conv3x3_winograd4_sync_c4_f32(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.u4, w.b, N, H, W, 1, sync, sync ? cnm_wino36_sync_floats() : 0, s);
        if (w.u) return cnm_conv3x3_winograd_c4_f32(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.u, w.b, N, H, W, 1, s);
        return cnm_conv2d_cat2_c4_f32(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.w, w.b, N, H, W, 3, 1, 1, s); }
    static int up(const float* in, int G, float* out, int N, int H, int W, void* s) { return cnm_upsample2x_c4_f32(in, G, 0, out, G, 0, N, G, H, W, s); }
    static int head(const float* in, int G, int C, const cnm_layer_weights& w, float scale, float* disp, float* up, int upG, int upg, int N, int H, int W, void* s) {
        return cnm_head_sigmoid_c4_f32(in, G, 0, C, w.w, w.b, scale, disp, up, upG, upg, N, H, W, s); }
    static int assemble_multi(const float* idp, const float* f, float* x, int B, int S, int H, int W, void* s) { return cnm_refine_assemble_multi_c4_f32(idp, f, x, B, S, 64, H, W, s); }
    static int sweep(const float* ref, const float* src, const float* hmkt, float* x, float* tex, size_t texn, int B, int S, int H, int W, int D, double lo, double hi, void* s, const InStrides& st) {
        return cnm_planesweep_cat_strided_c4_f32(ref, st.ref, src, st.src, hmkt, x, tex, texn, B, S, H, W, D, lo, hi, s); }
};

struct EngF16 {
    static constexpr int GD = 8;
    static constexpr bool HOST = false;
    static int homography(const float* rc, const float* sc, float* hmkt, int B, int S, void* s, const InStrides& st) { return cnm_homography_terms_strided_f32(rc, st.ref_cam, sc, st.src_cam, hmkt, B, S, s); }
    static int upconv(const float* in, int G, float* up_tmp, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, float* sync, void* s) {
        if (w.uu && w.bu && w.wr && G * 8 <= 256 && (long long)N * 4 * H * W >= g_upsampled_min_pixels_f16) {   // one fused pass over the low-resolution input + ring pass
            const int e = cnm_conv3x3_upsampled_c8_f16(in, G, 0, G, out, Gto, go0, Cout, w.uu, w.bu, N, H, W, 1, 1, s);
            return e != CNM_OK ? e : cnm_conv3x3_upsampled_ring_c8_f16(in, G, 0, G, out, Gto, go0, Cout, w.wr, w.b, N, H, W, 1, s);
        }
        const int e = up(in, G, up_tmp, N, H, W, s);
        return e != CNM_OK ? e : conv(up_tmp, G, 0, G, out, Gto, go0, Cout, w, N, 2 * H, 2 * W, 3, 1, sync, s);
    }
    static int conv(const float* in, int Gt, int g0, int Gin, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, int k, int st, float*, void* s) {
        return cnm_conv2d_c8_f16(in, Gt, g0, Gin, out, Gto, go0, Cout, w.w, w.b, N, H, W, k, st, 1, s); }
    static int conv2(const float* a, int Ga, const float* b, int Gb, float* out, int Gto, int Cout, const cnm_layer_weights& w, int N, int H, int W, float*, void* s) {
        return cnm_conv2d_cat2_c8_f16(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.w, w.b, N, H, W, 3, 1, 1, s); }
    static int up(const float* in, int G, float* out, int N, int H, int W, void* s) { return cnm_upsample2x_c8_f16(in, G, 0, out, G, 0, N, G, H, W, s); }
    static int head(const float* in, int G, int C, const cnm_layer_weights& w, float scale, float* disp, float* up, int upG, int upg, int N, int H, int W, void* s) {
        return cnm_head_sigmoid_c8_f16(in, G, 0, C, w.w, w.b, scale, disp, up, upG, upg, N, H, W, s); }
    static int assemble_multi(const float* idp, const float* f, float* x, int B, int S, int H, int W, void* s) { return cnm_refine_assemble_multi_c8_f16(idp, f, x, B, S, 64, H, W, s); }
    static int sweep(const float* ref, const float* src, const float* hmkt, float* x, float* tex, size_t texn, int B, int S, int H, int W, int D, double lo, double hi, void* s, const InStrides& st) {
        return cnm_planesweep_cat_strided_c8_f16(ref, st.ref, src, st.src, hmkt, x, tex, texn, B, S, H, W, D, lo, hi, s); }
};

// Host twin of the fp32 engine (host_twins.cpp): the same launch sequences on HOST pointers -- direct convolutions with the
// BatchNorm-folded filter of cnm_pack_conv_bn_cpu in cnm_layer_weights.w, no Winograd forms, no streams.  BASELINE configs[0].
struct EngHost {
    static constexpr int GD = 4;
    static constexpr bool HOST = true;
    static int homography(const float* rc, const float* sc, float* hmkt, int B, int S, void*, const InStrides&) { return cnmh::homography(rc, sc, hmkt, B, S); }   // dense inputs only
    static int up(const float* in, int G, float* out, int N, int H, int W, void*) { return cnmh::upsample2x(in, G, 0, out, G, 0, N, G, H, W); }
    static int conv(const float* in, int Gt, int g0, int Gin, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, int k, int st, float*, void*) {
        return cnmh::conv(in, Gt, g0, Gin, nullptr, 0, 0, 0, out, Gto, go0, Cout, w.w, w.b, N, H, W, k, st, 1); }
    static int conv2(const float* a, int Ga, const float* b, int Gb, float* out, int Gto, int Cout, const cnm_layer_weights& w, int N, int H, int W, float*, void*) {
        return cnmh::conv(a, Ga, 0, Ga, b, Gb, 0, Gb, out, Gto, 0, Cout, w.w, w.b, N, H, W, 3, 1, 1); }
    static int upconv(const float* in, int G, float* up_tmp, float* out, int Gto, int go0, int Cout, const cnm_layer_weights& w, int N, int H, int W, float*, void* s) {
        const int e = up(in, G, up_tmp, N, H, W, s);
        return e != CNM_OK ? e : conv(up_tmp, G, 0, G, out, Gto, go0, Cout, w, N, 2 * H, 2 * W, 3, 1, nullptr, s); }
    static int head(const float* in, int G, int C, const cnm_layer_weights& w, float scale, float* disp, float* up, int upG, int upg, int N, int H, int W, void*) {
        return cnmh::head(in, G, 0, C, w.w, w.b, scale, disp, up, upG, upg, N, H, W); }
    static int assemble(const float* a, const float* b, long long st, const float* f1, int G1, int g1, const float* f2, int G2, int g2, float* x, int N, int H, int W, void*) {
        return cnmh::assemble(a, b, st, f1, G1, g1, f2, G2, g2, x, N, 64, H, W); }
    static int assemble_multi(const float* idp, const float* f, float* x, int B, int S, int H, int W, void*) { return cnmh::assemble_multi(idp, f, x, B, S, 64, H, W); }
    static int sweep(const float* ref, const float* src, const float* hmkt, float* x, float*, size_t, int B, int S, int H, int W, int D, double lo, double hi, void*, const InStrides&) {
        return cnmh::sweep(ref, src, hmkt, x, B, S, H, W, D, lo, hi, 0); }
};

#define CNM_TRY(expr) do { int _e = (expr); if (_e != CNM_OK) return _e; } while (0)

// ------------------------------------------------------------------ workspace carving
struct Carver {
    float* base; size_t used;
    float* take(size_t n) { float* p = base ? base + used : nullptr; used += (n + 63) & ~(size_t)63; return p; }
};

struct DepthBufs {
    float *hmkt, *TEX, *SYNC, *X0, *A1, *CAT2, *A2, *CAT3, *A3, *CAT4, *A4, *CAT5, *A5, *C5, *U5, *I5, *U4, *I4, *U3, *I3, *U2, *I2, *U1, *CAT1;
};

// Buffers share memory by liveness (launch order of depthnet_forward, one stream):
//   X0, A1            sweep .. conv1.3                      the encoder's full-resolution level
//   A2 .. A5          one encoder level each (after A1 is dead)
//   U5 .. U2, I5..I3  decoder temporaries: all dead before disp2 / upconv1 write CAT1 / U1
//   U1, CAT1          upconv1 .. iconv1                     the decoder's full-resolution level
// so CAT1 lives where X0 was, U1 where A1 was, and the low-resolution temporaries (a level's A, U and I share one
// slot: A is dead when U is written, U when I is) sit inside the A1 / U1 slot.  Skip tensors (CAT2..CAT5), I2 (read
// while U1 / CAT1 are written) and the plane sweep's tile queue keep memory of their own: 1.1 GB instead of 2.4 GB
// for 16 pairs at 192x256.
template <class E>
static size_t carve_depth(float* ws, int P, int H, int W, int D, DepthBufs* b) {
    Carver c{ws, 0};
    const size_t q = (size_t)P * H * W * 4;     // floats (= 16 bytes x pixels) of one channel group at full resolution
    auto G = [](int C) { return (size_t)((C + E::GD - 1) / E::GD); };
    auto up64 = [](size_t n) { return (n + 63) & ~(size_t)63; };
    b->TEX = c.take(4);                         // FIRST and never reused: the plane sweep's tile queue, zero between calls
    b->SYNC = (E::GD == 4 && !E::HOST) ? c.take(cnm_wino36_sync_floats()) : nullptr;   // second, at a fixed offset, never reused: flag words (zero between calls) + partial-output slots of the staged F(4x4,3x3) kernel
    b->hmkt = c.take((size_t)P * 12);
    const size_t x0 = q * (G(D) + 1), cat1 = q * (G(64) + 1);
    b->X0 = b->CAT1 = c.take(x0 > cat1 ? x0 : cat1);
    const size_t lvl[4] = {up64(q / 4 * G(256)), up64(q / 16 * G(512)), up64(q / 64 * G(512)), up64(q / 256 * G(512))};   // A2..A5
    const size_t a1 = q * G(128), inner = lvl[0] + lvl[1] + lvl[2] + lvl[3];
    float* slot = c.take(a1 > inner ? a1 : inner);
    b->A1 = b->U1 = slot;
    size_t o = 0;
    b->A2 = b->U2 = slot ? slot + o : nullptr; o += lvl[0];
    b->A3 = b->U3 = b->I3 = slot ? slot + o : nullptr; o += lvl[1];
    b->A4 = b->U4 = b->I4 = slot ? slot + o : nullptr; o += lvl[2];
    b->A5 = b->U5 = b->I5 = slot ? slot + o : nullptr;
    b->I2 = c.take(q / 4 * G(128));
    b->CAT2 = c.take(q / 4 * (2 * G(128) + 1));
    b->CAT3 = c.take(q / 16 * (2 * G(256) + 1));
    b->CAT4 = c.take(q / 64 * 2 * G(512));
    b->CAT5 = c.take(q / 256 * 2 * G(512));
    b->C5 = c.take(q / 1024 * G(512));
    return c.used;
}

extern "C" size_t cnm_depthnet_workspace_floats(int P, int H, int W, int D) {
    if (P <= 0 || H <= 0 || W <= 0 || D < 4 || (H % 32) || (W % 32) || (D % 4)) return 0;
    DepthBufs b;
    return carve_depth<EngF32>(nullptr, P, H, W, D, &b);
}

extern "C" size_t cnm_depthnet_workspace_floats_f16(int P, int H, int W, int D) {
    if (P <= 0 || H <= 0 || W <= 0 || D < 8 || (H % 32) || (W % 32) || (D % 8)) return 0;
    DepthBufs b;
    return carve_depth<EngF16>(nullptr, P, H, W, D, &b);
}

template <class E>
static int depthnet_forward(const cnm_layer_weights* wt, float idepth_scale, int D,
                            const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                            float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1,
                            float* ws, size_t ws_floats, int B, int S, int H, int W, void* s, const InStrides& strides = InStrides{0, 0, 0, 0}) {
    CNM_REQUIRE(wt && ref && src && ref_cam && src_cam && disp1 && disp2 && disp3 && disp4 && iconv1 && ws, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(B > 0 && S > 0 && D >= E::GD && D % E::GD == 0 && D <= 128, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, CNM_ERR_BAD_SHAPE);
    double idmin, idmax;
    CNM_TRY(cnm_idepth_range_host((double)idepth_scale, &idmin, &idmax));
    for (int i = 0; i < D_NUM; ++i) CNM_REQUIRE((wt[i].w || wt[i].u || wt[i].u4) && wt[i].b, CNM_ERR_BAD_ARG);
    const int P = B * S;
    DepthBufs b;
    CNM_REQUIRE(carve_depth<E>(ws, P, H, W, D, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    auto G = [](int C) { return (C + E::GD - 1) / E::GD; };
    const int G0 = G(D) + 1, g64 = G(64), g128 = G(128), g256 = G(256), g512 = G(512);
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8, H4 = H / 16, W4 = W / 16, H5 = H / 32, W5 = W / 32;
#define CONV(L, in, Gt, g0, Gin, out, Gto, go0, Cout, HH, WW) \
    CNM_TRY(E::conv(in, Gt, g0, Gin, out, Gto, go0, Cout, wt[L], P, HH, WW, kDepthLayers[L].ksize, kDepthLayers[L].stride, b.SYNC, s))
    // geometry + cost volume                                                   depthNet_model.py:228-233
    CNM_TRY(E::homography(ref_cam, src_cam, b.hmkt, B, S, s, strides));
    CNM_TRY(E::sweep(ref, src, b.hmkt, b.X0, b.TEX, 4, B, S, H, W, D, idmin, idmax, s, strides));
    // encoder                                                                  :235-239
    CONV(D_CONV1_0, b.X0, G0, 0, G0, b.A1, g128, 0, 128, H, W);
    CONV(D_CONV1_3, b.A1, g128, 0, g128, b.CAT2, 2 * g128 + 1, g128, 128, H, W);              // conv1 -> skip slot of iconv2
    CONV(D_CONV2_0, b.CAT2, 2 * g128 + 1, g128, g128, b.A2, g256, 0, 256, H1, W1);
    CONV(D_CONV2_3, b.A2, g256, 0, g256, b.CAT3, 2 * g256 + 1, g256, 256, H1, W1);            // conv2 -> skip slot of iconv3
    CONV(D_CONV3_0, b.CAT3, 2 * g256 + 1, g256, g256, b.A3, g512, 0, 512, H2, W2);
    CONV(D_CONV3_3, b.A3, g512, 0, g512, b.CAT4, 2 * g512, g512, 512, H2, W2);                // conv3 -> skip slot of iconv4
    CONV(D_CONV4_0, b.CAT4, 2 * g512, g512, g512, b.A4, g512, 0, 512, H3, W3);
    CONV(D_CONV4_3, b.A4, g512, 0, g512, b.CAT5, 2 * g512, g512, 512, H3, W3);                // conv4 -> skip slot of iconv5
    CONV(D_CONV5_0, b.CAT5, 2 * g512, g512, g512, b.A5, g512, 0, 512, H4, W4);
    CONV(D_CONV5_3, b.A5, g512, 0, g512, b.C5, g512, 0, 512, H4, W4);
    // decoder                                                                  :241-261
    CNM_TRY(E::upconv(b.C5, g512, b.U5, b.CAT5, 2 * g512, 0, 512, wt[D_UPCONV5], P, H5, W5, b.SYNC, s));
    CONV(D_ICONV5, b.CAT5, 2 * g512, 0, 2 * g512, b.I5, g512, 0, 512, H4, W4);
    CNM_TRY(E::upconv(b.I5, g512, b.U4, b.CAT4, 2 * g512, 0, 512, wt[D_UPCONV4], P, H4, W4, b.SYNC, s));
    CONV(D_ICONV4, b.CAT4, 2 * g512, 0, 2 * g512, b.I4, g512, 0, 512, H3, W3);
    CNM_TRY(E::head(b.I4, g512, 512, wt[D_DISP4], idepth_scale, disp4, b.CAT3, 2 * g256 + 1, 2 * g256, P, H3, W3, s));
    CNM_TRY(E::upconv(b.I4, g512, b.U3, b.CAT3, 2 * g256 + 1, 0, 256, wt[D_UPCONV3], P, H3, W3, b.SYNC, s));
    CONV(D_ICONV3, b.CAT3, 2 * g256 + 1, 0, 2 * g256 + 1, b.I3, g256, 0, 256, H2, W2);
    CNM_TRY(E::head(b.I3, g256, 256, wt[D_DISP3], idepth_scale, disp3, b.CAT2, 2 * g128 + 1, 2 * g128, P, H2, W2, s));
    CNM_TRY(E::upconv(b.I3, g256, b.U2, b.CAT2, 2 * g128 + 1, 0, 128, wt[D_UPCONV2], P, H2, W2, b.SYNC, s));
    CONV(D_ICONV2, b.CAT2, 2 * g128 + 1, 0, 2 * g128 + 1, b.I2, g128, 0, 128, H1, W1);
    CNM_TRY(E::head(b.I2, g128, 128, wt[D_DISP2], idepth_scale, disp2, b.CAT1, g64 + 1, g64, P, H1, W1, s));
    CNM_TRY(E::upconv(b.I2, g128, b.U1, b.CAT1, g64 + 1, 0, 64, wt[D_UPCONV1], P, H1, W1, b.SYNC, s));
    CONV(D_ICONV1, b.CAT1, g64 + 1, 0, g64 + 1, iconv1, g64, 0, 64, H, W);
    CNM_TRY(E::head(iconv1, g64, 64, wt[D_DISP1], idepth_scale, disp1, nullptr, 0, 0, P, H, W, s));
#undef CONV
    return CNM_OK;
}

extern "C" int cnm_depthnet_forward_f32(const cnm_layer_weights* wt, float idepth_scale, int D,
                                        const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                                        float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                                        float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream) {
    return depthnet_forward<EngF32>(wt, idepth_scale, D, ref, src, ref_cam, src_cam, disp1, disp2, disp3, disp4, iconv1_c4, ws, ws_floats, B, S, H, W, stream);
}

// [r6] The same two forwards with the images and cameras as VIEWS of the caller's frame tensors: *_bstride = floats between consecutive frames
// (0 = dense).  ref = frames[:, 0] / src = frames[:, 1:] of frames [B][1 + S][3][H][W] and the like for cams [B][1 + S][2][4][4] are read where
// they lie -- the reference slices them the same way (eval.py:440-447) and lets every consumer make its own contiguous copy.
extern "C" int cnm_depthnet_forward_strided_f32(const cnm_layer_weights* wt, float idepth_scale, int D,
                                                const float* ref, long long ref_bstride, const float* src, long long src_bstride,
                                                const float* ref_cam, long long ref_cam_bstride, const float* src_cam, long long src_cam_bstride,
                                                float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                                                float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream) {
    return depthnet_forward<EngF32>(wt, idepth_scale, D, ref, src, ref_cam, src_cam, disp1, disp2, disp3, disp4, iconv1_c4, ws, ws_floats, B, S, H, W, stream,
                                    InStrides{ref_bstride, src_bstride, ref_cam_bstride, src_cam_bstride});
}
extern "C" int cnm_depthnet_forward_strided_f16(const cnm_layer_weights* wt, float idepth_scale, int D,
                                                const float* ref, long long ref_bstride, const float* src, long long src_bstride,
                                                const float* ref_cam, long long ref_cam_bstride, const float* src_cam, long long src_cam_bstride,
                                                float* disp1, float* disp2, float* disp3, float* disp4, void* iconv1_c8,
                                                float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream) {
    return depthnet_forward<EngF16>(wt, idepth_scale, D, ref, src, ref_cam, src_cam, disp1, disp2, disp3, disp4, static_cast<float*>(iconv1_c8), ws, ws_floats, B, S, H, W, stream,
                                    InStrides{ref_bstride, src_bstride, ref_cam_bstride, src_cam_bstride});
}

extern "C" int cnm_depthnet_forward_f16(const cnm_layer_weights* wt, float idepth_scale, int D,
                                        const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                                        float* disp1, float* disp2, float* disp3, float* disp4, void* iconv1_c8,
                                        float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream) {
    return depthnet_forward<EngF16>(wt, idepth_scale, D, ref, src, ref_cam, src_cam, disp1, disp2, disp3, disp4, static_cast<float*>(iconv1_c8), ws, ws_floats, B, S, H, W, stream);
}

// ------------------------------------------------------------------ refine net
struct DecoderBufs { float *UC3, *I3, *U2, *UC2, *I2, *U1, *UC1, *I1; };     // one set per decoder: the two run concurrently
struct RefineBufs { float *SYNC[2], *X, *A1, *C1, *A2, *C2, *A3, *C3, *U3; DecoderBufs d[2]; };   // SYNC: one sync workspace per stream (encoder + depth decoder, probability decoder)



// The encoder's buffers are dead when the two decoders start (only the skip tensors C1, C2 and the shared U3 are read
// by them), so decoder 0's temporaries reuse them: U1 <- A1, UC1 <- X, U2 <- A2, UC3 + I3 <- A3.  Decoder 1 runs
// concurrently on the side stream and keeps buffers of its own.
template <class E>
static size_t carve_refine(float* ws, int N, int H, int W, RefineBufs* b) {
    Carver c{ws, 0};
    const size_t q = (size_t)N * H * W * 4;
    auto G = [](int C) { return (size_t)((C + E::GD - 1) / E::GD); };
    for (int k = 0; k < 2; ++k) b->SYNC[k] = (E::GD == 4 && !E::HOST) ? c.take(cnm_wino36_sync_floats()) : nullptr;   // FIRST, fixed offsets, never reused: flag words zero between calls
    b->X = c.take(q * (G(64) + 1)); b->A1 = c.take(q * G(128));
    b->C1 = c.take(q / 4 * G(128)); b->A2 = c.take(q / 4 * G(256));
    b->C2 = c.take(q / 16 * G(256)); b->A3 = c.take(q / 16 * G(512)); b->U3 = c.take(q / 16 * G(512));
    for (int k = 0; k < 2; ++k) {
        DecoderBufs& d = b->d[k];
        const bool share = k == 0;
        d.U1 = share ? b->A1 : c.take(q * G(128));
        d.UC1 = share ? b->X : c.take(q * G(64));
        d.I1 = c.take(q * G(64));
        d.U2 = share ? b->A2 : c.take(q / 4 * G(256));
        d.UC2 = c.take(q / 4 * G(128)); d.I2 = c.take(q / 4 * G(128));
        d.UC3 = share ? b->A3 : c.take(q / 16 * G(256));
        d.I3 = share ? (b->A3 ? b->A3 + ((q / 16 * G(256) + 63) & ~(size_t)63) : nullptr) : c.take(q / 16 * G(256));
    }
    b->C3 = c.take(q / 64 * G(512));
    return c.used;
}

extern "C" size_t cnm_refinenet_workspace_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
    RefineBufs b;
    return carve_refine<EngF32>(nullptr, N, H, W, &b);
}

template <class E>
static int refinenet_body(const cnm_layer_weights* wt, float idepth_scale, const RefineBufs& b,
                          float* disp_refined, float* prob_map, float* iconv1_depth, int N, int H, int W, void* s) {
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8;
    auto G = [](int C) { return (C + E::GD - 1) / E::GD; };
    const int g64 = G(64), g128 = G(128), g256 = G(256), g512 = G(512);
#define CONV(L, in, Gin, out, Cout, HH, WW) \
    CNM_TRY(E::conv(in, Gin, 0, Gin, out, G(Cout), 0, Cout, wt[L], N, HH, WW, 3, kRefineLayers[L].stride, b.SYNC[0], s))
    CONV(R_CONV1_0, b.X, g64 + 1, b.A1, 128, H, W);
    CONV(R_CONV1_3, b.A1, g128, b.C1, 128, H, W);
    CONV(R_CONV2_0, b.C1, g128, b.A2, 256, H1, W1);
    CONV(R_CONV2_3, b.A2, g256, b.C2, 256, H1, W1);
    CONV(R_CONV3_0, b.C2, g256, b.A3, 512, H2, W2);
    CONV(R_CONV3_3, b.A3, g512, b.C3, 512, H2, W2);
    CNM_TRY(E::up(b.C3, g512, b.U3, N, H3, W3, s));                                    // shared by both decoders
    SideStream* side = E::HOST ? nullptr : side_stream();
    void* st[2] = {s, side ? (void*)side->stream : s};
    if (side) {
        if (hipEventRecord(side->fork, (hipStream_t)s) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess) return CNM_ERR_LAUNCH;
    }
#undef CONV
    // Between the fork and the join nothing returns early: a failed launch ends the loop, and the side stream is
    // joined back in every case (kernels already queued on it must be ordered before the caller's stream goes on, and
    // a stream capture must not be left with an unjoined branch).
#define CONV(L, in, Gin, out, Cout, HH, WW) \
    rc = E::conv(in, Gin, 0, Gin, out, G(Cout), 0, Cout, wt[L], N, HH, WW, 3, kRefineLayers[L].stride, sy, q)
    int rc = CNM_OK;
    for (int step = 0; step < 7 && rc == CNM_OK; ++step) {                             // launches interleaved decoder by decoder
        for (int br = 0; br < 2 && rc == CNM_OK; ++br) {                               // 0: depth (:341-351), 1: prob (:357-365)
            const int L = R_BRANCH0 + 6 * br;
            const DecoderBufs& d = b.d[br];
            void* q = st[br];
            float* sy = b.SYNC[side ? br : 0];                                         // one sync workspace per stream
            float* feat = (br == 0 && iconv1_depth) ? iconv1_depth : d.I1;
            switch (step) {
                case 0: CONV(L + 0, b.U3, g512, d.UC3, 256, H2, W2); break;
                case 1: rc = E::conv2(d.UC3, g256, b.C2, g256, d.I3, g256, 256, wt[L + 1], N, H2, W2, sy, q); break;
                case 2: rc = E::upconv(d.I3, g256, d.U2, d.UC2, g128, 0, 128, wt[L + 2], N, H2, W2, sy, q); break;
                case 3: rc = E::conv2(d.UC2, g128, b.C1, g128, d.I2, g128, 128, wt[L + 3], N, H1, W1, sy, q); break;
                case 4: rc = E::upconv(d.I2, g128, d.U1, d.UC1, g64, 0, 64, wt[L + 4], N, H1, W1, sy, q); break;
                case 5: CONV(L + 5, d.UC1, g64, feat, 64, H, W); break;
                case 6: rc = E::head(feat, g64, 64, wt[R_HEAD0 + br], br == 0 ? idepth_scale : 1.0f, br == 0 ? disp_refined : prob_map,
                                     nullptr, 0, 0, N, H, W, q); break;
            }
        }
    }
    if (side) {
        if (hipEventRecord(side->join, side->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)s, side->join, 0) != hipSuccess)
            return rc != CNM_OK ? rc : CNM_ERR_LAUNCH;
    }
#undef CONV
    return rc;
}

template <class E>
static int refinenet_forward(const cnm_layer_weights* wt, float idepth_scale,
                             const float* idepth01, const float* idepth02, long long idepth_stride,
                             const float* iconv01, int G1_total, int g1, const float* iconv02, int G2_total, int g2,
                             float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                             float* ws, size_t ws_floats, int N, int H, int W, void* stream) {
    CNM_REQUIRE(wt && idepth01 && idepth02 && iconv01 && iconv02 && disp_refined && prob_map && ws && N > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, CNM_ERR_BAD_SHAPE);
    for (int i = 0; i < R_NUM; ++i) CNM_REQUIRE((wt[i].w || wt[i].u || wt[i].u4) && wt[i].b, CNM_ERR_BAD_ARG);
    RefineBufs b;
    CNM_REQUIRE(carve_refine<E>(ws, N, H, W, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    CNM_TRY(E::assemble(idepth01, idepth02, idepth_stride, iconv01, G1_total, g1, iconv02, G2_total, g2, b.X, N, H, W, stream));  // :332-333
    return refinenet_body<E>(wt, idepth_scale, b, disp_refined, prob_map, iconv1_depth_c4, N, H, W, stream);
}

extern "C" int cnm_refinenet_forward_f32(const cnm_layer_weights* wt, float idepth_scale,
                                         const float* idepth01, const float* idepth02, long long idepth_stride,
                                         const float* iconv01, int G1_total, int g1,
                                         const float* iconv02, int G2_total, int g2,
                                         float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                         float* ws, size_t ws_floats, int N, int H, int W, void* stream) {
    return refinenet_forward<EngF32>(wt, idepth_scale, idepth01, idepth02, idepth_stride, iconv01, G1_total, g1, iconv02, G2_total, g2,
                                     disp_refined, prob_map, iconv1_depth_c4, ws, ws_floats, N, H, W, stream);
}

template <class E>
static int refinenet_forward_multi(const cnm_layer_weights* wt, float idepth_scale,
                                   const float* idepth_pairs, const float* iconv_pairs, int S,
                                   float* disp_refined, float* prob_map, float* iconv1_depth,
                                   float* ws, size_t ws_floats, int B, int H, int W, void* stream) {
    CNM_REQUIRE(wt && idepth_pairs && iconv_pairs && disp_refined && prob_map && ws && B > 0 && S >= 2 && S % 2 == 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, CNM_ERR_BAD_SHAPE);
    for (int i = 0; i < R_NUM; ++i) CNM_REQUIRE((wt[i].w || wt[i].u || wt[i].u4) && wt[i].b, CNM_ERR_BAD_ARG);
    RefineBufs b;
    CNM_REQUIRE(carve_refine<E>(ws, B, H, W, &b) <= ws_floats, CNM_ERR_WORKSPACE);
    CNM_TRY(E::assemble_multi(idepth_pairs, iconv_pairs, b.X, B, S, H, W, stream));
    return refinenet_body<E>(wt, idepth_scale, b, disp_refined, prob_map, iconv1_depth, B, H, W, stream);
}

extern "C" int cnm_refinenet_forward_multi_f32(const cnm_layer_weights* wt, float idepth_scale,
                                               const float* idepth_pairs, const float* iconv_pairs_c4, int S,
                                               float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                               float* ws, size_t ws_floats, int B, int H, int W, void* stream) {
    return refinenet_forward_multi<EngF32>(wt, idepth_scale, idepth_pairs, iconv_pairs_c4, S, disp_refined, prob_map, iconv1_depth_c4, ws, ws_floats, B, H, W, stream);
}

extern "C" int cnm_refinenet_forward_multi_f16(const cnm_layer_weights* wt, float idepth_scale,
                                               const float* idepth_pairs, const void* iconv_pairs_c8, int S,
                                               float* disp_refined, float* prob_map, void* iconv1_depth_c8,
                                               float* ws, size_t ws_floats, int B, int H, int W, void* stream) {
    return refinenet_forward_multi<EngF16>(wt, idepth_scale, idepth_pairs, static_cast<const float*>(iconv_pairs_c8), S, disp_refined, prob_map,
                                           static_cast<float*>(iconv1_depth_c8), ws, ws_floats, B, H, W, stream);
}

// ------------------------------------------------------------------ host twins of the whole-network entry points (EngHost)
// Same arguments with HOST pointers and no stream; cnm_layer_weights.w / .b = cnm_pack_conv_bn_cpu (heads: cnm_pack_head_cpu + the
// bias vector) outputs, the other slots unused.  BASELINE configs[0]; depthNet_model.py:226-263, :331-370.
extern "C" size_t cnm_depthnet_workspace_floats_cpu(int P, int H, int W, int D) {
    if (P <= 0 || H <= 0 || W <= 0 || D < 4 || (H % 32) || (W % 32) || (D % 4)) return 0;
    DepthBufs b;
    return carve_depth<EngHost>(nullptr, P, H, W, D, &b);
}

extern "C" int cnm_depthnet_forward_cpu(const cnm_layer_weights* wt, float idepth_scale, int D,
                                        const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                                        float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                                        float* ws, size_t ws_floats, int B, int S, int H, int W) {
    return depthnet_forward<EngHost>(wt, idepth_scale, D, ref, src, ref_cam, src_cam, disp1, disp2, disp3, disp4, iconv1_c4, ws, ws_floats, B, S, H, W, nullptr);
}

extern "C" size_t cnm_refinenet_workspace_floats_cpu(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
    RefineBufs b;
    return carve_refine<EngHost>(nullptr, N, H, W, &b);
}

extern "C" int cnm_refinenet_forward_cpu(const cnm_layer_weights* wt, float idepth_scale,
                                         const float* idepth01, const float* idepth02, long long idepth_stride,
                                         const float* iconv01, int G1_total, int g1, const float* iconv02, int G2_total, int g2,
                                         float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                         float* ws, size_t ws_floats, int N, int H, int W) {
    return refinenet_forward<EngHost>(wt, idepth_scale, idepth01, idepth02, idepth_stride, iconv01, G1_total, g1, iconv02, G2_total, g2,
                                      disp_refined, prob_map, iconv1_depth_c4, ws, ws_floats, N, H, W, nullptr);
}

extern "C" int cnm_refinenet_forward_multi_cpu(const cnm_layer_weights* wt, float idepth_scale,
                                               const float* idepth_pairs, const float* iconv_pairs_c4, int S,
                                               float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                               float* ws, size_t ws_floats, int B, int H, int W) {
    return refinenet_forward_multi<EngHost>(wt, idepth_scale, idepth_pairs, iconv_pairs_c4, S, disp_refined, prob_map, iconv1_depth_c4, ws, ws_floats, B, H, W, nullptr);
}
