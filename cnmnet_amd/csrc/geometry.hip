// K6 depth->normal and K7 depth-based inverse warp.
//
// K6 replaces Depth2normal.forward without the plane-instance branch
//    (reference depthnet/depth_util.py:149-203).  The reference unfolds [B,H,W,k*k,3] patch
//    tensors several times (48 MB each per 192x256 sample) and runs batched 3x3 det/inverse;
//    here one workgroup stages a (32+2r)x(8+2r) tile of masked camera-space points in LDS and
//    every lane accumulates the 6+3 normal-equation sums of its own window, then solves the
//    3x3 system in closed form.  The sums and the solve are carried in fp64: the normal
//    equations (sum p p^T is not centred) are ill-conditioned in fp32 -- the reference's own
//    fp32 result moves by ~1e-4..1e-3 with summation order -- so fp64 puts this kernel at the
//    exact answer of the fp32 point cloud, i.e. as close to the reference as the reference is
//    to exact.  HBM traffic per pixel: read 4 B (depth), write 24 B (normal + points).
// K7 replaces inverse_warp / pixel2cam / cam2pixel (reference depthnet/inverse_warp.py:27-118),
//    padding_mode = 'zeros', including its mixed normalisation (x_norm uses W-1, the sampler
//    un-normalises with align_corners=False; inverse_warp.py:69-70,116).
#include "cnm_common.h"

#define D2N_TW 32
#define D2N_TH 8
#define D2N_MAXR 7

__global__ __launch_bounds__(256) void depth2normal_kernel(const float* __restrict__ depth, const float* __restrict__ Kinv,
                                                           float* __restrict__ normal, float* __restrict__ points,
                                                           int B, int H, int W, int r, int inv_in) {
    __shared__ float4 pts[(D2N_TH + 2 * D2N_MAXR) * (D2N_TW + 2 * D2N_MAXR)];
    const int b = blockIdx.z, tx0 = blockIdx.x * D2N_TW, ty0 = blockIdx.y * D2N_TH;
    const int HW = H * W;
    const float* ki = Kinv + (size_t)b * 9;
    const float k00 = ki[0], k01 = ki[1], k02 = ki[2], k10 = ki[3], k11 = ki[4], k12 = ki[5], k20 = ki[6], k21 = ki[7], k22 = ki[8];
    const int tw = D2N_TW + 2 * r, th = D2N_TH + 2 * r;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ly = i / tw, lx = i - ly * tw;
        const int x = tx0 + lx - r, y = ty0 + ly - r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                 // zero padding: zero AND invalid (depth_util.py:165)
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            float z = depth[(size_t)b * HW + (size_t)y * W + x];
            if (inv_in) z = 1.0f / z;                                   // eval.py:452
            const float fx = (float)x, fy = (float)y;
            // pixel2cam: K^-1 (x,y,1) * depth                     (inverse_warp.py:40-43)
            const float px = (k00 * fx + k01 * fy + k02) * z, py = (k10 * fx + k11 * fy + k12) * z, pz = (k20 * fx + k21 * fy + k22) * z;
            const bool inside_tile = lx >= r && lx < r + D2N_TW && ly >= r && ly < r + D2N_TH;
            if (inside_tile) {
                const size_t o = (size_t)b * 3 * HW + (size_t)y * W + x;
                points[o] = px; points[o + HW] = py; points[o + 2 * (size_t)HW] = pz;   // un-masked point map
            }
            if (z > 0.f && z < 10.0f) v = make_float4(px, py, pz, 1.f);                   // depth_util.py:162
        }
        pts[i] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x % D2N_TW, ly = threadIdx.x / D2N_TW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0, sx = 0, sy = 0, sz = 0;
    const int k = 2 * r + 1;
    for (int dy = 0; dy < k; ++dy) {
        const float4* row = pts + (ly + dy) * tw + lx;
        for (int dx = 0; dx < k; ++dx) {
            const float4 q = row[dx];
            const double px = q.x, py = q.y, pz = q.z;
            sxx = fma(px, px, sxx); sxy = fma(px, py, sxy); sxz = fma(px, pz, sxz);
            syy = fma(py, py, syy); syz = fma(py, pz, syz); szz = fma(pz, pz, szz);
            sx += px; sy += py; sz += pz;
        }
    }
    // closed-form symmetric 3x3 solve; identity fallback when det < 1e-5 or NaN (depth_util.py:185-198)
    const double c00 = syy * szz - syz * syz, c01 = sxz * syz - sxy * szz, c02 = sxy * syz - sxz * syy;
    const double c11 = sxx * szz - sxz * sxz, c12 = sxy * sxz - sxx * syz, c22 = sxx * syy - sxy * sxy;
    const double det = sxx * c00 + sxy * c01 + sxz * c02;
    double gx, gy, gz;
    if (!(det >= 1e-5)) { gx = sx; gy = sy; gz = sz; }
    else {
        const double id = 1.0 / det;
        gx = (c00 * sx + c01 * sy + c02 * sz) * id;
        gy = (c01 * sx + c11 * sy + c12 * sz) * id;
        gz = (c02 * sx + c12 * sy + c22 * sz) * id;
    }
    const double inv = 1.0 / (sqrt(gx * gx + gy * gy + gz * gz) + 1e-5);     // depth_util.py:201
    const size_t o = (size_t)b * 3 * HW + (size_t)y * W + x;
    normal[o] = (float)(gx * inv); normal[o + HW] = (float)(gy * inv); normal[o + 2 * (size_t)HW] = (float)(gz * inv);
}

extern "C" int cnm_depth2normal_f32(const float* depth, const float* K_inv, float* normal, float* points,
                                    int B, int H, int W, int ksize, int input_is_idepth, void* stream) {
    CNM_REQUIRE(depth && K_inv && normal && points && B > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ksize >= 1 && (ksize & 1) && ksize / 2 <= D2N_MAXR && B <= 65535, CNM_ERR_BAD_ARG);
    dim3 grid(cnm_ceil_div(W, D2N_TW), cnm_ceil_div(H, D2N_TH), B);
    depth2normal_kernel<<<grid, 256, 0, cnm_stream(stream)>>>(depth, K_inv, normal, points, B, H, W, ksize / 2, input_is_idepth);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

__global__ void intrinsics_inverse_kernel(const float* __restrict__ cam, long long stride, float* __restrict__ Kinv, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* k = cam + (size_t)b * stride + 16;
    const double a = k[0], bb = k[1], c = k[2], d = k[4], e = k[5], f = k[6], g = k[8], h = k[9], i = k[10];
    const double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + bb * Bc + c * C, id = 1.0 / det;
    float* o = Kinv + (size_t)b * 9;
    o[0] = (float)(A * id); o[1] = (float)(-(bb * i - c * h) * id); o[2] = (float)((bb * f - c * e) * id);
    o[3] = (float)(Bc * id); o[4] = (float)((a * i - c * g) * id); o[5] = (float)(-(a * f - c * d) * id);
    o[6] = (float)(C * id); o[7] = (float)(-(a * h - bb * g) * id); o[8] = (float)((a * e - bb * d) * id);
}

extern "C" int cnm_intrinsics_inverse_f32(const float* cam, long long cam_stride, float* K_inv, int B, void* stream) {
    CNM_REQUIRE(cam && K_inv && B > 0 && cam_stride >= 32, CNM_ERR_BAD_ARG);
    intrinsics_inverse_kernel<<<cnm_ceil_div(B, 64), 64, 0, cnm_stream(stream)>>>(cam, cam_stride, K_inv, B);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

// ------------------------------------------------------------------ K7
__global__ __launch_bounds__(256) void inverse_warp_kernel(const float* __restrict__ feat, const float* __restrict__ depth,
                                                           const float* __restrict__ pose, const float* __restrict__ K,
                                                           const float* __restrict__ Kinv, float* __restrict__ out,
                                                           int B, int C, int H, int W) {
    const int HW = H * W;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * HW) return;
    const int b = (int)(idx / HW), pix = (int)(idx - (long long)b * HW);
    const int y = pix / W, x = pix - y * W;
    const float* ki = Kinv + (size_t)b * 9; const float* kk = K + (size_t)b * 9; const float* ps = pose + (size_t)b * 12;
    float P[12];                                                   // proj = K @ pose [3,4]  (inverse_warp.py:110)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            P[i * 4 + j] = kk[i * 3 + 0] * ps[0 * 4 + j] + kk[i * 3 + 1] * ps[1 * 4 + j] + kk[i * 3 + 2] * ps[2 * 4 + j];
    const float z = depth[idx], fx = (float)x, fy = (float)y;
    const float cx = (ki[0] * fx + ki[1] * fy + ki[2]) * z, cy = (ki[3] * fx + ki[4] * fy + ki[5]) * z, cz = (ki[6] * fx + ki[7] * fy + ki[8]) * z;
    const float X = P[0] * cx + P[1] * cy + P[2] * cz + P[3];
    const float Y = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
    const float Z = fmaxf(P[8] * cx + P[9] * cy + P[10] * cz + P[11], 1e-3f);   // :67
    float xn = 2.f * (X / Z) / (float)(W - 1) - 1.f;                            // :69
    float yn = 2.f * (Y / Z) / (float)(H - 1) - 1.f;                            // :70
    if (xn > 1.f || xn < -1.f) xn = 2.f;                                         // :71-75
    if (yn > 1.f || yn < -1.f) yn = 2.f;
    const float ix = ((xn + 1.f) * W - 1.f) * 0.5f, iy = ((yn + 1.f) * H - 1.f) * 0.5f;   // grid_sample, align_corners=False
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f; int xi = 0, yi = 0;
    bool x0in = false, x1in = false, y0in = false, y1in = false;
    if (fabsf(ix) < 1e7f && fabsf(iy) < 1e7f) {
        const float flx = floorf(ix), fly = floorf(iy);
        const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        xi = (int)flx; yi = (int)fly;
        x0in = (unsigned)xi < (unsigned)W; x1in = (unsigned)(xi + 1) < (unsigned)W;
        y0in = (unsigned)yi < (unsigned)H; y1in = (unsigned)(yi + 1) < (unsigned)H;
        w00 = wx0 * wy0; w01 = wx1 * wy0; w10 = wx0 * wy1; w11 = wx1 * wy1;
    }
    for (int c = 0; c < C; ++c) {
        const float* s = feat + ((size_t)b * C + c) * HW + (ptrdiff_t)yi * W + xi;
        float v = 0.f;
        if (y0in && x0in) v = w00 * s[0];
        if (y0in && x1in) v = fmaf(w01, s[1], v);
        if (y1in && x0in) v = fmaf(w10, s[W], v);
        if (y1in && x1in) v = fmaf(w11, s[W + 1], v);
        out[((size_t)b * C + c) * HW + pix] = v;
    }
}

extern "C" int cnm_inverse_warp_f32(const float* feat, const float* depth, const float* pose,
                                    const float* K, const float* K_inv, float* out,
                                    int B, int C, int H, int W, void* stream) {
    CNM_REQUIRE(feat && depth && pose && K && K_inv && out && B > 0 && C > 0 && H > 1 && W > 1, CNM_ERR_BAD_ARG);
    const long long total = (long long)B * H * W;
    inverse_warp_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(feat, depth, pose, K, K_inv, out, B, C, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}
