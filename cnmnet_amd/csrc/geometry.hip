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

// ------------------------------------------------------------------ K6 backward (training, train.py:204-263)
// n = g/(|g|+eps), g = S^-1 s, S = sum m p p^T, s = sum m p over the window (S := I when det < 1e-5 or NaN).
// With gbar = d n/d g applied to nbar and u = S^-1 gbar (u = gbar on the identity branch), the adjoint of a
// window member is  pbar_j += m_j [u_i - u_i (g_i.p_j) - g_i (u_i.p_j)].  Summed over the windows that contain j
// (the same k x k neighbourhood) this is  pbar_j = m_j (U_j - M_j p_j)  with box sums U = sum u_i and
// M = sum (u_i g_i^T + g_i u_i^T) -- i.e. the backward pass is two passes of the forward's structure:
//   pass A: per pixel u (3) and sym(u g^T + g u^T) (6) -> 9-channel map;  pass B: k x k box sums of that map
//   (fp64) and  zbar = pbar . ray  (+ the direct gradient of the point-map output).
__global__ __launch_bounds__(256) void depth2normal_bwd_prepare_kernel(const float* __restrict__ depth, const float* __restrict__ Kinv,
                                                                       const float* __restrict__ gnormal, float* __restrict__ maps,
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
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            float z = depth[(size_t)b * HW + (size_t)y * W + x];
            if (inv_in) z = 1.0f / z;
            const float fx = (float)x, fy = (float)y;
            if (z > 0.f && z < 10.0f)
                v = make_float4((k00 * fx + k01 * fy + k02) * z, (k10 * fx + k11 * fy + k12) * z, (k20 * fx + k21 * fy + k22) * z, 1.f);
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
    const double c00 = syy * szz - syz * syz, c01 = sxz * syz - sxy * szz, c02 = sxy * syz - sxz * syy;
    const double c11 = sxx * szz - sxz * sxz, c12 = sxy * sxz - sxx * syz, c22 = sxx * syy - sxy * sxy;
    const double det = sxx * c00 + sxy * c01 + sxz * c02;
    const bool ident = !(det >= 1e-5);
    const double id = ident ? 0.0 : 1.0 / det;
    double gx, gy, gz;
    if (ident) { gx = sx; gy = sy; gz = sz; }
    else { gx = (c00 * sx + c01 * sy + c02 * sz) * id; gy = (c01 * sx + c11 * sy + c12 * sz) * id; gz = (c02 * sx + c12 * sy + c22 * sz) * id; }
    const double rn = sqrt(gx * gx + gy * gy + gz * gz), den = rn + 1e-5;
    const size_t o = (size_t)b * 3 * HW + (size_t)y * W + x;
    const double nbx = gnormal[o], nby = gnormal[o + HW], nbz = gnormal[o + 2 * (size_t)HW];
    // gbar = nbar/den - g (g.nbar)/(r den^2)
    const double gdot = gx * nbx + gy * nby + gz * nbz;
    const double f = rn > 0 ? gdot / (rn * den * den) : 0.0;
    const double bx = nbx / den - gx * f, by = nby / den - gy * f, bz = nbz / den - gz * f;
    double ux, uy, uz;
    if (ident) { ux = bx; uy = by; uz = bz; }
    else { ux = (c00 * bx + c01 * by + c02 * bz) * id; uy = (c01 * bx + c11 * by + c12 * bz) * id; uz = (c02 * bx + c12 * by + c22 * bz) * id; }
    float* m = maps + (size_t)b * 9 * HW + (size_t)y * W + x;
    m[0] = (float)ux; m[HW] = (float)uy; m[2 * (size_t)HW] = (float)uz;
    const double sgn = ident ? 0.0 : 1.0;                      // identity branch: S is a constant, no gradient through it
    m[3 * (size_t)HW] = (float)(sgn * 2 * ux * gx); m[4 * (size_t)HW] = (float)(sgn * (ux * gy + uy * gx)); m[5 * (size_t)HW] = (float)(sgn * (ux * gz + uz * gx));
    m[6 * (size_t)HW] = (float)(sgn * 2 * uy * gy); m[7 * (size_t)HW] = (float)(sgn * (uy * gz + uz * gy)); m[8 * (size_t)HW] = (float)(sgn * 2 * uz * gz);
}

__global__ __launch_bounds__(256) void depth2normal_bwd_gather_kernel(const float* __restrict__ depth, const float* __restrict__ Kinv,
                                                                      const float* __restrict__ maps, const float* __restrict__ gpoints,
                                                                      float* __restrict__ gdepth, int B, int H, int W, int r, int inv_in) {
    __shared__ float tile[9][(D2N_TH + 2 * D2N_MAXR) * (D2N_TW + 2 * D2N_MAXR)];
    const int b = blockIdx.z, tx0 = blockIdx.x * D2N_TW, ty0 = blockIdx.y * D2N_TH;
    const int HW = H * W;
    const int tw = D2N_TW + 2 * r, th = D2N_TH + 2 * r;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ly = i / tw, lx = i - ly * tw;
        const int x = tx0 + lx - r, y = ty0 + ly - r;
        const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
        const float* m = maps + (size_t)b * 9 * HW + (size_t)y * W + x;
#pragma unroll
        for (int c = 0; c < 9; ++c) tile[c][i] = in ? m[(size_t)c * HW] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x % D2N_TW, ly = threadIdx.x / D2N_TW;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int k = 2 * r + 1;
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) {
            const int i = (ly + dy) * tw + lx + dx;
#pragma unroll
            for (int c = 0; c < 9; ++c) acc[c] += (double)tile[c][i];
        }
    const float* ki = Kinv + (size_t)b * 9;
    const float fx = (float)x, fy = (float)y;
    const double rx = ki[0] * fx + ki[1] * fy + ki[2], ry = ki[3] * fx + ki[4] * fy + ki[5], rz = ki[6] * fx + ki[7] * fy + ki[8];
    const size_t oi = (size_t)b * HW + (size_t)y * W + x;
    const float zin = depth[oi];
    const float z = inv_in ? 1.0f / zin : zin;
    double zb = 0.0;
    if (z > 0.f && z < 10.0f) {
        const double px = rx * z, py = ry * z, pz = rz * z;
        const double qx = acc[0] - (acc[3] * px + acc[4] * py + acc[5] * pz);
        const double qy = acc[1] - (acc[4] * px + acc[6] * py + acc[7] * pz);
        const double qz = acc[2] - (acc[5] * px + acc[7] * py + acc[8] * pz);
        zb = qx * rx + qy * ry + qz * rz;
    }
    if (gpoints) {
        const size_t o = (size_t)b * 3 * HW + (size_t)y * W + x;
        zb += gpoints[o] * rx + gpoints[o + HW] * ry + gpoints[o + 2 * (size_t)HW] * rz;      // points = z * ray, unmasked
    }
    gdepth[oi] = (float)(inv_in ? -zb * (double)z * (double)z : zb);
}

extern "C" int cnm_depth2normal_backward_f32(const float* depth, const float* K_inv, const float* grad_normal,
                                             const float* grad_points, float* grad_depth, float* ws,
                                             int B, int H, int W, int ksize, int input_is_idepth, void* stream) {
    CNM_REQUIRE(depth && K_inv && grad_normal && grad_depth && ws && B > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(ksize >= 1 && (ksize & 1) && ksize / 2 <= D2N_MAXR && B <= 65535, CNM_ERR_BAD_ARG);
    dim3 grid(cnm_ceil_div(W, D2N_TW), cnm_ceil_div(H, D2N_TH), B);
    depth2normal_bwd_prepare_kernel<<<grid, 256, 0, cnm_stream(stream)>>>(depth, K_inv, grad_normal, ws, B, H, W, ksize / 2, input_is_idepth);
    depth2normal_bwd_gather_kernel<<<grid, 256, 0, cnm_stream(stream)>>>(depth, K_inv, ws, grad_points, grad_depth, B, H, W, ksize / 2, input_is_idepth);
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
// PAD: grid_sample's padding_mode, which the reference hands through unchanged (inverse_warp.py:116): 0 'zeros' (with the
// reference's own out-of-view masking, :71-75), 1 'border' (coordinate clipped to the pixel centres), 2 'reflection'
// (reflected about the image edges -0.5 / size - 0.5, then clipped) -- align_corners=False in all three.
__device__ __forceinline__ float iw_reflect(float x, int size) {        // ATen reflect_coordinates(x, -1, 2 size - 1) on half-pixel units, then clip
    const float span = (float)size;                                      // (2 size - 1 - (-1)) / 2
    x = fabsf(x + 0.5f);                                                 // distance from the low edge -0.5
    const float extra = fmodf(x, span);
    const int flips = (int)floorf(x / span);
    const float r = (flips & 1) ? span - extra - 0.5f : extra - 0.5f;
    return fminf(fmaxf(r, 0.f), (float)(size - 1));
}
template <int PAD>
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
    if constexpr (PAD == 0) {
        if (xn > 1.f || xn < -1.f) xn = 2.f;                                     // :71-75
        if (yn > 1.f || yn < -1.f) yn = 2.f;
    }
    float ix = ((xn + 1.f) * W - 1.f) * 0.5f, iy = ((yn + 1.f) * H - 1.f) * 0.5f;         // grid_sample, align_corners=False
    if constexpr (PAD == 1) { ix = fminf(fmaxf(ix, 0.f), (float)(W - 1)); iy = fminf(fmaxf(iy, 0.f), (float)(H - 1)); }
    if constexpr (PAD == 2) { if (fabsf(ix) < 1e7f && fabsf(iy) < 1e7f) { ix = iw_reflect(ix, W); iy = iw_reflect(iy, H); } }
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

// K7 backward w.r.t. the target depth (the warped-depth loss of train.py:284-293 differentiates the sampling
// position): d out_c / d z = d out_c/d ix * d ix/d z + d out_c/d iy * d iy/d z with the bilinear-tap derivative
// of grid_sample (zeros padding: absent corners contribute 0) and the chain through X/Z, Y/Z of inverse_warp.py:57-70
// (zero where Z is clamped or the normalised coordinate was forced to 2).
__global__ __launch_bounds__(256) void inverse_warp_bwd_depth_kernel(const float* __restrict__ feat, const float* __restrict__ depth,
                                                                     const float* __restrict__ pose, const float* __restrict__ K,
                                                                     const float* __restrict__ Kinv, const float* __restrict__ gout,
                                                                     float* __restrict__ gdepth, int B, int C, int H, int W) {
    const int HW = H * W;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * HW) return;
    const int b = (int)(idx / HW), pix = (int)(idx - (long long)b * HW);
    const int y = pix / W, x = pix - y * W;
    const float* ki = Kinv + (size_t)b * 9; const float* kk = K + (size_t)b * 9; const float* ps = pose + (size_t)b * 12;
    float P[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            P[i * 4 + j] = kk[i * 3 + 0] * ps[0 * 4 + j] + kk[i * 3 + 1] * ps[1 * 4 + j] + kk[i * 3 + 2] * ps[2 * 4 + j];
    const float z = depth[idx], fx = (float)x, fy = (float)y;
    const float rx = ki[0] * fx + ki[1] * fy + ki[2], ry = ki[3] * fx + ki[4] * fy + ki[5], rz = ki[6] * fx + ki[7] * fy + ki[8];
    const float cx = rx * z, cy = ry * z, cz = rz * z;
    const float X = P[0] * cx + P[1] * cy + P[2] * cz + P[3];
    const float Y = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
    const float Zr = P[8] * cx + P[9] * cy + P[10] * cz + P[11];
    const float Z = fmaxf(Zr, 1e-3f);
    const float dX = P[0] * rx + P[1] * ry + P[2] * rz, dY = P[4] * rx + P[5] * ry + P[6] * rz;
    const float dZ = Zr > 1e-3f ? P[8] * rx + P[9] * ry + P[10] * rz : 0.f;
    const float xn = 2.f * (X / Z) / (float)(W - 1) - 1.f, yn = 2.f * (Y / Z) / (float)(H - 1) - 1.f;
    const bool xin = !(xn > 1.f || xn < -1.f), yin = !(yn > 1.f || yn < -1.f);
    float g = 0.f;
    if (xin && yin) {
        const float ix = ((xn + 1.f) * W - 1.f) * 0.5f, iy = ((yn + 1.f) * H - 1.f) * 0.5f;
        const float dix = (float)W / (float)(W - 1) * (dX * Z - X * dZ) / (Z * Z);      // d ix / d z
        const float diy = (float)H / (float)(H - 1) * (dY * Z - Y * dZ) / (Z * Z);
        const float flx = floorf(ix), fly = floorf(iy);
        const float wx1 = ix - flx, wy1 = iy - fly, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const int xi = (int)flx, yi = (int)fly;
        const bool x0in = (unsigned)xi < (unsigned)W, x1in = (unsigned)(xi + 1) < (unsigned)W;
        const bool y0in = (unsigned)yi < (unsigned)H, y1in = (unsigned)(yi + 1) < (unsigned)H;
        for (int c = 0; c < C; ++c) {
            const float* s = feat + ((size_t)b * C + c) * HW + (ptrdiff_t)yi * W + xi;
            const float p00 = (y0in && x0in) ? s[0] : 0.f, p01 = (y0in && x1in) ? s[1] : 0.f;
            const float p10 = (y1in && x0in) ? s[W] : 0.f, p11 = (y1in && x1in) ? s[W + 1] : 0.f;
            const float dodx = (p01 - p00) * wy0 + (p11 - p10) * wy1;
            const float dody = (p10 - p00) * wx0 + (p11 - p01) * wx1;
            g += gout[((size_t)b * C + c) * HW + pix] * (dodx * dix + dody * diy);
        }
    }
    gdepth[idx] = g;
}

extern "C" int cnm_inverse_warp_backward_depth_f32(const float* feat, const float* depth, const float* pose,
                                                   const float* K, const float* K_inv, const float* grad_out,
                                                   float* grad_depth, int B, int C, int H, int W, void* stream) {
    CNM_REQUIRE(feat && depth && pose && K && K_inv && grad_out && grad_depth && B > 0 && C > 0 && H > 1 && W > 1, CNM_ERR_BAD_ARG);
    const long long total = (long long)B * H * W;
    inverse_warp_bwd_depth_kernel<<<(unsigned)cnm_ceil_div_ll(total, 256), 256, 0, cnm_stream(stream)>>>(feat, depth, pose, K, K_inv, grad_out, grad_depth, B, C, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_inverse_warp_pad_f32(const float* feat, const float* depth, const float* pose,
                                        const float* K, const float* K_inv, float* out,
                                        int B, int C, int H, int W, int padding_mode, void* stream) {
    CNM_REQUIRE(feat && depth && pose && K && K_inv && out && B > 0 && C > 0 && H > 1 && W > 1, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(padding_mode >= 0 && padding_mode <= 2, CNM_ERR_BAD_ARG);
    const long long total = (long long)B * H * W;
    const unsigned nb = (unsigned)cnm_ceil_div_ll(total, 256);
    if (padding_mode == 0) inverse_warp_kernel<0><<<nb, 256, 0, cnm_stream(stream)>>>(feat, depth, pose, K, K_inv, out, B, C, H, W);
    else if (padding_mode == 1) inverse_warp_kernel<1><<<nb, 256, 0, cnm_stream(stream)>>>(feat, depth, pose, K, K_inv, out, B, C, H, W);
    else inverse_warp_kernel<2><<<nb, 256, 0, cnm_stream(stream)>>>(feat, depth, pose, K, K_inv, out, B, C, H, W);
    CNM_LAUNCH_CHECK();
    return CNM_OK;
}

extern "C" int cnm_inverse_warp_f32(const float* feat, const float* depth, const float* pose,
                                    const float* K, const float* K_inv, float* out,
                                    int B, int C, int H, int W, void* stream) {
    return cnm_inverse_warp_pad_f32(feat, depth, pose, K, K_inv, out, B, C, H, W, 0, stream);
}

// ------------------------------------------------------------------ plane-instance regularisation of a normal map
// Reference depth_util.py:205-238 (Depth2normal with planes_num) and :243-278 (get_normal_by_planes): for every image b
// and plane instance i < planes_num[b], IN ORDER (instances may overlap; later ones see earlier overwrites):
//   mean = sum_pix(n * seg_i) / sum_pix(seg_i);   loss += mean_pix(1 - cos(mean, seg_i ? n : 0));   n = seg_i ? mean : n
// One launch per instance index, one workgroup per image: pass 1 reduces the masked sum (fp64), pass 2 applies the
// mean and reduces the loss term into loss_terms[b * P + i] (summed by the caller in a fixed order: deterministic).
__global__ __launch_bounds__(1024) void plane_normals_kernel(float* __restrict__ normal, const unsigned char* __restrict__ seg,
                                                             const int* __restrict__ planes_num, float* __restrict__ loss_terms,
                                                             int P, int HW, int inst) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (inst >= planes_num[b]) return;                                   // uniform per workgroup
    float* n = normal + (size_t)b * 3 * HW;
    const unsigned char* m = seg + ((size_t)b * P + inst) * HW;
    __shared__ double red[4][16];
    double sx = 0, sy = 0, sz = 0, cnt = 0;
    for (int p = t; p < HW; p += 1024)
        if (m[p]) { sx += n[p]; sy += n[p + HW]; sz += n[p + 2 * (size_t)HW]; cnt += 1; }
    auto block_sum4 = [&](double& a, double& b2, double& c, double& d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b2 += __shfl_down(b2, o); c += __shfl_down(c, o); d += __shfl_down(d, o); }
        if ((t & 63) == 0) { red[0][t >> 6] = a; red[1][t >> 6] = b2; red[2][t >> 6] = c; red[3][t >> 6] = d; }
        __syncthreads();
        a = b2 = c = d = 0;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; b2 += red[1][w]; c += red[2][w]; d += red[3][w]; }   // same order in every thread
        __syncthreads();
    };
    block_sum4(sx, sy, sz, cnt);
    const float mx = (float)(sx / cnt), my = (float)(sy / cnt), mz = (float)(sz / cnt);   // 0/0 = NaN for an empty instance, as the reference
    const float mnorm = fmaxf(sqrtf(mx * mx + my * my + mz * mz), 1e-8f);                // F.cosine_similarity: each norm clamped at eps = 1e-8
    double ls = 0, z0 = 0, z1 = 0, z2 = 0;
    for (int p = t; p < HW; p += 1024) {
        float sim = 0.f * (mx + my + mz);                                // outside the instance the second operand is the zero vector (NaN mean of an empty instance propagates, as in the reference)
        if (m[p]) {
            const float x = n[p], y = n[p + HW], z = n[p + 2 * (size_t)HW];
            sim = (mx * x + my * y + mz * z) / (mnorm * fmaxf(sqrtf(x * x + y * y + z * z), 1e-8f));
            n[p] = mx; n[p + HW] = my; n[p + 2 * (size_t)HW] = mz;
        }
        ls += 1.0 - (double)sim;
    }
    block_sum4(ls, z0, z1, z2);
    if (t == 0 && loss_terms) loss_terms[b * P + inst] = (float)(ls / HW);
}

extern "C" int cnm_plane_normals_f32(float* normal, const unsigned char* instance_segs, const int* planes_num, int max_planes_num,
                                     float* loss_terms, int B, int P, int H, int W, void* stream) {
    CNM_REQUIRE(normal && instance_segs && planes_num && B > 0 && B <= 65535 && P > 0 && H > 0 && W > 0, CNM_ERR_BAD_ARG);
    CNM_REQUIRE(max_planes_num >= 0 && max_planes_num <= P, CNM_ERR_BAD_ARG);
    for (int i = 0; i < max_planes_num; ++i) {
        plane_normals_kernel<<<B, 1024, 0, cnm_stream(stream)>>>(normal, instance_segs, planes_num, loss_terms, P, H * W, i);
        CNM_LAUNCH_CHECK();
    }
    return CNM_OK;
}
