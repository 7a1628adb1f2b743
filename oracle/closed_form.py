"""Independent numpy restatement (explicit taps, closed-form solves) of the three
geometry operators on the CNMNet hot path.  TEST INFRASTRUCTURE ONLY.

Written from the formulas, not from torch ops, so that a mistake shared by
``ref_arrangement.py`` and the HIP kernels (e.g. a wrong grid_sample convention)
cannot hide.  All arithmetic in float64 unless ``dtype`` says otherwise; the
plane count D, image size and window size are free parameters, which makes this
the oracle for the D != 64 configurations the reference cannot run
(depthnet/depthNet_model.py:194,199,208 hard-code 64).
"""
import numpy as np

IDEPTH_RANGE = {2.0: (0.02, 2.0), 3.0: (0.1, 3.0)}   # depthNet_model.py:186-191


def homography_terms(left_cam, right_cam):
    """[B,2,4,4] x2 -> (Hm [B,3,3], KT [B,3]) in float64.  depth_util.py:33-52"""
    lc, rc = np.asarray(left_cam, np.float64), np.asarray(right_cam, np.float64)
    rel = rc[:, 0] @ np.linalg.inv(lc[:, 0])
    K_l, K_r = lc[:, 1, :3, :3], rc[:, 1, :3, :3]
    Hm = K_r @ rel[:, :3, :3] @ np.linalg.inv(K_l)
    KT = np.einsum("bij,bj->bi", K_r, rel[:, :3, 3])
    return Hm, KT


def bilinear_zero(img, ix, iy):
    """grid_sample(bilinear, zeros, align_corners=False) on UN-normalised
    coordinates: img [C,H,W], ix/iy [...] -> [C,...].  Each of the four corners
    contributes only if it is itself inside the image."""
    C, H, W = img.shape
    x0, y0 = np.floor(ix), np.floor(iy)
    fx, fy = ix - x0, iy - y0
    out = np.zeros((C,) + ix.shape, img.dtype)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            xs = np.clip(np.nan_to_num(xi), 0, W - 1).astype(np.int64)
            ys = np.clip(np.nan_to_num(yi), 0, H - 1).astype(np.int64)
            out += np.where(ok, wx * wy, 0.0) * img[:, ys, xs]
    return out


def plane_sweep_volume(left, right, left_cam, right_cam, idepth_scale=3.0, planes=64, window=None):
    """cost [B,D,H,W]; ix = u' - 0.5, iy = v' - 0.5 (SURVEY appendix A.2,
    depthNet_model.py:204-223).  window = (y0, y1, x0, x1) evaluates only that crop of the reference
    pixels (cost [B,D,y1-y0,x1-x0]) -- for spot checks at sizes where the whole volume takes minutes."""
    left, right = np.asarray(left, np.float64), np.asarray(right, np.float64)
    B, _, H, W = left.shape
    Hm, KT = homography_terms(left_cam, right_cam)
    lo, hi = IDEPTH_RANGE[float(idepth_scale)]
    y0, y1, x0, x1 = (0, H, 0, W) if window is None else window
    ys, xs = np.mgrid[y0:y1, x0:x1].astype(np.float64)
    left = left[:, :, y0:y1, x0:x1]
    vol = np.empty((B, planes, y1 - y0, x1 - x0))
    for b in range(B):
        a = [Hm[b, i, 0] * xs + Hm[b, i, 1] * ys + Hm[b, i, 2] for i in range(3)]
        for d in range(planes):
            z = 1.0 / (lo + d * (hi - lo) / (planes - 1.0))
            den = a[2] * z + KT[b, 2] + 1e-6
            u, v = (a[0] * z + KT[b, 0]) / den, (a[1] * z + KT[b, 1]) / den
            vol[b, d] = np.abs(bilinear_zero(right[b], u - 0.5, v - 0.5) - left[b]).sum(0)
    return vol


def depth_to_normal(depth, K_inv, k_size=9):
    """normal [B,3,H,W], points [B,3,H,W]  (depth_util.py:160-203).
    Also returns the fallback mask (det<1e-5 or NaN) so tests can exclude pixels
    sitting on that discontinuity."""
    depth, K_inv = np.asarray(depth, np.float64), np.asarray(K_inv, np.float64)
    B, H, W = depth.shape
    r = k_size // 2
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    pix = np.stack([xs, ys, np.ones_like(xs)], 0)
    pts = np.einsum("bij,jhw->bihw", K_inv, pix) * depth[:, None]
    valid = (depth > 0) & (depth < 10.0)
    pv = np.pad(pts * valid[:, None], ((0, 0), (0, 0), (r, r), (r, r)))
    S = np.zeros((B, 3, 3, H, W)); s = np.zeros((B, 3, H, W))
    for dy in range(k_size):
        for dx in range(k_size):
            q = pv[:, :, dy:dy + H, dx:dx + W]
            s += q
            S += q[:, :, None] * q[:, None, :]
    S = np.moveaxis(S, (1, 2), (-2, -1)); s = np.moveaxis(s, 1, -1)
    det = np.linalg.det(S)
    bad = np.isnan(det) | (det < 1e-5)
    S = np.where(bad[..., None, None], np.eye(3), S)
    g = np.linalg.solve(S, s[..., None])[..., 0]
    n = g / (np.linalg.norm(g, axis=-1, keepdims=True) + 1e-5)
    return np.moveaxis(n, -1, 1), pts, bad


def inverse_warp(feat, depth, pose, K, K_inv):
    """[B,C,H,W] (inverse_warp.py:46-118, padding 'zeros')."""
    feat, depth = np.asarray(feat, np.float64), np.asarray(depth, np.float64)
    B, C, H, W = feat.shape
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    pix = np.stack([xs, ys, np.ones_like(xs)], 0)
    out = np.empty_like(feat)
    for b in range(B):
        cam = np.einsum("ij,jhw->ihw", np.asarray(K_inv[b], np.float64), pix) * depth[b]
        P = np.asarray(K[b], np.float64) @ np.asarray(pose[b], np.float64)
        pc = np.einsum("ij,jhw->ihw", P[:, :3], cam) + P[:, 3][:, None, None]
        Z = np.maximum(pc[2], 1e-3)
        xn = 2 * (pc[0] / Z) / (W - 1) - 1
        yn = 2 * (pc[1] / Z) / (H - 1) - 1
        xn = np.where((xn > 1) | (xn < -1), 2.0, xn)
        yn = np.where((yn > 1) | (yn < -1), 2.0, yn)
        out[b] = bilinear_zero(feat[b], ((xn + 1) * W - 1) / 2, ((yn + 1) * H - 1) / 2)
    return out


def upsample2x_bilinear(x):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=False) on [...,H,W]
    (SURVEY appendix A.4).  Check vector: [0,4,8] -> [0,1,3,5,7,8]."""
    def up1(a, axis):
        n = a.shape[axis]
        dst = np.arange(2 * n)
        src = np.maximum((dst + 0.5) / 2 - 0.5, 0.0)
        i0 = np.floor(src).astype(int); i1 = np.minimum(i0 + 1, n - 1); lam = src - i0
        shp = [1] * a.ndim; shp[axis] = -1
        return np.take(a, i0, axis) * (1 - lam).reshape(shp) + np.take(a, i1, axis) * lam.reshape(shp)
    return up1(up1(np.asarray(x, np.float64), -2), -1)
