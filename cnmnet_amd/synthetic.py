"""Seeded synthetic frames, cameras and weights (numpy only, reproducible without torch).

The recipe is SURVEY.md section 8(d): images ~ N(0,1) (post-ImageNet-normalisation
statistics, cf. reference scannet/preprocess.py:16-26), pin-hole intrinsics with a
ScanNet/7-Scenes-like field of view, reference extrinsic = identity, source s =
small seeded rotation + baseline +-0.10 m * ceil(s/2).  The camera tensor layout
is the reference's [B,2,4,4]: [:,0] = 4x4 world->camera extrinsic, [:,1,:3,:3] = K
(reference scannet/preprocess.py:29-46, eval.py:136-146).
"""
import math
import zlib

import numpy as np


def intrinsics(H, W):
    K = np.eye(4, dtype=np.float64)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 1.125 * W, 1.5 * H, W / 2.0, H / 2.0
    return K


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def cameras(B, S, H, W, rng, max_rot_deg=3.0, baseline=0.10):
    """-> cams float32 [B, 1+S, 2, 4, 4]; index 0 = reference view."""
    cams = np.zeros((B, 1 + S, 2, 4, 4), np.float32)
    K = intrinsics(H, W)
    for b in range(B):
        for s in range(1 + S):
            E = np.eye(4)
            if s > 0:
                ang = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg, 3))
                E[:3, :3] = _rot(*ang)
                E[:3, 3] = [(-1.0 if s % 2 else 1.0) * baseline * math.ceil(s / 2),
                            rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02)]
            cams[b, s, 0], cams[b, s, 1] = E, K
    return cams


def smooth_images(shape, rng, passes=2):
    """N(0,1) noise, lightly low-passed (so the cost volume is not pure noise),
    re-normalised to unit variance."""
    x = rng.standard_normal(shape).astype(np.float32)
    for _ in range(passes):
        x = (x + np.roll(x, 1, -1) + np.roll(x, -1, -1) + np.roll(x, 1, -2) + np.roll(x, -1, -2)) / 5.0
    return (x / x.std()).astype(np.float32)


def frames(B, S, H, W, seed=1234, smooth=True):
    """-> images [B,1+S,3,H,W] float32, cams [B,1+S,2,4,4] float32."""
    rng = np.random.default_rng(seed)
    shape = (B, 1 + S, 3, H, W)
    img = smooth_images(shape, rng) if smooth else rng.standard_normal(shape).astype(np.float32)
    return img, cameras(B, S, H, W, rng)


def state_dict_like(shapes, seed=7, randomize_bn=False):
    """Deterministic weights for a {key: shape} mapping that follows the reference's
    state_dict naming (conv weight = 4-D, BN = weight/bias/running_mean/running_var/
    num_batches_tracked, head bias = 1-D next to a 4-D weight with Cout=1).
    Conv: Kaiming-normal fan_out (reference depthNet_model.py:165-182); BN gamma=1,
    beta=0, stats (0,1) unless randomize_bn, in which case every BN tensor is perturbed
    so that folding mistakes cannot cancel."""
    out = {}
    for key, shp in shapes.items():
        shp = tuple(shp)
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])   # independent of key order
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros((), np.int64)
        elif len(shp) == 4:
            fan_out = shp[0] * shp[2] * shp[3]
            w = rng.standard_normal(shp) * math.sqrt(2.0 / fan_out)
            if shp[0] == 1 and randomize_bn:
                w *= 0.05   # heads: keep the sigmoid out of saturation so tests stay sensitive
            out[key] = w.astype(np.float32)
        elif key.endswith("running_mean"):
            out[key] = (rng.normal(0, 0.1, shp) if randomize_bn else np.zeros(shp)).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = (rng.uniform(0.5, 1.5, shp) if randomize_bn else np.ones(shp)).astype(np.float32)
        elif key.endswith("weight"):          # BN gamma
            out[key] = (rng.uniform(0.8, 1.2, shp) if randomize_bn else np.ones(shp)).astype(np.float32)
        else:                                 # BN beta or head bias
            out[key] = (rng.normal(0, 0.05, shp) if randomize_bn else np.zeros(shp)).astype(np.float32)
    return out
