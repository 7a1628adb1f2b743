"""Generate tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE (xxlong0/CNMNet,
/root/reference, unmodified) on seeded inputs.  Authoring container only:

    python tests/golden/make_golden.py

Fixtures are data only: inputs (or the seed that regenerates them, plus an input
checksum) and the reference's outputs.  The reference has no tests or golden
vectors of its own, so these files ARE the parity pin.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from oracle import import_reference as ir          # noqa: E402
from cnmnet_amd import synthetic as syn            # noqa: E402

ICONV_CHANNELS = [0, 1, 7, 13, 31, 32, 50, 63]     # stored subset of the 64-ch feature


def tn(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_weights(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    w = syn.state_dict_like(shapes, seed=seed, randomize_bn=True)
    module.load_state_dict({k: tn(np.asarray(v)) for k, v in w.items()})
    return module.eval()


def hard_cameras(cams):
    """pair 1: strong yaw + forward motion so part of the sweep leaves the frame and
    some far-plane pixels project behind the source camera (negative t2)."""
    c = cams.copy()
    a = np.deg2rad(75.0)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    c[1, 1, 0, :3, :3] = R
    c[1, 1, 0, :3, 3] = [0.25, 0.05, -0.6]
    return c


@torch.no_grad()
def main():
    ns = ir.load()
    torch.set_num_threads(8)

    # (1) plane-sweep volume, H != W on purpose (u-major pixel order, depth_util.py:15-18)
    img, cams = syn.frames(2, 1, 32, 64, seed=101)
    cams = hard_cameras(cams)
    net = ns.depthNet(3.0)
    L, R, lc, rc = tn(img[:, 0]), tn(img[:, 1]), tn(cams[:, 0]), tn(cams[:, 1])
    KR, KT = ns.process_camera_parameters(lc, rc, ns.get_pixel_coordinates(32, 64))
    vol = net.getVolume(L, R, KR, KT)
    t2 = (KR[:, 2] * 10.0 + KT[:, 2])
    np.savez_compressed(os.path.join(HERE, "planesweep_32x64.npz"), seed=101, left=img[:, 0], right=img[:, 1],
                        left_cam=cams[:, 0], right_cam=cams[:, 1], idepth_scale=3.0, volume=vol.numpy(),
                        frac_negative_t2=float((t2 <= 0).float().mean()))
    print("planesweep: volume", tuple(vol.shape), "neg-depth frac", float((t2 <= 0).float().mean()),
          "finite", bool(torch.isfinite(vol).all()))
    # idepth_scale = 2.0 branch (depthNet_model.py:186-188)
    net2 = ns.depthNet(2.0)
    vol2 = net2.getVolume(L[:1], R[:1], KR[:1], KT[:1])
    np.savez_compressed(os.path.join(HERE, "planesweep_scale2_32x64.npz"), seed=101, volume=vol2.numpy())

    # (2) depthNet.forward eval, 64x96, B=2
    img, cams = syn.frames(2, 2, 64, 96, seed=202)
    dn = load_weights(ns.depthNet(3.0), seed=11)
    outs, feat = dn(tn(img[:, 0]), tn(img[:, 1]), tn(cams[:, 0]), tn(cams[:, 1]))
    outs_b, feat_b = dn(tn(img[:, 0]), tn(img[:, 2]), tn(cams[:, 0]), tn(cams[:, 2]))
    np.savez_compressed(os.path.join(HERE, "depthnet_64x96.npz"), seed=202, weight_seed=11,
                        input_checksum=float(np.abs(img).sum()), cams=cams,
                        disp1=outs[0].numpy(), disp2=outs[1].numpy(), disp3=outs[2].numpy(), disp4=outs[3].numpy(),
                        iconv1_channels=np.array(ICONV_CHANNELS), iconv1=feat[:, ICONV_CHANNELS].numpy(),
                        iconv1_abs_sum=float(feat.abs().sum()),
                        disp1_b=outs_b[0].numpy(), iconv1_b=feat_b[:, ICONV_CHANNELS].numpy())
    print("depthnet: disp1 mean/std", float(outs[0].mean()), float(outs[0].std()))

    # (3) DepthRefineNet.forward on the two depthNet outputs above
    rn = load_weights(ns.DepthRefineNet(32, 3.0), seed=12)
    disp, prob, vfeat = rn(idepth01=outs[0], idepth02=outs_b[0], iconv01=feat, iconv02=feat_b, ReturnVolume=True)
    np.savez_compressed(os.path.join(HERE, "refine_64x96.npz"), weight_seed=12, disp_refined=disp.numpy(),
                        prob_map=prob.numpy(), iconv1_depth=vfeat[:, ICONV_CHANNELS].numpy())
    print("refine: disp mean/std", float(disp.mean()), float(disp.std()), "prob", float(prob.mean()))

    # (4) Depth2normal, k=9 and k=5, 48x64, with zero holes and a >10 m patch
    rng = np.random.default_rng(404)
    ys, xs = np.mgrid[0:48, 0:64].astype(np.float32)
    depth = np.stack([2.0 + 0.01 * xs + 0.02 * ys + 0.15 * np.sin(xs / 5.0),
                      1.2 + 0.5 * np.cos(ys / 7.0) + 0.004 * xs]).astype(np.float32)
    depth += rng.normal(0, 0.002, depth.shape).astype(np.float32)
    depth[0, 10:14, 20:30] = 0.0
    depth[1, 30:40, 5:12] = 12.0
    depth[1, 0:3, 50:64] = 0.0
    K = syn.intrinsics(48, 64)[:3, :3].astype(np.float32)
    Kinv = np.repeat(np.linalg.inv(K)[None].astype(np.float32), 2, 0)
    d2n = {}
    for k in (9, 5):
        n, p = ns.Depth2normal(k)(tn(depth), tn(Kinv))
        d2n["normal_k%d" % k], d2n["points_k%d" % k] = n.numpy(), p.numpy()
    np.savez_compressed(os.path.join(HERE, "depth2normal_48x64.npz"), depth=depth, K_inv=Kinv, **d2n)
    print("depth2normal: |n| mean", float(np.linalg.norm(d2n["normal_k9"], axis=1).mean()))

    # (5) inverse_warp, 32x64, C=3 and C=1
    img, cams = syn.frames(2, 1, 32, 64, seed=505)
    ys, xs = np.mgrid[0:32, 0:64].astype(np.float32)
    depth = np.stack([1.5 + 0.01 * xs, 2.5 - 0.02 * ys]).astype(np.float32)
    rel = cams[:, 1, 0].astype(np.float64) @ np.linalg.inv(cams[:, 0, 0].astype(np.float64))
    pose = rel[:, :3, :].astype(np.float32)
    K = cams[:, 0, 1, :3, :3].copy()
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    w3 = ns.inverse_warp(tn(img[:, 1]), tn(depth), tn(pose), tn(K), tn(Kinv))
    w1 = ns.inverse_warp(tn(img[:, 1, :1]), tn(depth), tn(pose), tn(K), tn(Kinv))
    np.savez_compressed(os.path.join(HERE, "inverse_warp_32x64.npz"), feat=img[:, 1], depth=depth, pose=pose, K=K,
                        K_inv=Kinv, warped_c3=w3.numpy(), warped_c1=w1.numpy())
    print("inverse_warp: nonzero frac", float((w3 != 0).float().mean()))

    # (6) upsample check vector
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear")
    x = tn(np.random.default_rng(606).standard_normal((1, 2, 5, 7)).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "upsample2x_5x7.npz"), x=x.numpy(), y=up(x).numpy())

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-32s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == "__main__":
    main()
