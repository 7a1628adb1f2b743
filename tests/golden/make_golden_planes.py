"""Generate tests/golden/planes_24x32.npz by RUNNING the imported reference's Depth2normal plane branch
(depth_util.py:205-238) and get_normal_by_planes (:243-278) on seeded inputs.  Authoring container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import import_reference as ir   # noqa: E402
from cnmnet_amd import synthetic as syn     # noqa: E402


def main():
    ns = ir.load()
    import importlib
    du = importlib.import_module("depthnet.depth_util")
    rng = np.random.default_rng(77)
    B, P, H, W = 2, 20, 24, 32
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = np.stack([1.2 + 0.02 * xs + 0.01 * ys + 0.02 * rng.standard_normal((H, W)),
                      2.5 - 0.03 * ys + 0.05 * np.sin(xs / 3.0)]).astype(np.float32)
    depth[0, 3:5, 4:9] = 0.0                                             # holes
    K = np.asarray(syn.intrinsics(H, W))[:3, :3]
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)[None].repeat(B, 0)
    seg = np.zeros((B, P, H, W), bool)
    seg[0, 0, 2:12, 2:14] = True; seg[0, 1, 8:20, 10:30] = True           # overlapping instances: order matters
    seg[0, 2, 20:24, 0:6] = True
    seg[1, 0, 0:24, 0:16] = True; seg[1, 1, 5:9, 20:25] = True
    planes_num = np.array([3, 2])
    d2n = du.Depth2normal(9)
    with torch.no_grad():
        n_plain, pts = d2n(torch.from_numpy(depth), torch.from_numpy(Kinv))
        n_reg, loss, _ = d2n(torch.from_numpy(depth), torch.from_numpy(Kinv), torch.from_numpy(seg), planes_num)
        gt_ref = du.get_normal_by_planes(n_plain.clone(), torch.from_numpy(seg), planes_num)
    np.savez_compressed(os.path.join(HERE, "planes_24x32.npz"), depth=depth, K_inv=Kinv, seg=seg, planes_num=planes_num,
                        normal_plain=n_plain.numpy(), normal_reg=n_reg.numpy(), loss=np.float32(loss), normal_by_planes=gt_ref.numpy())
    print("planes golden: loss", float(loss), "changed px", int((n_reg != n_plain).any(1).sum()))


if __name__ == "__main__":
    main()
