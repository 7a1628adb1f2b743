"""Geometry operators with the reference's names (reference depthnet/depth_util.py)."""
import torch
import torch.nn as nn

from .. import ops


def get_pixel_coordinates(height, width):
    """Reference depth_util.py:13-21 builds a [3, W*H] host grid and uploads it on every
    forward.  The engine derives (u, v, 1) from the thread index, so nothing is needed here;
    kept (returning the image size) so call sites written against the reference still read."""
    return (height, width)


def process_camera_parameters(left_cam, right_cam, pixel_coordinates=None):
    """Reference depth_util.py:24-56 returns (KRKiUV [B,3,H*W], KT [B,3,1]).  The engine's
    equivalent is 12 floats per pair: returns (Hm [B,3,3], KT [B,3,1]) with
    KRKiUV = Hm @ (u,v,1) left implicit."""
    hmkt = ops.homography_terms(left_cam, right_cam.unsqueeze(1))
    return hmkt[:, :9].reshape(-1, 3, 3), hmkt[:, 9:].reshape(-1, 3, 1)


class Depth2normal(nn.Module):
    """depth [B,H,W], intrinsic_inv [B,3,3] -> (normal [B,3,H,W], points [B,3,H,W]) (reference depth_util.py:140-203);
    with instance_segs [B,P,H,W] bool and planes_num [B] -> (normal, loss, points) (reference :205-238)."""

    def __init__(self, k_size=9):
        super().__init__()
        self.k_size = k_size

    def forward(self, depth, intrinsic_inv, instance_segs=None, planes_num=None):
        if planes_num is not None:
            # plane branch (reference :205-238): -> (normal, loss, points); inference semantics (no autograd through it)
            normal, points = ops.depth2normal(depth.detach(), intrinsic_inv, self.k_size)
            normal, loss = ops.plane_normals(normal, instance_segs, planes_num)
            return normal, loss, points
        if torch.is_grad_enabled() and depth.requires_grad:
            from ..autograd import Depth2NormalFn
            return Depth2NormalFn.apply(depth, intrinsic_inv, self.k_size, False)
        return ops.depth2normal(depth, intrinsic_inv, self.k_size)


def get_normal_by_planes(gt_normal, instance_segs, planes_num):
    """Reference depth_util.py:243-278: every plane instance's pixels take the instance's mean normal."""
    return ops.plane_normals(gt_normal, instance_segs, planes_num, with_loss=False)[0]
