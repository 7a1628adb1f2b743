"""Depth-based inverse warp with the reference's interface (reference depthnet/inverse_warp.py)."""
import torch

from .. import ops


def check_sizes(input, input_name, expected):
    """Same assertion text as reference inverse_warp.py:18-24."""
    condition = [input.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            condition.append(input.size(i) == int(size))
    assert all(condition), "wrong size for {}, expected {}, got  {}".format(input_name, "x".join(expected), list(input.size()))


def pixel2cam(depth, intrinsics_inv):
    """[B,H,W], [B,3,3] -> camera-space points [B,3,H,W] (reference inverse_warp.py:27-43).
    Computed by the depth->normal kernel's point output (k=1 window)."""
    return ops.depth2normal(depth, intrinsics_inv, 1)[1]


def inverse_warp(feat, depth, pose, intrinsics, intrinsics_inv, padding_mode="zeros"):
    """feat [B,C,H,W] sampled at the reprojection of the target depth (reference inverse_warp.py:81-118)."""
    check_sizes(depth, "depth", "BHW")
    check_sizes(pose, "pose", "B34")
    check_sizes(intrinsics, "intrinsics", "B33")
    check_sizes(intrinsics_inv, "intrinsics", "B33")
    assert intrinsics_inv.size() == intrinsics.size()
    if torch.is_grad_enabled() and depth.requires_grad:
        if padding_mode != "zeros":
            raise NotImplementedError("the gradient w.r.t. depth is built for padding_mode='zeros' (the reference's default and only use) only")
        from ..autograd import InverseWarpFn
        return InverseWarpFn.apply(feat, depth, pose, intrinsics, intrinsics_inv)
    return ops.inverse_warp(feat, depth, pose, intrinsics, intrinsics_inv, padding_mode)
