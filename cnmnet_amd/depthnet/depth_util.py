"""Geometry operators with the reference's names (reference depthnet/depth_util.py)."""
import torch
import torch.nn as nn

from .. import ops
from .losses import total, row_totals, mean_all


def get_pixel_coordinates(height, width, device=None):
    """Reference depth_util.py:13-21: homogeneous pixel grid [3, W*H] in u-major order (p = u*H + v), on the GPU.
    The engine itself never needs it (the grid is implicit in the thread index); it exists so that the reference's
    own call sequence get_pixel_coordinates -> process_camera_parameters -> getVolume (depthNet_model.py:226-233)
    runs unchanged."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    u = torch.arange(width, device=device, dtype=torch.float32).repeat_interleave(height)
    v = torch.arange(height, device=device, dtype=torch.float32).repeat(width)
    return torch.stack((u, v, torch.ones_like(u)), 0)


def process_camera_parameters(left_cam, right_cam, pixel_coordinates):
    """Reference depth_util.py:24-56: (KRKiUV [B,3,H*W], KT [B,3,1]).  Hm = K_r R K_l^-1 and KT = K_r T come from the
    engine's camera kernel (fp64 inside, 12 floats per pair); KRKiUV = Hm @ pixel_coordinates is materialised only
    because the reference's signature returns it."""
    hmkt = ops.homography_terms(left_cam, right_cam.unsqueeze(1))
    Hm, KT = hmkt[:, :9].reshape(-1, 3, 3), hmkt[:, 9:].reshape(-1, 3, 1).contiguous()
    KRKiUV = torch.matmul(Hm, pixel_coordinates)
    KRKiUV.hmkt = hmkt          # the 12 terms ride along (a derived tensor does not inherit them): getVolume uses them as they are
    return KRKiUV, KT


def homography_from_grid_product(KRKiUV, KT, height, width):
    """Recover the 12 camera terms from a plain KRKiUV tensor [B,3,W*H] (u-major): column p is Hm (u,v,1)^T, so
    Hm[:,2] = column (0,0) and the other two columns are finite differences across the whole image (far-apart
    columns keep the fp32 rounding of the product below 1e-7 relative per coefficient)."""
    c00 = KRKiUV[:, :, 0]
    cu = KRKiUV[:, :, (width - 1) * height] if width > 1 else c00
    cv = KRKiUV[:, :, height - 1] if height > 1 else c00
    h0 = (cu - c00) / max(width - 1, 1)
    h1 = (cv - c00) / max(height - 1, 1)
    Hm = torch.stack((h0, h1, c00), 2)                                   # columns of Hm
    return torch.cat((Hm.reshape(-1, 9), KT.reshape(-1, 3)), 1).contiguous()


class Depth2normal(nn.Module):
    """depth [B,H,W], intrinsic_inv [B,3,3] -> (normal [B,3,H,W], points [B,3,H,W]) (reference depth_util.py:140-203);
    with instance_segs [B,P,H,W] bool and planes_num [B] -> (normal, loss, points) (reference :205-238)."""

    def __init__(self, k_size=9):
        super().__init__()
        self.k_size = k_size

    def forward(self, depth, intrinsic_inv, instance_segs=None, planes_num=None):
        if planes_num is not None:
            # plane branch (reference :205-238): -> (normal, loss, points)
            if torch.is_grad_enabled() and depth.requires_grad:
                # differentiable, as the reference's is: normals through the HIP backward kernel, the per-instance means / cosine
                # loss as the reference's own torch expressions on the device (same values as cnm_plane_normals_f32)
                from ..autograd import Depth2NormalFn
                normal, points = Depth2NormalFn.apply(depth, intrinsic_inv, self.k_size, False)
                normal, loss = _plane_normals_autograd(normal, instance_segs, planes_num)
                return normal, loss, points
            normal, points = ops.depth2normal(depth.detach(), intrinsic_inv, self.k_size)
            normal, loss = ops.plane_normals(normal, instance_segs, planes_num)
            return normal, loss, points
        if torch.is_grad_enabled() and depth.requires_grad:
            from ..autograd import Depth2NormalFn
            return Depth2NormalFn.apply(depth, intrinsic_inv, self.k_size, False)
        return ops.depth2normal(depth, intrinsic_inv, self.k_size)


def _plane_normals_autograd(normal, instance_segs, planes_num):
    """Reference depth_util.py:205-238 with autograd: plane instances in order, each instance's pixels replaced by the
    instance's mean normal (:221-236), loss = sum over instances of mean(1 - cos(mean, inside ? n : 0)) (:228-233).
    normal [B,3,H,W] -> (regularised normal [B,3,H,W], loss)."""
    n = normal.permute(0, 2, 3, 1)
    B, H, W, _ = n.shape
    loss = normal.new_zeros(())
    rows = []
    for b in range(B):
        nb = n[b]
        for i in range(int(planes_num[b])):
            m = instance_segs[b, i].to(normal.device).bool().unsqueeze(-1)       # [H,W,1]
            mf = m.to(nb.dtype)
            mean = row_totals((nb * mf).permute(2, 0, 1)) / total(mf)           # :221-225
            reg = mean.expand(H, W, 3)
            orig = torch.where(m, nb, torch.zeros_like(nb))                      # :228
            loss = loss + mean_all(1 - torch.nn.functional.cosine_similarity(reg.reshape(-1, 3), orig.reshape(-1, 3), dim=1))   # :230-233
            nb = torch.where(m, reg, nb)                                         # :235-236
        rows.append(nb)
    return torch.stack(rows, 0).permute(0, 3, 1, 2), loss


def get_normal_by_planes(gt_normal, instance_segs, planes_num):
    """Reference depth_util.py:243-278: every plane instance's pixels take the instance's mean normal."""
    return ops.plane_normals(gt_normal, instance_segs, planes_num, with_loss=False)[0]
