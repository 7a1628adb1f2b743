"""Reference-arrangement CPU graph: stock torch ops wired as CNMNet wires them.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never imported by cnmnet_amd.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Parity of this file against the imported reference is
checked by tests/test_oracle_vs_reference.py (authoring container only) and
frozen in tests/golden/*.npz.  Differences from the reference, all deliberate:
  * device-agnostic (the reference hard-codes torch.cuda.FloatTensor),
  * depth-plane count D is a parameter (the reference hard-codes 64:
    depthnet/depthNet_model.py:194,199,208 and conv1 in-channels 67 at :134),
  * align_corners is spelled out as False everywhere, which is what
    torch>=1.3 computes for the reference's default-argument calls.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

IDEPTH_RANGE = {2.0: (0.02, 2.0), 3.0: (0.1, 3.0)}   # depthNet_model.py:186-191


# --------------------------------------------------------------------------
# camera prep                                            depth_util.py:13-56
# --------------------------------------------------------------------------
def pixel_grid_umajor(height, width):
    """(u,v,1) for every pixel, flat index p = u*H + v.   depth_util.py:13-21"""
    uv = np.indices([width, height]).astype(np.float32)
    uv1 = np.concatenate([uv, np.ones([1, width, height], np.float32)], 0)
    return torch.from_numpy(uv1.reshape(3, -1))


def homography_terms(left_cam, right_cam):
    """Hm = K_r R K_l^-1  [B,3,3] and KT = K_r T  [B,3,1].   depth_util.py:33-52"""
    E_l, K_l = left_cam[:, 0], left_cam[:, 1, :3, :3]
    E_r, K_r = right_cam[:, 0], right_cam[:, 1, :3, :3]
    rel = torch.matmul(E_r, E_l.inverse())
    R, T = rel[:, :3, :3], rel[:, :3, 3]
    Hm = torch.matmul(K_r, torch.matmul(R, K_l.inverse()))
    KT = torch.stack([torch.matmul(K_r[i], T[i]) for i in range(K_r.shape[0])], 0)
    return Hm, KT.unsqueeze(-1)


# --------------------------------------------------------------------------
# plane-sweep cost volume                          depthNet_model.py:185-224
# --------------------------------------------------------------------------
def plane_depths(idepth_scale, planes):
    lo, hi = IDEPTH_RANGE[float(idepth_scale)]
    step = (hi - lo) / (planes - 1.0)                    # :194 (63.0 for D=64)
    return [1.0 / (lo + d * step) for d in range(planes)]   # :209, python doubles


def plane_sweep_volume(left, right, left_cam, right_cam, idepth_scale=3.0, planes=64):
    """cost[b,d,y,x] = sum_c |grid_sample(right)[b,c,y,x] - left[b,c,y,x]|."""
    B, _, H, W = left.shape
    Hm, KT = homography_terms(left_cam, right_cam)
    KRKiUV = torch.matmul(Hm, pixel_grid_umajor(H, W).to(left))          # depth_util.py:43
    half = torch.tensor([W / 2.0, H / 2.0], dtype=left.dtype).view(1, 2, 1)  # :204-206
    vol = left.new_empty(B, planes, H, W)
    for d, z in enumerate(plane_depths(idepth_scale, planes)):
        t = KRKiUV * z + KT                                               # :210
        uv = t[:, 0:2, :] / (t[:, 2:3, :] + 1e-6)                         # :211-212
        uv = (uv - half) / half                                           # :213
        grid = uv.view(B, 2, W, H).permute(0, 3, 2, 1)                    # :214-219
        warped = F.grid_sample(right, grid, mode="bilinear", padding_mode="zeros",
                               align_corners=False)                       # :220
        vol[:, d] = (warped - left).abs().sum(1)                          # :222-223
    return vol


# --------------------------------------------------------------------------
# layer builders                                     depthNet_model.py:19-112
# --------------------------------------------------------------------------
def _cbr(cin, cout, k, stride):
    return [nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU()]


def _down(cin, cout, k):      # :19-39  conv s1 -> BN -> ReLU -> conv s2 -> BN -> ReLU
    return nn.Sequential(*(_cbr(cin, cout, k, 1) + _cbr(cout, cout, k, 2)))


def _same(cin, cout, k):      # :60-69
    return nn.Sequential(*_cbr(cin, cout, k, 1))


def _up(cin, cout, k):        # :91-101  bilinear x2 -> conv -> BN -> ReLU
    return nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False),
                         *_cbr(cin, cout, k, 1))


def _head(cin):               # :82-84
    return nn.Sequential(nn.Conv2d(cin, 1, 3, padding=1), nn.Sigmoid())


def _init(module):            # :165-182 (kaiming fan_out, BN 1/0, bias 0)
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class DepthNetCPU(nn.Module):
    """depthNet_model.py:124-263 with D planes (state_dict keys identical at D=64)."""

    def __init__(self, idepth_scale=3, planes=64):
        super().__init__()
        self.idepth_scale, self.planes = idepth_scale, planes
        self.conv1 = _down(3 + planes, 128, 7)
        self.conv2 = _down(128, 256, 5)
        self.conv3 = _down(256, 512, 3)
        self.conv4 = _down(512, 512, 3)
        self.conv5 = _down(512, 512, 3)
        self.upconv5, self.iconv5 = _up(512, 512, 3), _same(1024, 512, 3)
        self.upconv4, self.iconv4, self.disp4 = _up(512, 512, 3), _same(1024, 512, 3), _head(512)
        self.upconv3, self.iconv3, self.disp3 = _up(512, 256, 3), _same(513, 256, 3), _head(256)
        self.upconv2, self.iconv2, self.disp2 = _up(256, 128, 3), _same(257, 128, 3), _head(128)
        self.upconv1, self.iconv1, self.disp1 = _up(128, 64, 3), _same(65, 64, 3), _head(64)
        _init(self)

    def forward(self, left, right, left_cam, right_cam):
        s = self.idepth_scale
        vol = plane_sweep_volume(left, right, left_cam, right_cam, s, self.planes)
        c1 = self.conv1(torch.cat((left, vol), 1))                        # :233-235
        c2 = self.conv2(c1); c3 = self.conv3(c2); c4 = self.conv4(c3); c5 = self.conv5(c4)
        i5 = self.iconv5(torch.cat((self.upconv5(c5), c4), 1))            # :241-242
        i4 = self.iconv4(torch.cat((self.upconv4(i5), c3), 1))            # :244-245
        d4 = s * self.disp4(i4)
        u4 = F.interpolate(d4, scale_factor=2, mode="nearest")            # :247 F.upsample
        i3 = self.iconv3(torch.cat((self.upconv3(i4), c2, u4), 1))        # :249-250
        d3 = s * self.disp3(i3)
        u3 = F.interpolate(d3, scale_factor=2, mode="nearest")
        i2 = self.iconv2(torch.cat((self.upconv2(i3), c1, u3), 1))        # :254-255
        d2 = s * self.disp2(i2)
        u2 = F.interpolate(d2, scale_factor=2, mode="nearest")
        i1 = self.iconv1(torch.cat((self.upconv1(i2), u2), 1))            # :259-260
        d1 = s * self.disp1(i1)
        return [d1, d2, d3, d4], i1


class DepthRefineNetCPU(nn.Module):
    """depthNet_model.py:268-370."""

    def __init__(self, base_channels_num=32, idepth_scale=2):
        super().__init__()
        self.base_channels_num, self.idepth_scale = base_channels_num, idepth_scale
        self.conv1, self.conv2, self.conv3 = _down(67, 128, 3), _down(128, 256, 3), _down(256, 512, 3)
        for tag in ("depth", "prob"):                                     # :282-308
            setattr(self, "upconv3_" + tag, _up(512, 256, 3))
            setattr(self, "iconv3_" + tag, _same(512, 256, 3))
            setattr(self, "upconv2_" + tag, _up(256, 128, 3))
            setattr(self, "iconv2_" + tag, _same(256, 128, 3))
            setattr(self, "upconv1_" + tag, _up(128, 64, 3))
            setattr(self, "iconv1_" + tag, _same(64, 64, 3))
            setattr(self, "disp_refine" if tag == "depth" else "prob", _head(64))   # :296, :308
        _init(self)

    def _decode(self, tag, c1, c2, c3):
        g = lambda n: getattr(self, n + "_" + tag)
        i3 = g("iconv3")(torch.cat((g("upconv3")(c3), c2), 1))
        i2 = g("iconv2")(torch.cat((g("upconv2")(i3), c1), 1))
        return g("iconv1")(g("upconv1")(i2))

    def forward(self, idepth01, idepth02, iconv01, iconv02, ReturnVolume=False):
        x = torch.cat((idepth01, idepth02, (idepth01 - idepth02).abs(), iconv01 + iconv02), 1)  # :332-333
        c1 = self.conv1(x); c2 = self.conv2(c1); c3 = self.conv3(c2)
        feat = self._decode("depth", c1, c2, c3)
        disp = self.idepth_scale * self.disp_refine(feat)                 # :351
        prob = self.prob(self._decode("prob", c1, c2, c3))                # :365
        return (disp, prob, feat) if ReturnVolume else (disp, prob)


# --------------------------------------------------------------------------
# depth -> normal                                      depth_util.py:140-203
# --------------------------------------------------------------------------
def backproject(depth, K_inv):
    """p = z * K^-1 (x,y,1)  -> [B,3,H,W].               inverse_warp.py:27-43"""
    B, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype),
                            indexing="ij")
    pix = torch.stack((xs, ys, torch.ones_like(xs)), 0).view(1, 3, -1).expand(B, 3, H * W)
    return K_inv.bmm(pix).view(B, 3, H, W) * depth.unsqueeze(1)


def depth_to_normal(depth, K_inv, k_size=9):
    """Unfold-based least squares exactly as depth_util.py:160-203 -> (normal, points)."""
    B, H, W = depth.shape
    kk = k_size * k_size
    pts = backproject(depth, K_inv)
    valid = ((depth > 0) & (depth < 10.0)).to(depth.dtype).unsqueeze(1)            # :162
    unfold = nn.Unfold(kernel_size=k_size, padding=k_size // 2, stride=1)           # :165
    A = unfold(pts).view(-1, 3, kk, H, W).permute(0, 3, 4, 2, 1)                    # :166-168
    V = unfold(valid).view(-1, 1, kk, H, W).permute(0, 3, 4, 2, 1) > 0.5            # :170-174
    A = torch.where(V.expand_as(A), A, torch.zeros_like(A))                         # :176-177
    At = A.transpose(3, 4).reshape(-1, 3, kk)
    A = A.reshape(-1, kk, 3)
    AtA = torch.bmm(At, A)                                                          # :183
    det = AtA.det()                                                                 # :185
    bad = torch.isnan(det) | (det < 1e-5)                                           # :187
    AtA = torch.where(bad.view(-1, 1, 1), torch.eye(3, dtype=depth.dtype).expand_as(AtA), AtA)  # :190-197
    n = torch.matmul(torch.matmul(torch.inverse(AtA), At), torch.ones(A.shape[0], kk, 1, dtype=depth.dtype)).squeeze(-1)  # :198-200
    n = n / (n.norm(dim=1, keepdim=True) + 1e-5)                                    # :201
    return n.view(B, H, W, 3).permute(0, 3, 1, 2), pts


def plane_normals(normal, instance_segs, planes_num, with_loss=True):
    """Plane branch of Depth2normal.forward (depth_util.py:205-238) / get_normal_by_planes (:243-278) on a [B,3,H,W] map:
    instances in order, each replaced by its mean normal; loss = sum of mean(1 - cos(mean, inside ? n : 0))."""
    n = normal.permute(0, 2, 3, 1).clone()                                            # [B,H,W,3]
    B, H, W, _ = n.shape
    loss = torch.zeros((), dtype=normal.dtype)
    for b in range(B):
        for i in range(int(planes_num[b])):
            m = instance_segs[b, i].bool()                                            # [H,W]
            mean = (n[b] * m.unsqueeze(-1).to(n.dtype)).reshape(-1, 3).sum(0) / m.sum().to(n.dtype)   # :221-225
            reg = mean.expand(H, W, 3)
            orig = torch.where(m.unsqueeze(-1), n[b], torch.zeros_like(n[b]))         # :228
            if with_loss:
                loss = loss + (1 - F.cosine_similarity(reg.reshape(-1, 3), orig.reshape(-1, 3), dim=1)).mean()   # :230-233
            n[b] = torch.where(m.unsqueeze(-1), reg, n[b])                            # :235-236
    return n.permute(0, 3, 1, 2), (loss if with_loss else None)


# --------------------------------------------------------------------------
# inverse warp                                         inverse_warp.py:46-118
# --------------------------------------------------------------------------
def inverse_warp(feat, depth, pose, K, K_inv, padding_mode="zeros"):
    B, _, H, W = feat.shape
    cam = backproject(depth, K_inv).view(B, 3, -1)
    P = K.bmm(pose)                                                                 # :110
    pc = P[:, :, :3].bmm(cam) + P[:, :, 3:]                                         # :57-64
    X, Y, Z = pc[:, 0], pc[:, 1], pc[:, 2].clamp(min=1e-3)                          # :65-67
    xn = 2 * (X / Z) / (W - 1) - 1                                                  # :69
    yn = 2 * (Y / Z) / (H - 1) - 1                                                  # :70
    if padding_mode == "zeros":                                                     # :71-75
        xn = torch.where((xn > 1) | (xn < -1), torch.full_like(xn, 2.0), xn)
        yn = torch.where((yn > 1) | (yn < -1), torch.full_like(yn, 2.0), yn)
    grid = torch.stack((xn, yn), 2).view(B, H, W, 2)
    return F.grid_sample(feat, grid, mode="bilinear", padding_mode=padding_mode, align_corners=False)  # :116


# --------------------------------------------------------------------------
# a "frame" of the headline metric: ref + 2 src           eval.py:440-455
# --------------------------------------------------------------------------
@torch.no_grad()
def frame_forward(depth_net, refine_net, ref, src1, src2, ref_cam, cam1, cam2, k_size=9, normals=True):
    o1, f1 = depth_net(ref, src1, ref_cam, cam1)
    o2, f2 = depth_net(ref, src2, ref_cam, cam2)
    disp, prob = refine_net(idepth01=o1[0], idepth02=o2[0], iconv01=f1, iconv02=f2)
    out = {"disp": disp, "prob": prob, "disp_a": o1[0], "disp_b": o2[0]}
    if normals:
        depth = 1.0 / disp.squeeze(1)                                               # eval.py:452
        K_inv = ref_cam[:, 1, :3, :3].inverse()
        out["normal"], out["points"] = depth_to_normal(depth, K_inv, k_size)        # eval.py:455
    return out


@torch.no_grad()
def frame_forward_multi(depth_net, refine_net, images, cams):
    """S = 4 / 6 sources: ref replicated, sources stacked on the batch axis, ONE depth_net call,
    even/odd outputs averaged into the two refine sides (eval.py:635-663 for S=4, :885-929 for S=6).
    images [1+S,3,H,W], cams [1+S,2,4,4] for ONE frame."""
    S = images.shape[0] - 1
    outs, feat = depth_net(images[0:1].repeat(S, 1, 1, 1), images[1:], cams[0:1].repeat(S, 1, 1, 1), cams[1:])
    d = outs[0]
    if S == 4:
        a, b = (d[0:1] + d[2:3]) * 0.5, (d[1:2] + d[3:4]) * 0.5
        fa, fb = (feat[0:1] + feat[2:3]) * 0.5, (feat[1:2] + feat[3:4]) * 0.5
    elif S == 6:
        a, b = (d[0:1] + d[2:3] + d[4:5]) / 3., (d[1:2] + d[3:4] + d[5:6]) / 3.
        fa, fb = (feat[0:1] + feat[2:3] + feat[4:5]) / 3., (feat[1:2] + feat[3:4] + feat[5:6]) / 3.
    else:
        raise ValueError(S)
    return refine_net(idepth01=a, idepth02=b, iconv01=fa, iconv02=fb)
