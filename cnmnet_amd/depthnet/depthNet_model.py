"""depthNet / DepthRefineNet with the reference's interface, executed by the HIP engine.

Interface parity (reference depthnet/depthNet_model.py):
  depthNet(idepth_scale=3).forward(left_image, right_image, left_cam, right_cam)
      -> ([disp1, disp2, disp3, disp4], iconv1)                                   (:127, :226-263)
  DepthRefineNet(base_channels_num=32, idepth_scale=2).forward(
      idepth01, idepth02, iconv01, iconv02, ReturnVolume=False)
      -> (disp_refined, prob_map[, iconv1_depth])                                 (:273, :331-370)
state_dict keys and shapes are identical to the reference's (nn.Sequential indices
.0/.1/.3/.4 in down blocks, .1/.2 in up blocks, .0 in heads), so its checkpoints load.

The torch modules below only CONTAIN parameters.  forward() folds eval-mode BatchNorm
into packed weights on the device (cached until a parameter changes) and makes ONE call
into the C ABI (cnm_depthnet_forward_f32 / cnm_refinenet_forward_f32).  There is no
eager-torch fallback.  A module left on the CPU runs eval-mode fp32 forwards through the
library's host twins (cnm_depthnet_forward_cpu / cnm_refinenet_forward_cpu: plain C++ in the
same library, BASELINE configs[0]); training and the f16 engine need the GPU.
"""
import ctypes

import os
import torch
import torch.nn as nn

from .. import _lib, host, ops
from .. import autograd as ag


class _Slot(nn.Identity):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference
    (ReLU / Upsample / Sigmoid positions); the engine fuses or runs those ops itself."""


def _seq(*mods):
    return nn.Sequential(*mods)


def _cbr(cin, cout, k, stride=1):
    return [nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False), nn.BatchNorm2d(cout), _Slot()]


def _down(cin, cout, k):
    return _seq(*(_cbr(cin, cout, k, 1) + _cbr(cout, cout, k, 2)))


def _same(cin, cout, k):
    return _seq(*_cbr(cin, cout, k))


def _up(cin, cout, k):
    return _seq(_Slot(), *_cbr(cin, cout, k))


def _head(cin):
    return _seq(nn.Conv2d(cin, 1, 3, padding=1), _Slot())


def _init_like_reference(module):
    """Kaiming-normal fan_out convs, BN (1,0), zero head bias (reference :165-182)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class _EngineNet(nn.Module):
    _NET = None

    def _engine_init(self, precision="f32"):
        if precision not in ("f32", "f16"):
            raise ValueError("precision must be 'f32' or 'f16'")
        self.precision = precision  # 'f16': fp16 storage + f16 MFMA (BASELINE config 5); eval only
        self._layers = None        # engine layer table (resolved lazily: needs the library)
        self.winograd = os.environ.get("CNM_WINOGRAD", "1") != "0"   # fp32 stride-1 layers in the Winograd domain
        self.winograd4 = os.environ.get("CNM_WINOGRAD4", "1") != "0"  # large 3x3 layers: F(4x4,3x3) instead of F(2x2,3x3)
        self.quad_all = os.environ.get("CNM_QUAD_ALL", "0") == "1"     # pack the four-wave kernel's filter for EVERY 3x3 stride-1 layer (A/B with cnm_tune_wino36_quad(2))
        self.fused_upsample = os.environ.get("CNM_FUSED_UPSAMPLE", "1") != "0"   # up_conv layers: upsample folded into the conv (composed phase filters)
        self._packed = None        # [(w, b[, u])] device tensors, one per engine layer
        self._packed_key = None
        self._weights_arr = None
        self._ws = {}

    # -- parameter lookup by the engine's state_dict-style keys ("conv1.0", "upconv3_depth.2")
    def _sub(self, key):
        name, idx = key.split(".")
        return getattr(self, name)[int(idx)]

    def _param_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return (self.precision, self.winograd, self.winograd4, self.fused_upsample, str(ts[0].device), tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts[:4]))

    def _first_cin(self):
        return None

    def _ensure_packed(self):
        key = self._param_key()
        if self._packed is not None and key == self._packed_key:
            return
        if self._layers is None:
            self._layers = _lib.net_layers(self._NET)
        packed = []
        on_host = not next(self.parameters()).is_cuda
        for i, L in enumerate(self._layers):
            conv = self._sub(L["conv_key"])
            w = conv.weight.detach()
            if on_host:                                                   # host twins: BatchNorm-folded direct filters only (cnm_pack_conv_bn_cpu)
                if L["is_head"]:
                    packed.append((host.pack_head(w.float()), conv.bias.detach().float().contiguous()))
                else:
                    bn = self._sub(L["bn_key"])
                    packed.append(host.pack_conv(w.float(), (bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var), rot=L["rot"], eps=bn.eps))
                continue
            if L["is_head"]:
                packed.append((ops.pack_head(w), conv.bias.detach().contiguous()))
            else:
                bn = self._sub(L["bn_key"])
                pack = ops.pack_conv_f16 if self.precision == "f16" else ops.pack_conv
                bnp = (bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var)
                wp, bp = pack(w, bnp, rot=L["rot"], eps=bn.eps)
                # fp32 layers in the Winograd domain: 3x3 stride 1 (F(2x2,3x3)); 5x5 / 7x7 stride 1 and 2 (row-wise)
                wino = (self.precision == "f32" and self.winograd and L["Cout"] % 64 == 0 and
                        (L["ksize"] == 3 or L["ksize"] in (5, 7)))            # 3x3 stride 2: F(2x2) filter for the low-resolution layers (nets.hip)
                if not wino:
                    # fp16 up_conv layers the executor may run fused with their bilinear upsampling (<= 256 input channels)
                    if (self.precision == "f16" and L["ksize"] == 3 and L["stride"] == 1 and L["conv_key"].startswith("upconv")
                            and L["Cin"] <= 256 and L["Cout"] % 64 == 0 and self.fused_upsample):
                        packed.append((wp, bp, None, None) + tuple(ops.pack_upsampled_f16(w, bnp, eps=bn.eps)))
                    else:
                        packed.append((wp, bp))
                else:
                    up = ops.pack_winograd(w, bnp, rot=L["rot"], eps=bn.eps, stride=L["stride"])
                    # 3x3 / 5x5 stride 1: also the 36-point filter (F(4x4,3x3) / F(2x2,5x5)); the executor picks per call by tile count
                    u4 = (ops.pack_winograd4(w, bnp, rot=L["rot"], eps=bn.eps)
                          if (L["ksize"] in (3, 5) and L["stride"] == 1 and self.winograd4) else None)
                    if L["ksize"] == 3 and L["stride"] == 2 and self.winograd4:      # 3x3 stride 2: u = F(2x2) filter (small layers), u4 = F(4,2) row phases
                        u4 = ops.pack_winograd_rows(w, bnp, rot=L["rot"], eps=bn.eps, stride=2, tile=4)
                    if L["ksize"] in (5, 7) and L["stride"] == 2 and self.winograd4 and L["Cout"] % 128 == 0:   # stride-2 5x5 / 7x7: the four pixel phases on the 36-point kernel
                        u4 = ops.pack_winograd4_s2(w, bnp, rot=L["rot"], eps=bn.eps)
                    # up_conv layers the executor may run fused with their bilinear upsampling (<= 256 input channels)
                    fused = (ops.pack_winograd4_upsampled(w, bnp, eps=bn.eps)
                             if (u4 is not None and L["ksize"] == 3 and L["conv_key"].startswith("upconv") and L["Cin"] <= 256
                                 and self.fused_upsample) else (None, None, None))
                    if L["ksize"] == 3 and L["stride"] == 2 and self.winograd4 and L["Cout"] % 128 == 0:
                        # 3x3 stride 2: a third filter, in the uu slot (bu / wr stay empty: not an up_conv) -- the pixel-phase form for the
                        # shapes that would otherwise run the implicit GEMM (nets.hip)
                        fused = (ops.pack_winograd4_s2(w, bnp, rot=L["rot"], eps=bn.eps), None, None)
                    # [r6] 3x3 stride 1: the quad re-ordering of u4 for the four-wave kernel -- by default only where the executor uses it (Cout not a
                    # multiple of 128: the iconv1 layers); quad_all packs it for every such layer (A/B through cnm_tune_wino36_quad(2))
                    u4q = (ops.repack_winograd4_quad(u4, L["Cout"], L["Cin"])
                           if (u4 is not None and L["ksize"] == 3 and L["stride"] == 1 and (L["Cout"] % 128 != 0 or getattr(self, "quad_all", False))) else None)
                    packed.append((wp, bp, up, u4) + tuple(fused) + (u4q,))
        arr = (_lib.LayerWeights * len(packed))()
        for i, t in enumerate(packed):
            arr[i].w, arr[i].b = t[0].data_ptr(), t[1].data_ptr()
            arr[i].u = t[2].data_ptr() if len(t) > 2 and t[2] is not None else None
            arr[i].u4 = t[3].data_ptr() if len(t) > 3 and t[3] is not None else None
            arr[i].uu, arr[i].bu, arr[i].wr = [t[j].data_ptr() if len(t) > 6 and t[j] is not None else None for j in (4, 5, 6)]
            arr[i].u4q = t[7].data_ptr() if len(t) > 7 and t[7] is not None else None
        self._packed, self._weights_arr, self._packed_key = packed, arr, key

    def _workspace(self, device, nfloats):
        """The net's long-lived workspace.  Its head holds state that must be zero when a call starts and is left zero when
        the call's kernels have run (the plane sweep's tile queue, the sync words of the staged convolutions), so two calls
        may never overlap in it: a call on a different stream than the previous one first waits for that stream."""
        ws = self._ws.get(str(device))
        if ws is None or ws.numel() < nfloats:
            self._ws = {str(device): ops.register_sync_owner(torch.zeros(nfloats, device=device, dtype=torch.float32))}
            ws = self._ws[str(device)]
            if not torch.cuda.is_current_stream_capturing():
                with torch.cuda.device(device):
                    ops.engine_status(clear=False, device=device)       # allocates this device's status word before anybody captures a graph over this net
                    if self._NET == _lib.NET_DEPTH:
                        ops.calibrate_sweep_store(device)                # [r6] the plane sweep's store policy: decided here, once per device, never by a launch
        cur = torch.cuda.current_stream(device)
        last = getattr(self, "_ws_stream", None)
        if last is not None and last != cur and not torch.cuda.is_current_stream_capturing():
            cur.wait_stream(last)
        self._ws_stream = cur
        return ws

    def _require_gpu(self, *tensors):
        ops._dev(*tensors)
        p = next(self.parameters())
        if not p.is_cuda or p.device != tensors[0].device:
            raise _lib.EngineError("module parameters are on %s but inputs are on %s" % (p.device, tensors[0].device))

    def _on_host(self, *tensors):
        """True when this call runs on the host twins: module and every input on the CPU, eval mode, fp32.  A CPU call that
        cannot (training, f16, mixed devices) fails loudly -- there is no silent fallback in either direction."""
        if next(self.parameters()).is_cuda or not host.is_host(*tensors):
            return False
        if self.training or self.precision != "f32":
            raise _lib.EngineError("the host twins run eval-mode fp32 inference only; training and the f16 engine need the GPU")
        return True


def _down_train(x, seq, rot=0, groups=1):
    """down_conv_layer in train mode: conv s1 -> BN -> ReLU -> conv s2 -> BN -> ReLU (reference :19-39)."""
    return ag.conv_bn_relu(ag.conv_bn_relu(x, seq[0], seq[1], rot, groups), seq[3], seq[4], 0, groups)


def _up_train(x, seq, groups=1):
    """up_conv_layer in train mode: bilinear x2 -> conv -> BN -> ReLU (reference :91-101)."""
    return ag.conv_bn_relu(ag.Upsample2xC4.apply(x), seq[1], seq[2], 0, groups)


FRAME_VIEWS = os.environ.get("CNM_FRAME_VIEWS", "1") != "0"   # 0: copy sliced inputs as before (A/B)


def _frame_view(t, unit=1):
    """(tensor, floats between consecutive frames): `t` itself when every frame t[b] is dense and the frames are at least a frame apart and a
    multiple of `unit` floats apart (a slice of a larger frame tensor; the plane sweep wants whole images), else a contiguous copy."""
    if t.dtype != torch.float32:
        t = t.float()
    inner = t[0].numel()
    if FRAME_VIEWS and t[0].is_contiguous() and (t.shape[0] == 1 or (t.stride(0) >= inner and t.stride(0) % unit == 0 and t.stride(0) // unit < 65536)):
        return t, (t.stride(0) if t.shape[0] > 1 else inner)
    t = t.contiguous()
    return t, inner


class depthNet(_EngineNet):
    """Plane-sweep cost volume + hourglass regression (reference depthNet_model.py:124-263).

    ``planes`` (default 64, what the reference hard-codes at :194,:199,:208 and in conv1's 67
    input channels) is exposed for the 32/96-plane configurations; state_dict compatibility
    with reference checkpoints holds at planes=64."""
    _NET = _lib.NET_DEPTH

    def __init__(self, idepth_scale=3, planes=64, precision="f32"):
        super().__init__()
        if planes % 4 or not (4 <= planes <= 128) or (precision == "f16" and planes % 8):
            raise ValueError("planes must be a multiple of 4 (8 for f16) in [4,128]")
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
        _init_like_reference(self)
        self._engine_init(precision)

    max_call_bytes = 3.9e9      # the convolution kernels address an activation tensor with 32-bit byte offsets

    def forward_pairs(self, ref, src, ref_cam, src_cam):
        """ref [B,3,H,W], src [B,S,3,H,W], ref_cam [B,2,4,4], src_cam [B,S,2,4,4]
        -> ([disp1..4] each [B*S,1,h,w], iconv1 as c4 [B*S,16,H,W,4]); pair p = b*S + s.
        One engine call for all pairs; the reference replicates the ref image instead
        (eval.py:635-657)."""
        ops.idepth_range(self.idepth_scale)
        B, S, _, H, W = src.shape
        if self._on_host(ref, src, ref_cam, src_cam):
            if H % 32 or W % 32:
                raise ValueError("image height and width must be multiples of 32 (got %dx%d)" % (H, W))
            self._ensure_packed()
            return host.depthnet_forward(self._weights_arr, self.idepth_scale, self.planes, ref, src, ref_cam, src_cam)
        self._require_gpu(ref, src, ref_cam, src_cam)
        if H % 32 or W % 32:
            raise ValueError("image height and width must be multiples of 32 (got %dx%d)" % (H, W))
        if self.training:
            if self.precision != "f32":
                raise NotImplementedError("training runs in fp32; the f16 engine is inference-only")
            return self._forward_train(ref, src, ref_cam, src_cam)
        # the conv kernels address activations with 32-bit byte offsets (< 4 GB per tensor); the widest one holds
        # 128 channels at full resolution = 512 B (fp32) per pixel and pair -> split very large batches by frames
        max_pairs = int(self.max_call_bytes // ((512 if self.precision == "f32" else 256) * H * W))
        if B * S > max_pairs and B > 1:
            nb = max(1, max_pairs // S)
            parts = [self.forward_pairs(ref[i:i + nb], src[i:i + nb], ref_cam[i:i + nb], src_cam[i:i + nb]) for i in range(0, B, nb)]
            return [torch.cat([p[0][k] for p in parts], 0) for k in range(4)], torch.cat([p[1] for p in parts], 0)
        self._ensure_packed()
        lib, P, dev = _lib.load(), B * S, ref.device
        # [r6] the four inputs are usually SLICES of the caller's frame tensors (images[:, 0], images[:, 1:], the reference's own eval.py:440-447):
        # dense inside a frame, a larger stride from frame to frame.  The engine reads them where they lie (cnm_depthnet_forward_strided_*);
        # anything else is made contiguous first (four copy launches per call: 1 % of an fp16 step).
        ref, rs = _frame_view(ref, 3 * H * W); src, ss = _frame_view(src, 3 * H * W); ref_cam, rcs = _frame_view(ref_cam); src_cam, scs = _frame_view(src_cam)
        disp = [torch.empty(P, 1, H >> i, W >> i, device=dev, dtype=torch.float32) for i in range(4)]
        f16 = self.precision == "f16"
        feat = (torch.empty(P, 8, H, W, 8, device=dev, dtype=torch.float16) if f16
                else torch.empty(P, 16, H, W, 4, device=dev, dtype=torch.float32))
        n = (lib.cnm_depthnet_workspace_floats_f16 if f16 else lib.cnm_depthnet_workspace_floats)(P, H, W, self.planes)
        ws = self._workspace(dev, n)
        with torch.cuda.device(dev):
            _lib.check((lib.cnm_depthnet_forward_strided_f16 if f16 else lib.cnm_depthnet_forward_strided_f32)(
                self._weights_arr, float(self.idepth_scale), self.planes,
                ref.data_ptr(), rs, src.data_ptr(), ss, ref_cam.data_ptr(), rcs, src_cam.data_ptr(), scs,
                disp[0].data_ptr(), disp[1].data_ptr(), disp[2].data_ptr(), disp[3].data_ptr(), feat.data_ptr(),
                ws.data_ptr(), ws.numel(), B, S, H, W, torch.cuda.current_stream().cuda_stream))
        return disp, feat

    def _forward_train(self, ref, src, ref_cam, src_cam, per_source_statistics=False):
        """Train mode (batch-statistics BatchNorm, autograd): the same graph as reference :226-263 built
        from cnmnet_amd.autograd Functions; the cost volume is a constant of the graph.
        per_source_statistics: every BatchNorm normalises pair p = b * S + s with the statistics of source s over the batch (and updates
        the running statistics once per source, in order) -- S calls with one source each, as the reference makes them, in one pass."""
        s = float(self.idepth_scale)
        gr = src.shape[1] if per_source_statistics else 1
        with torch.no_grad():
            hmkt = ops.homography_terms(ref_cam, src_cam)
            x0 = ops.plane_sweep_cat_c4(ref, src, hmkt, s, self.planes)
        cbr = lambda x, seq: ag.conv_bn_relu(x, seq[0], seq[1], 0, gr)
        c1 = _down_train(x0, self.conv1, 3, gr)
        c2 = _down_train(c1, self.conv2, 0, gr); c3 = _down_train(c2, self.conv3, 0, gr)
        c4 = _down_train(c3, self.conv4, 0, gr); c5 = _down_train(c4, self.conv5, 0, gr)
        # what the ENCODER hands to the decoder goes through identity nodes that only the decoder reads (the encoder itself continues
        # on the raw tensors, so the nodes are not ancestors of one another): a caller that replays backward in pieces -- the
        # trainer's segmented HIP-graph step -- can stop the decoder's backward at them and start the encoder's from their gradients
        c1d, c2d, c3d, c4d, c5d = (t.view_as(t) for t in (c1, c2, c3, c4, c5))
        if getattr(self, "_enc_cut", None) is not None:                  # a list set by the trainer for the duration of one forward; otherwise nothing is kept
            self._enc_cut += [c1d, c2d, c3d, c4d, c5d]
        c1, c2, c3, c4, c5 = c1d, c2d, c3d, c4d, c5d
        i5 = cbr(torch.cat((_up_train(c5, self.upconv5, gr), c4), 1), self.iconv5)
        i4 = cbr(torch.cat((_up_train(i5, self.upconv4, gr), c3), 1), self.iconv4)
        d4 = ag.head(i4, self.disp4[0], s)
        g4 = ag.scalar_maps_to_group(ag.nearest_up2(d4))
        i3 = cbr(torch.cat((_up_train(i4, self.upconv3, gr), c2, g4), 1), self.iconv3)
        d3 = ag.head(i3, self.disp3[0], s)
        g3 = ag.scalar_maps_to_group(ag.nearest_up2(d3))
        i2 = cbr(torch.cat((_up_train(i3, self.upconv2, gr), c1, g3), 1), self.iconv2)
        d2 = ag.head(i2, self.disp2[0], s)
        g2 = ag.scalar_maps_to_group(ag.nearest_up2(d2))
        i1 = cbr(torch.cat((_up_train(i2, self.upconv1, gr), g2), 1), self.iconv1)
        d1 = ag.head(i1, self.disp1[0], s)
        return [d1, d2, d3, d4], i1

    def forward_sources(self, left_image, right_images, left_cam, right_cams):
        """Train mode: depthNet(left, right_images[:, s], ...) for every source s, as S separate calls would compute them -- same
        outputs, same BatchNorm running statistics, same gradients -- in ONE pass over B * S pairs (per-source BatchNorm statistics).
        right_images [B,S,3,H,W], right_cams [B,S,2,4,4] -> [(disp list, iconv1 NCHW) for each source]."""
        if not self.training or self.precision != "f32":
            return [self.forward(left_image, right_images[:, s], left_cam, right_cams[:, s]) for s in range(right_images.shape[1])]
        ops.idepth_range(self.idepth_scale)
        self._require_gpu(left_image, right_images, left_cam, right_cams)
        B, S, _, H, W = right_images.shape
        if H % 32 or W % 32:
            raise ValueError("image height and width must be multiples of 32 (got %dx%d)" % (H, W))
        # every activation of the one-pass form is S times as large as in S calls, and the convolution kernels address a
        # tensor with 32-bit byte offsets (512 B per pixel and pair at full resolution): past that, make the S calls
        if S * B * 512 * H * W > self.max_call_bytes:
            return [self.forward(left_image, right_images[:, s], left_cam, right_cams[:, s]) for s in range(S)]
        disp, feat = self._forward_train(left_image.contiguous(), right_images.contiguous(), left_cam.contiguous(), right_cams.contiguous(),
                                         per_source_statistics=True)
        feats = ag.SplitSources.apply(feat, S)                          # sample n of the pass belongs to source n % S
        disps = [ag.SplitSources.apply(d, S) for d in disp]
        out = []
        for s in range(S):
            iconv1 = ag.C4ToNCHW.apply(feats[s], 64)
            iconv1._cnm_c4 = feats[s]
            out.append(([d[s] for d in disps], iconv1))
        return out

    def getVolume(self, left_image, right_image, KRKiUV_T, KT_T):
        """Reference depthNet_model.py:185-224: plane-sweep L1 cost volume [B,planes,H,W] from what
        process_camera_parameters returns.  KRKiUV_T is either that function's result (the 12 camera terms ride
        along), a plain [B,3,W*H] grid product in the reference's u-major order (the homography is recovered from
        it), or the homography itself [B,3,3]."""
        from .depth_util import homography_from_grid_product
        if not host.is_host(left_image, right_image, KRKiUV_T, KT_T):
            self._require_gpu(left_image, right_image, KRKiUV_T, KT_T)
        B, _, H, W = left_image.shape
        hmkt = getattr(KRKiUV_T, "hmkt", None)
        if hmkt is None:
            if tuple(KRKiUV_T.shape[1:]) == (3, 3) and H * W != 3:
                hmkt = torch.cat((KRKiUV_T.reshape(B, 9), KT_T.reshape(B, 3)), 1).contiguous()
            else:
                if tuple(KRKiUV_T.shape) != (B, 3, H * W):
                    raise ValueError("KRKiUV_T must be [B,3,H*W] = %s, got %s" % ((B, 3, H * W), tuple(KRKiUV_T.shape)))
                hmkt = homography_from_grid_product(KRKiUV_T, KT_T, H, W)
        return ops.plane_sweep_volume_hmkt(left_image, right_image, hmkt.float(), self.idepth_scale, self.planes)

    def forward(self, left_image, right_image, left_cam, right_cam):
        disp, feat_c4 = self.forward_pairs(left_image, right_image.unsqueeze(1), left_cam, right_cam.unsqueeze(1))
        if self.precision == "f16":
            iconv1 = ops.c8_to_nchw(feat_c4, 64)
            iconv1._cnm_c8 = feat_c4
            return disp, iconv1
        iconv1 = ag.C4ToNCHW.apply(feat_c4, 64) if self.training else ops.c4_to_nchw(feat_c4, 64)
        iconv1._cnm_c4 = feat_c4           # lets DepthRefineNet skip the NCHW->c4 round trip
        return disp, iconv1


class DepthRefineNet(_EngineNet):
    """Occlusion-aware two-view fusion (reference depthNet_model.py:268-370)."""
    _NET = _lib.NET_REFINE

    def __init__(self, base_channels_num=32, idepth_scale=2, precision="f32"):
        super().__init__()
        self.base_channels_num, self.idepth_scale = base_channels_num, idepth_scale
        self.conv1, self.conv2, self.conv3 = _down(67, 128, 3), _down(128, 256, 3), _down(256, 512, 3)
        for tag in ("depth", "prob"):
            setattr(self, "upconv3_" + tag, _up(512, 256, 3))
            setattr(self, "iconv3_" + tag, _same(512, 256, 3))
            setattr(self, "upconv2_" + tag, _up(256, 128, 3))
            setattr(self, "iconv2_" + tag, _same(256, 128, 3))
            setattr(self, "upconv1_" + tag, _up(128, 64, 3))
            setattr(self, "iconv1_" + tag, _same(64, 64, 3))
            setattr(self, "disp_refine" if tag == "depth" else "prob", _head(64))
        _init_like_reference(self)
        self._engine_init(precision)

    def forward_c4(self, idepth01, idepth02, idepth_stride, f1, G1_total, g1, f2, G2_total, g2, N, H, W, return_volume=False):
        """Raw-view entry used by the frame pipeline (no layout conversion)."""
        self._ensure_packed()
        if self._on_host(idepth01, idepth02, f1, f2):
            return host.refinenet_forward(self._weights_arr, self.idepth_scale, idepth01, idepth02, idepth_stride, f1, G1_total, g1, f2, G2_total, g2,
                                          N, H, W, return_volume)
        lib, dev = _lib.load(), idepth01.device
        disp = torch.empty(N, 1, H, W, device=dev, dtype=torch.float32)
        prob = torch.empty_like(disp)
        vol = torch.empty(N, 16, H, W, 4, device=dev, dtype=torch.float32) if return_volume else None
        ws = self._workspace(dev, lib.cnm_refinenet_workspace_floats(N, H, W))
        with torch.cuda.device(dev):
            _lib.check(lib.cnm_refinenet_forward_f32(
                self._weights_arr, float(self.idepth_scale), idepth01.data_ptr(), idepth02.data_ptr(), idepth_stride,
                f1.data_ptr(), G1_total, g1, f2.data_ptr(), G2_total, g2,
                disp.data_ptr(), prob.data_ptr(), vol.data_ptr() if vol is not None else 0,
                ws.data_ptr(), ws.numel(), N, H, W, torch.cuda.current_stream().cuda_stream))
        return disp, prob, vol

    def forward_multi(self, idepth_pairs, feat_pairs_c4, S, return_volume=False):
        """S (even) sources per frame from ONE depthNet.forward_pairs call: disp1 [B*S,1,H,W] and
        iconv1 c4 [B*S,16,H,W,4]; even sources average into side 1, odd into side 2
        (reference eval.py:656-663 for S=4, :917-929 for S=6; S=2 is the plain two-view case)."""
        if self._on_host(idepth_pairs, feat_pairs_c4):
            self._ensure_packed()
            return host.refinenet_forward_multi(self._weights_arr, self.idepth_scale, idepth_pairs, feat_pairs_c4, S, return_volume)
        self._require_gpu(idepth_pairs)
        self._ensure_packed()
        P, _, H, W = idepth_pairs.shape
        B = P // S
        lib, dev = _lib.load(), idepth_pairs.device
        f16 = self.precision == "f16"
        if f16 != (feat_pairs_c4.dtype == torch.float16):
            raise _lib.EngineError("feature layout/dtype does not match the refine net's precision (%s)" % self.precision)
        disp = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        prob = torch.empty_like(disp)
        vol = None
        if return_volume:
            vol = (torch.empty(B, 8, H, W, 8, device=dev, dtype=torch.float16) if f16
                   else torch.empty(B, 16, H, W, 4, device=dev, dtype=torch.float32))
        ws = self._workspace(dev, lib.cnm_refinenet_workspace_floats(B, H, W))
        with torch.cuda.device(dev):
            _lib.check((lib.cnm_refinenet_forward_multi_f16 if f16 else lib.cnm_refinenet_forward_multi_f32)(
                self._weights_arr, float(self.idepth_scale), idepth_pairs.contiguous().data_ptr(), feat_pairs_c4.data_ptr(), S,
                disp.data_ptr(), prob.data_ptr(), vol.data_ptr() if vol is not None else 0,
                ws.data_ptr(), ws.numel(), B, H, W, torch.cuda.current_stream().cuda_stream))
        return disp, prob, vol

    def _forward_train(self, idepth01, idepth02, f1, f2, ReturnVolume):
        """Train mode: reference :331-370 from cnmnet_amd.autograd Functions (c4 tensors throughout)."""
        grp = ag.scalar_maps_to_group(idepth01, idepth02, (idepth01 - idepth02).abs())
        x = torch.cat((f1 + f2, grp), 1)                               # rotated order: 64 features, then the 3 maps
        c1 = _down_train(x, self.conv1, rot=3); c2 = _down_train(c1, self.conv2); c3 = _down_train(c2, self.conv3)
        u3 = ag.Upsample2xC4.apply(c3)

        def decode(tag):
            g = lambda n: getattr(self, n + "_" + tag)
            uc3 = ag.conv_bn_relu(u3, g("upconv3")[1], g("upconv3")[2])
            i3 = ag.conv_bn_relu(torch.cat((uc3, c2), 1), g("iconv3")[0], g("iconv3")[1])
            i2 = ag.conv_bn_relu(torch.cat((_up_train(i3, g("upconv2")), c1), 1), g("iconv2")[0], g("iconv2")[1])
            return ag.conv_bn_relu(_up_train(i2, g("upconv1")), g("iconv1")[0], g("iconv1")[1])

        feat = decode("depth")
        disp = ag.head(feat, self.disp_refine[0], float(self.idepth_scale))
        prob = ag.head(decode("prob"), self.prob[0], 1.0)
        if not ReturnVolume:
            return disp, prob
        vol = ag.C4ToNCHW.apply(feat, 64)
        vol._cnm_c4 = feat
        return disp, prob, vol

    def forward(self, idepth01, idepth02, iconv01, iconv02, ReturnVolume=False):
        if not self._on_host(idepth01, idepth02, iconv01, iconv02):
            self._require_gpu(idepth01, idepth02, iconv01, iconv02)
        N, _, H, W = idepth01.shape
        if H % 8 or W % 8:
            raise ValueError("image height and width must be multiples of 8 (got %dx%d)" % (H, W))
        f1 = getattr(iconv01, "_cnm_c4", None)
        f2 = getattr(iconv02, "_cnm_c4", None)
        if self.precision == "f16":
            if self.training:
                raise NotImplementedError("training runs in fp32; the f16 engine is inference-only")
            h1 = getattr(iconv01, "_cnm_c8", None); h2 = getattr(iconv02, "_cnm_c8", None)
            h1 = h1 if h1 is not None else ops.nchw_to_c8(iconv01)
            h2 = h2 if h2 is not None else ops.nchw_to_c8(iconv02)
            pairs_f = torch.stack((h1, h2), 1).reshape(2 * N, 8, H, W, 8)          # pair p = 2n + side
            pairs_d = torch.stack((idepth01, idepth02), 1).reshape(2 * N, 1, H, W)
            disp, prob, vol = self.forward_multi(pairs_d, pairs_f, 2, ReturnVolume)
            if not ReturnVolume:
                return disp, prob
            out = ops.c8_to_nchw(vol, 64)
            out._cnm_c8 = vol
            return disp, prob, out
        if self.training:
            f1 = f1 if f1 is not None else ag.NCHWToC4.apply(iconv01)
            f2 = f2 if f2 is not None else ag.NCHWToC4.apply(iconv02)
            return self._forward_train(idepth01, idepth02, f1, f2, ReturnVolume)
        f1 = f1 if f1 is not None else ops.nchw_to_c4(iconv01)
        f2 = f2 if f2 is not None else ops.nchw_to_c4(iconv02)
        disp, prob, vol = self.forward_c4(idepth01.contiguous(), idepth02.contiguous(), H * W,
                                          f1, 16, 0, f2, 16, 0, N, H, W, ReturnVolume)
        if not ReturnVolume:
            return disp, prob
        iconv1_depth = ops.c4_to_nchw(vol, 64)
        iconv1_depth._cnm_c4 = vol
        return disp, prob, iconv1_depth
