"""GPU (-m gpu): the fp16 path (BASELINE config 5: MFMA-f16 convs + fp16 cost-volume storage) against the fp32
oracle.  STATED TOLERANCE: fp16 storage carries 11 bits (5e-4 relative) per activation through 24 conv layers
with fp32 accumulation.  Measured on the golden frame (random weights, inverse-depth range [0,3]): depthNet disp1
max 1.3e-2 / q99.9 1.1e-2, iconv features 1.0e-3 of their maximum, refined inverse depth max 3.1e-2, probability
5.8e-3 (the fp32 path on the same frame: 3.6e-5 / 3.5e-6 / 4.8e-5 / 8.2e-6).  Bars: 3e-2 on depthNet inverse depth,
6e-2 on the refined map, 2e-2 on probability, 5e-3 relative on features.  Normals are computed in fp32/fp64 from the
fp16 net's depth and inherit its depth error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cnmnet_amd import synthetic as syn
from conftest import torch_state
from oracle import ref_arrangement as ra

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True)))
    return module.eval()


@pytest.mark.parametrize("cin,cout,k,stride,rot,N,H,W", [(67, 128, 7, 1, 3, 2, 24, 40), (128, 128, 7, 2, 0, 1, 32, 32),
                                                          (513, 256, 3, 1, 0, 1, 12, 20), (65, 64, 3, 1, 0, 3, 20, 28), (512, 512, 3, 2, 0, 2, 6, 8),
                                                          (104, 128, 5, 2, 0, 2, 12, 16), (80, 64, 3, 1, 0, 1, 10, 14), (40, 64, 3, 2, 0, 2, 9, 11), (32, 128, 7, 1, 0, 1, 8, 8)])
@pytest.mark.parametrize("tile", [0, 1, 3, 4, 5])
def test_conv_f16_vs_fp32_torch(dev, cin, cout, k, stride, rot, N, H, W, tile):
    """conv_glds_kernel on small shapes: pixel tail, ragged K (9 / 65 channel groups; 13 / 10 / 5 / 4 groups: left-over steps of five, two, five
    and four groups, with and without full 8-group blocks before them -- the incremental k-walk), stride 2, rotation; tile 0 = the
    automatic choice, 1 = 128 x 256 (64 x 512 for the 64-cout layer), 3 = 64 x 128, 4 = 128 x 512, 5 = 256 x 256 (falling
    back as the header documents when Cout does not divide)."""
    from cnmnet_amd import _lib, ops
    lib = _lib.load()
    old_tile = lib.cnm_tune_glds_tile(tile)
    try:
        _conv_f16_case(dev, cin, cout, k, stride, rot, N, H, W)
    finally:
        lib.cnm_tune_glds_tile(old_tile)


@pytest.mark.parametrize("cin,cout,k,rot,N,H,W", [(67, 128, 7, 3, 2, 16, 256), (128, 256, 5, 0, 1, 32, 128), (65, 64, 3, 0, 2, 16, 64), (257, 128, 3, 0, 1, 32, 32),
                                                    (64, 64, 3, 0, 1, 8, 32), (513, 256, 3, 0, 1, 16, 64), (128, 512, 3, 0, 3, 8, 32), (72, 64, 5, 0, 1, 8, 64)])
def test_conv_f16_row_extended_kernel(dev, cin, cout, k, rot, N, H, W):
    """conv_gldsx_kernel [r5] (stride 1, W a power of two >= 32, H W a multiple of 256: the pixel operand of a filter row staged once and
    read shifted by the ks taps): every tile instance (Cout 64 / 128 / 256+), tiles of one row / several rows (W = 256 .. 32), filter
    sizes 3 / 5 / 7, group counts 8 n (no left-over step) and 8 n + 1 (the one-group left-over step: taps as slots), rotation -- against
    torch on the fp16-rounded operands, and against conv_glds_kernel on the same packed filter (same products, other summation order)."""
    from cnmnet_amd import _lib, ops
    lib = _lib.load()
    assert lib.cnm_tune_gldsx(1) in (0, 1)
    try:
        got_x = _conv_f16_case(dev, cin, cout, k, 1, rot, N, H, W)
        lib.cnm_tune_gldsx(0)
        got_o = _conv_f16_case(dev, cin, cout, k, 1, rot, N, H, W)
    finally:
        lib.cnm_tune_gldsx(1)
    assert np.abs(got_x - got_o).max() <= 2e-3 * np.abs(got_o).max()


def test_conv_f16_row_extended_two_inputs(dev):
    """The row-extended kernel reading two input views (torch.cat without the copy: 32 + 1 channel groups, the split inside the last block)."""
    from cnmnet_amd import _lib, ops
    rng = np.random.default_rng(5)
    N, H, W, ca, cb, cout = 2, 16, 64, 256, 8, 128
    xa, xb = T(rng.standard_normal((N, ca, H, W)).astype(np.float32)), T(rng.standard_normal((N, cb, H, W)).astype(np.float32))
    w = T((rng.standard_normal((cout, ca + cb, 3, 3)) * 0.05).astype(np.float32)); bias = T(rng.normal(0, 0.2, cout).astype(np.float32))
    want = F.relu(F.conv2d(torch.cat((xa, xb), 1).half().float(), w.half().float(), bias, padding=1)).numpy()
    wp, bp = ops.pack_conv_f16(w.to(dev), None, bias.to(dev), 0)
    a8, b8 = ops.nchw_to_c8(xa.to(dev)), ops.nchw_to_c8(xb.to(dev))
    out = torch.empty(N, cout // 8, H, W, 8, device=dev, dtype=torch.float16)
    lib = _lib.load()
    _lib.check(lib.cnm_conv2d_cat2_c8_f16(a8.data_ptr(), ca // 8, 0, ca // 8, b8.data_ptr(), 1, 0, 1, out.data_ptr(), cout // 8, 0, cout,
                                          wp.data_ptr(), bp.data_ptr(), N, H, W, 3, 1, 1, torch.cuda.current_stream().cuda_stream))
    got = ops.c8_to_nchw(out, cout).cpu().numpy()
    assert np.abs(got - want).max() < 1.5e-3 * np.abs(want).max()


def _conv_f16_case(dev, cin, cout, k, stride, rot, N, H, W):
    from cnmnet_amd import ops
    rng = np.random.default_rng(cin + k)
    x = T(rng.standard_normal((N, cin, H, W)).astype(np.float32)); w = T((rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32))
    bias = T(rng.normal(0, 0.2, cout).astype(np.float32))
    # reference on the fp16-ROUNDED operands in fp32: isolates the kernel (fp32 accumulate) from the input rounding
    xh, wh = x.half().float(), w.half().float()
    want = F.relu(F.conv2d(xh, wh, bias, stride=stride, padding=(k - 1) // 2)).numpy()
    xr = torch.cat((x[:, rot:], x[:, :rot]), 1) if rot else x
    wp, bp = ops.pack_conv_f16(w.to(dev), None, bias.to(dev), rot)
    got = ops.c8_to_nchw(ops.conv2d_c8(ops.nchw_to_c8(xr.to(dev)), wp, bp, cout, k, stride, True), cout).cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1.5e-3 * scale, (np.abs(got - want).max(), scale)     # output rounding to fp16: 2^-11 relative
    return got


@pytest.mark.parametrize("cin,cout,N,H,W", [(128, 64, 2, 24, 40), (256, 128, 1, 13, 21), (64, 64, 3, 8, 8)])
def test_conv3x3_upsampled_c8_vs_torch(dev, cin, cout, N, H, W):
    """fp16 fused up_conv (upsample x2 -> conv3x3 -> folded BN -> ReLU on the low-resolution input, composed phase filters +
    ring pass) against torch on the fp16-rounded input: interior, ring, corners, odd sizes, pixel tails of the tiles."""
    from cnmnet_amd import ops
    rng = np.random.default_rng(cin + H)
    x = T(rng.standard_normal((N, cin, H, W)).astype(np.float32)).half().float()
    w = T((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32))
    bn = tuple(T(a.astype(np.float32)) for a in (rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.2, cout), rng.normal(0, 0.2, cout), rng.uniform(0.5, 1.5, cout)))
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    want = F.relu(F.batch_norm(F.conv2d(up, w, None, padding=1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)).numpy()
    wp, bp, wr = ops.pack_upsampled_f16(w.to(dev), tuple(t.to(dev) for t in bn))
    got = ops.c8_to_nchw(ops.conv3x3_upsampled_c8(ops.nchw_to_c8(x.to(dev)), wp, bp, cout, True, wr), cout).cpu().numpy()
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert err.max() < 6e-3 * scale, (err.max(), scale, np.unravel_index(err.argmax(), err.shape))    # fp16 composed filters + fp16 output
    ring = np.ones(want.shape[2:], bool); ring[1:-1, 1:-1] = False
    assert err[:, :, ring].max() < 6e-3 * scale                                                        # the ring pass (zero padding of the upsampled image)


@pytest.mark.parametrize("N,C,H,W", [(2, 24, 12, 20), (1, 8, 1, 1), (1, 16, 5, 7), (3, 40, 6, 1), (1, 8, 1, 9)])
def test_upsample2x_c8_vs_torch(dev, N, C, H, W):
    """fp16 bilinear x2 (one thread per low-resolution pixel) against F.interpolate on the fp16-rounded input: odd sizes,
    single rows / columns (the clamped first and last neighbours)."""
    from cnmnet_amd import ops
    rng = np.random.default_rng(N * 100 + H * 10 + W)
    x = T(rng.standard_normal((N, C, H, W)).astype(np.float32)).half().float()
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False).numpy()
    got = ops.c8_to_nchw(ops.upsample2x_c8(ops.nchw_to_c8(x.to(dev))), C).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-3 * max(1.0, np.abs(want).max())     # fp16 rounding of the output


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 192, 256), (8, 64, 100, 130), (2, 64, 24, 40), (1, 256, 12, 20), (9, 128, 96, 128), (3, 72, 181, 187)])
def test_head_sigmoid_c8_vs_torch(dev, N, C, H, W):
    """fp16 disparity head (3x3 conv to one channel + bias + scaled sigmoid) against torch on the fp16-rounded input: the
    row-walking kernel of the high-resolution levels [r5] (tiles of 62 x 8 outputs, ragged in both directions), the 4-slice and the
    16-slice kernel, ragged pixel tails; plus the nearest x2 copy into a channel group of a wider tensor.
    (Round 3 measured a port of the fp32 engine's row-walking head at 50 us against 45 us and dropped it; what it lacked was
    v_fma_mix_f32 -- the half -> float conversions were a third of its instructions, see cnmnet_amd/build.py.)"""
    from cnmnet_amd import ops
    rng = np.random.default_rng(C + H)
    x = T(rng.standard_normal((N, C, H, W)).astype(np.float32)).half().float()
    w = T((rng.standard_normal((1, C, 3, 3)) * 0.05).astype(np.float32)); b = T(np.array([0.1], np.float32))
    want = (3.0 * torch.sigmoid(F.conv2d(x, w, b, padding=1))).numpy()
    up = torch.full((N, 3, 2 * H, 2 * W, 8), 7.0, device=dev, dtype=torch.float16)
    got = ops.head_sigmoid_c8(ops.nchw_to_c8(x.to(dev)), ops.pack_head(w.to(dev)), b.to(dev), 3.0, up, 1)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-5 * 3.0
    upn = up.float().cpu().numpy()
    assert (upn[:, 0] == 7.0).all() and (upn[:, 2] == 7.0).all() and (upn[:, 1, :, :, 1:] == 0.0).all()
    near = np.repeat(np.repeat(want[:, 0], 2, axis=1), 2, axis=2)
    assert np.abs(upn[:, 1, :, :, 0] - near).max() < 2e-3 * 3.0                                        # fp16 rounding of the copy


def test_frame_f16_vs_fp32_oracle(dev, golden):
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    g, gr = golden("depthnet_64x96.npz"), golden("refine_64x96.npz")
    img, cams = syn.frames(2, 2, 64, 96, seed=int(g["seed"]))
    dn = _load(depthNet(3.0, 64, precision="f16"), int(g["weight_seed"])).to(dev)
    rn = _load(DepthRefineNet(32, 3.0, precision="f16"), int(gr["weight_seed"])).to(dev)
    L, lc = T(img[:, 0]).to(dev), T(cams[:, 0]).to(dev)
    with torch.no_grad():
        o1, f1 = dn(L, T(img[:, 1]).to(dev), lc, T(cams[:, 1]).to(dev))           # module-level API
        o2, f2 = dn(L, T(img[:, 2]).to(dev), lc, T(cams[:, 2]).to(dev))
        disp, prob = rn(idepth01=o1[0], idepth02=o2[0], iconv01=f1, iconv02=f2)
        out = FramePipeline(dn, rn, k_size=9)(T(img).to(dev), T(cams).to(dev))    # fused pipeline
    for i in range(4):
        assert np.abs(o1[i].cpu().numpy() - g["disp%d" % (i + 1)]).max() < 3e-2
    ch = list(g["iconv1_channels"])
    assert np.abs(f1[:, ch].cpu().numpy() - g["iconv1"]).max() < 5e-3 * np.abs(g["iconv1"]).max()
    assert np.abs(disp.cpu().numpy() - gr["disp_refined"]).max() < 6e-2 and np.abs(prob.cpu().numpy() - gr["prob_map"]).max() < 2e-2
    assert float((out["disp"] - disp).abs().max()) < 1e-6 and float((out["prob"] - prob).abs().max()) < 1e-6   # both routes agree
    assert torch.isfinite(out["normal"]).all()


@pytest.mark.parametrize("B,S,H,W,D", [(2, 2, 96, 160, 32), (1, 2, 224, 352, 64), (1, 2, 32, 32, 64), (3, 4, 32, 64, 32)])   # the last two: the smallest legal images
def test_f16_engine_other_sizes_vs_f32_engine(dev, B, S, H, W, D):
    """fp16 frame pipeline against the fp32 one at sizes whose pyramid levels have odd / non-power-of-two widths (352 ->
    176, 88, 44, 22, 11) and another plane count: the LDS-DMA kernel's pixel tails and tile choices."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    img, cams = syn.frames(B, S, H, W, seed=11)
    img, cams = T(img).to(dev), T(cams).to(dev)
    outs = {}
    for prec in ("f32", "f16"):
        pipe = FramePipeline(_load(depthNet(3.0, D, precision=prec), 1).to(dev), _load(DepthRefineNet(32, 3.0, precision=prec), 2).to(dev), k_size=9, normals=True)
        with torch.no_grad():
            outs[prec] = pipe(img, cams)
    d = (outs["f32"]["disp"] - outs["f16"]["disp"]).abs()
    assert float(d.max()) < 5e-2 and float(d.mean()) < 1e-3, (float(d.max()), float(d.mean()))      # [0, 3] range
    assert float((outs["f32"]["prob"] - outs["f16"]["prob"]).abs().max()) < 5e-2
    assert bool(torch.isfinite(outs["f16"]["normal"]).all())


@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_forward_pairs_reads_frame_slices_in_place(dev, prec):
    """[r6] depthNet.forward_pairs on SLICES of the caller's frame tensors (images[:, 0], images[:, 1:], cams likewise -- the reference's own call
    pattern, eval.py:440-447): the engine reads them where they lie (cnm_depthnet_forward_strided_*), bit-identical to contiguous copies; a
    view whose frames are not dense (a channel slice) still goes through a copy."""
    from cnmnet_amd.depthnet import depthNet
    from cnmnet_amd.depthnet.depthNet_model import _frame_view
    B, S, H, W = 3, 2, 64, 96
    img, cams = syn.frames(B, S, H, W, seed=23)
    img, cams = T(img).to(dev), T(cams).to(dev)
    net = _load(depthNet(3.0, 32, precision=prec), 4).to(dev)
    with torch.no_grad():
        a = net.forward_pairs(img[:, 0], img[:, 1:], cams[:, 0], cams[:, 1:])                                   # strided views
        b = net.forward_pairs(img[:, 0].contiguous(), img[:, 1:].contiguous(), cams[:, 0].contiguous(), cams[:, 1:].contiguous())
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    assert torch.equal(a[1], b[1])
    v, st = _frame_view(img[:, 0])
    assert v.data_ptr() == img.data_ptr() and st == (1 + S) * 3 * H * W                                         # no copy was made
    v, st = _frame_view(img[:, 1:])
    assert v.data_ptr() == img[:, 1:].data_ptr() and st == (1 + S) * 3 * H * W
    wide = torch.randn(B, 4, H, W, device=dev)
    v, st = _frame_view(wide[:, :3])                                                                             # frames dense: stride 4 H W
    assert v.data_ptr() == wide.data_ptr() and st == 4 * H * W
    v, st = _frame_view(wide[:, :3], 3 * H * W)                                                                  # ... but not whole images apart: copied for the sweep
    assert v.is_contiguous() and st == 3 * H * W
    v, st = _frame_view(wide[:, :, :, ::2])                                                                      # not dense inside a frame: copied
    assert v.is_contiguous() and st == 4 * H * (W // 2)


def test_f16_fused_upsample_networks_agree(dev):
    """fp16 nets with every eligible up_conv layer fused (threshold lowered to 1 pixel) against the same nets with the fused
    path off: the two differ only by fp16 roundings (composed filters instead of an upsampled tensor), and the fused path
    really runs."""
    from cnmnet_amd import _lib
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    lib = _lib.load()
    img, cams = syn.frames(2, 2, 64, 96, seed=31)
    outs = []
    old = lib.cnm_tune_upsampled_min_pixels_f16(1)
    try:
        for fused in (True, False):
            net = _load(depthNet(3.0, precision="f16"), 5).to(dev); net.fused_upsample = fused
            ref = _load(DepthRefineNet(32, 3.0, precision="f16"), 6).to(dev); ref.fused_upsample = fused
            with torch.no_grad():
                o1, f1 = net(T(img[:, 0]).to(dev), T(img[:, 1]).to(dev), T(cams[:, 0]).to(dev), T(cams[:, 1]).to(dev))
                o2, f2 = net(T(img[:, 0]).to(dev), T(img[:, 2]).to(dev), T(cams[:, 0]).to(dev), T(cams[:, 2]).to(dev))
                d, p = ref(o1[0], o2[0], f1, f2)
            outs.append([t.float().cpu().numpy() for t in (o1[0], o1[1], d, p)])
    finally:
        lib.cnm_tune_upsampled_min_pixels_f16(old)
    for a, b in zip(*outs):
        assert np.abs(a - b).max() < 4e-2 and np.abs(a - b).mean() < 3e-3, (np.abs(a - b).max(), np.abs(a - b).mean())   # [0, 3] range; two fp16 evaluation orders (measured 2.4e-2 / 1.2e-3)
    assert any(np.abs(a - b).max() > 0 for a, b in zip(*outs))           # the fused path really ran
