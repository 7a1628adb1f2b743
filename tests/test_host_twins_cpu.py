"""CPU: the host twins of the C ABI (csrc/host_twins.cpp, EngHost in csrc/nets.hip) against the golden fixtures written by the
imported reference -- the product's own C++ on host pointers, called through ctypes exactly like the HIP entry points; oracle/
appears only as the checker (closed forms).  SURVEY 8(b) "_cpu twins", BASELINE configs[0] (plumbing on the CPU)."""
import numpy as np
import pytest
import torch

from cnmnet_amd import host, ops, synthetic as syn
from conftest import torch_state
from oracle import closed_form as cf
from oracle import ref_arrangement as ra

T = torch.from_numpy


def _load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True)))
    return module.eval()


def _max(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_geometry_and_plane_sweep_golden(golden):
    g = golden("planesweep_32x64.npz")
    hmkt = ops.homography_terms(T(g["left_cam"]), T(g["right_cam"]).unsqueeze(1)).numpy()
    Hm, KT = cf.homography_terms(g["left_cam"], g["right_cam"])
    np.testing.assert_allclose(hmkt[:, :9].reshape(-1, 3, 3), Hm, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(hmkt[:, 9:], KT, rtol=2e-6, atol=1e-6)
    vol = ops.plane_sweep_volume(T(g["left"]), T(g["right"]), T(g["left_cam"]), T(g["right_cam"]), 3.0, 64).numpy()
    assert np.isfinite(vol).all()
    assert _max(vol[0], g["volume"][0]) < 2e-4 and _max(vol[1], g["volume"][1]) < 1e-5      # the GPU test's bars
    # the concatenated conv input: planes in groups of four, then (r, g, b, 0)                 depthNet_model.py:233
    x = host.plane_sweep_cat_c4(T(g["left"]), T(g["right"]).unsqueeze(1), T(hmkt), 0.1, 3.0, 64)
    assert torch.equal(x[:, :16].permute(0, 1, 4, 2, 3).reshape(2, 64, 32, 64), T(vol))
    assert torch.equal(x[:, 16, :, :, :3].permute(0, 3, 1, 2), T(g["left"])) and float(x[:, 16, :, :, 3].abs().max()) == 0.0
    # other plane counts against the float64 closed form (the reference cannot run them)
    v32 = ops.plane_sweep_volume(T(g["left"]), T(g["right"]), T(g["left_cam"]), T(g["right_cam"]), 3.0, 32).numpy()
    assert _max(v32[0], cf.plane_sweep_volume(g["left"], g["right"], g["left_cam"], g["right_cam"], 3.0, 32)[0]) < 3e-4


@pytest.mark.parametrize("H,W,D", [(24, 40, 128), (12, 20, 2), (9, 17, 5), (1, 1, 8)])
def test_plane_sweep_limits_on_the_host(H, W, D):
    """The ends of the plane sweep's range on the host twin, as tests/test_gpu_parity.py::test_planesweep_plane_count_and_size_limits
    has them for the HIP kernel: 128 and 2 planes, a count that is no multiple of 4, a one-pixel image -- vs the closed form."""
    img, cams = syn.frames(2, 1, max(H, 8), max(W, 8), seed=500 + D)
    img = np.ascontiguousarray(img[..., :H, :W])
    vol = ops.plane_sweep_volume(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]), 3.0, D).numpy()
    want = cf.plane_sweep_volume(img[:, 0], img[:, 1], cams[:, 0], cams[:, 1], 3.0, D)
    assert vol.shape == (2, D, H, W) and _max(vol, want) < 1e-3
    from cnmnet_amd import _lib
    for bad in (1, 129):
        with pytest.raises(_lib.EngineError):
            ops.plane_sweep_volume(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]), 3.0, bad)


@pytest.mark.parametrize("cin,cin2,cout,k,stride,rot,N,H,W", [(7, 0, 8, 3, 1, 0, 2, 9, 13), (67, 0, 16, 7, 1, 3, 1, 12, 20), (12, 9, 20, 5, 2, 0, 2, 11, 14),
                                                               (16, 0, 8, 3, 2, 0, 1, 8, 8), (8, 5, 12, 7, 2, 0, 1, 10, 17)])
def test_conv_bn_relu_vs_torch(cin, cin2, cout, k, stride, rot, N, H, W):
    """cnm_conv2d_cat2_c4_cpu with the BatchNorm-folded filter of cnm_pack_conv_bn_cpu against torch's fp64 Conv2d + BatchNorm (eval)
    + ReLU: ragged channel groups, two input views, rotated first-layer channels, odd sizes, both strides."""
    rng = np.random.default_rng(cin + 10 * k + H)
    cp = 4 * ((cin + 3) // 4) if cin2 else cin
    x = T(rng.standard_normal((N, cin, H, W)).astype(np.float32))
    x2 = T(rng.standard_normal((N, cin2, H, W)).astype(np.float32)) if cin2 else None
    ct = cp + cin2
    w = T((rng.standard_normal((cout, ct, k, k)) * (2.0 / (ct * k * k)) ** 0.5).astype(np.float32))
    bn = [T(a.astype(np.float32)) for a in (rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.2, cout), rng.normal(0, 0.2, cout), rng.uniform(0.5, 1.5, cout))]
    xin = x if not cin2 else torch.cat([x, torch.zeros(N, cp - cin, H, W), x2], 1)
    sc = bn[0].double() / torch.sqrt(bn[3].double() + 1e-5)
    want = torch.relu(torch.nn.functional.conv2d(xin.double(), w.double(), stride=stride, padding=k // 2) * sc[None, :, None, None]
                      + (bn[1].double() - bn[2].double() * sc)[None, :, None, None]).numpy()
    wp, bp = host.pack_conv(w, bn, rot=rot)
    xr = torch.cat((x[:, rot:], x[:, :rot]), 1) if rot else x
    got = ops.c4_to_nchw(host.conv2d_c4(ops.nchw_to_c4(xr), wp, bp, cout, k, stride, True, x2=ops.nchw_to_c4(x2) if cin2 else None), cout).numpy()
    assert got.shape == want.shape and _max(got, want) < 2e-5 * max(np.abs(want).max(), 1.0)


def test_depthnet_and_refine_golden_on_the_host(golden):
    """Both nets through cnm_depthnet_forward_cpu / cnm_refinenet_forward_cpu against the reference's golden outputs at 64x96 (the bars
    of tests/test_gpu_parity.py::test_depthnet_and_refine_golden), and the multi-source entry against the two-view one."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    g, gr = golden("depthnet_64x96.npz"), golden("refine_64x96.npz")
    img, cams = syn.frames(2, 2, 64, 96, seed=int(g["seed"]))
    net = _load(depthNet(3.0), int(g["weight_seed"]))
    L, lc = T(img[:, 0]), T(cams[:, 0])
    with torch.no_grad():
        outs, feat = net(L, T(img[:, 1]), lc, T(cams[:, 1]))
        outs_b, feat_b = net(L, T(img[:, 2]), lc, T(cams[:, 2]))
    for i in range(4):
        assert _max(outs[i].numpy(), g["disp%d" % (i + 1)]) < 1e-3
    ch = list(g["iconv1_channels"])
    assert _max(feat[:, ch].numpy(), g["iconv1"]) < 1e-4 * np.abs(g["iconv1"]).max()
    assert _max(outs_b[0].numpy(), g["disp1_b"]) < 1e-3
    ref = _load(DepthRefineNet(32, 3.0), int(gr["weight_seed"]))
    with torch.no_grad():
        disp, prob, vf = ref(idepth01=outs[0], idepth02=outs_b[0], iconv01=feat, iconv02=feat_b, ReturnVolume=True)
        disp2, prob2 = ref(outs[0].clone(), outs_b[0].clone(), feat.clone(), feat_b.clone())     # NCHW -> c4 conversion path
    assert _max(disp.numpy(), gr["disp_refined"]) < 1e-3 and _max(prob.numpy(), gr["prob_map"]) < 1e-3
    assert _max(vf[:, ch].numpy(), gr["iconv1_depth"]) < 1e-4 * np.abs(gr["iconv1_depth"]).max()
    assert torch.equal(disp, disp2) and torch.equal(prob, prob2)
    # S = 2 through the multi-source entry = the two-view call                                eval.py:635-663
    pairs_d = torch.stack((outs[0], outs_b[0]), 1).reshape(4, 1, 64, 96)
    pairs_f = torch.stack((feat._cnm_c4, feat_b._cnm_c4), 1).reshape(4, 16, 64, 96, 4)
    dm, pm, _ = ref.forward_multi(pairs_d, pairs_f, 2)
    assert torch.equal(dm, disp) and torch.equal(pm, prob)


def test_frame_pipeline_and_errors_on_the_host():
    """FramePipeline with CPU modules and tensors (depthNet -> refine -> normals, all host twins); training and f16 on the CPU fail loudly."""
    from cnmnet_amd import _lib
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    img, cams = syn.frames(1, 2, 32, 64, seed=5)
    dn, rn = _load(depthNet(3.0, 32), 11), _load(DepthRefineNet(32, 3.0), 12)                  # 32 planes: BASELINE configs[0]'s count
    out = FramePipeline(dn, rn, k_size=9)(T(img), T(cams))
    assert out["disp"].shape == (1, 1, 32, 64) and out["normal"].shape == (1, 3, 32, 64)
    assert torch.isfinite(out["disp"]).all() and float(out["disp"].min()) > 0 and float(out["disp"].max()) < 3.0
    n = out["normal"].norm(dim=1)
    assert bool((((n - 1).abs() < 1e-4) | (n < 1e-6)).all())
    with pytest.raises(ValueError):
        dn(T(img[:, 0, :, :, :40]), T(img[:, 1, :, :, :40]), T(cams[:, 0]), T(cams[:, 1]))     # W = 40
    with pytest.raises(_lib.EngineError):
        dn.train()(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]))                   # no host training
    with pytest.raises(_lib.EngineError):
        depthNet(3.0, precision="f16").eval()(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]))


@pytest.mark.parametrize("k", [9, 5])
def test_depth2normal_golden_on_the_host(golden, k):
    from cnmnet_amd.depthnet import Depth2normal
    g = golden("depth2normal_48x64.npz")
    n, p = Depth2normal(k)(T(g["depth"]), T(g["K_inv"]))
    n, p = n.numpy(), p.numpy()
    np.testing.assert_allclose(p, g["points_k%d" % k], atol=2e-6, rtol=1e-6)
    n64, _, bad = cf.depth_to_normal(g["depth"], g["K_inv"], k)
    good = ~bad
    assert np.abs(n - n64).max(1)[good].max() < 2e-5                                         # fp64 window sums: the exact answer of the fp32 point cloud
    err_ref = np.abs(g["normal_k%d" % k] - n64).max(1)[good]
    err_us = np.abs(n - g["normal_k%d" % k]).max(1)[good]
    assert np.quantile(err_us, 0.99) < 1e-3 and err_us.max() <= err_ref.max() + 2e-5
    np.testing.assert_allclose(np.moveaxis(n, 1, -1)[bad], np.moveaxis(n64, 1, -1)[bad], atol=2e-5)


def test_inverse_warp_golden_on_the_host(golden):
    from cnmnet_amd.depthnet import inverse_warp
    g = golden("inverse_warp_32x64.npz")
    a = [T(g[k]) for k in ("depth", "pose", "K", "K_inv")]
    for feat, want in ((g["feat"], g["warped_c3"]), (g["feat"][:, :1], g["warped_c1"])):
        err = np.abs(inverse_warp(T(np.ascontiguousarray(feat)), *a).numpy().astype(np.float64) - want.astype(np.float64))
        # continuous except on the +-1 boundary of the reference's out-of-view test (inverse_warp.py:71-75): a handful of pixels there
        assert float(np.median(err)) < 1e-6 and int((err > 1e-4).sum()) <= 24, (float(np.median(err)), int((err > 1e-4).sum()))


def test_config0_depthnet_on_the_host_vs_oracle():
    """BASELINE configs[0] literally: "DepthNet eval, 1 ref + 1 src, 256x192, 32 depth planes, batch=1 on CPU" -- through the product's host
    twins, against the CPU oracle with the same weights (the reference itself cannot run 32 planes: depthNet_model.py:194,199,208 hard-code 64;
    the oracle restatement is proven equal to it at 64).  Inverse depth at all four scales within the north-star's 1e-3."""
    from cnmnet_amd.depthnet import depthNet
    img, cams = syn.frames(1, 2, 192, 256, seed=77)
    net, ref = _load(depthNet(3.0, 32), 21), _load(ra.DepthNetCPU(3.0, 32), 21)
    a = (T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]))
    with torch.no_grad():
        outs, feat = net(*a)
        want, wfeat = ref(*a)
    for o, w in zip(outs, want):
        assert o.shape == w.shape and _max(o.numpy(), w.numpy()) < 1e-3
    assert _max(feat.numpy(), wfeat.numpy()) < 1e-4 * float(wfeat.abs().max())


@pytest.mark.parametrize("S", [4, 6])
def test_multi_source_frame_on_the_host_vs_oracle(S):
    """a-8 on the host: 4- and 6-source fusion (eval.py:635-663, :885-929) through FramePipeline with CPU modules."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    img, cams = syn.frames(1, S, 32, 64, seed=90 + S)
    pipe = FramePipeline(_load(depthNet(3.0), 31), _load(DepthRefineNet(32, 3.0), 32), normals=False)
    out = pipe(T(img), T(cams))
    disp, prob = ra.frame_forward_multi(_load(ra.DepthNetCPU(3.0), 31), _load(ra.DepthRefineNetCPU(32, 3.0), 32), T(img[0]), T(cams[0]))
    assert _max(out["disp"].numpy(), disp.numpy()) < 1e-3 and _max(out["prob"].numpy(), prob.numpy()) < 1e-3
