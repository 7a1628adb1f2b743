"""CPU: the oracle (oracle/ref_arrangement.py, oracle/closed_form.py) against the
golden fixtures produced by the imported reference (tests/golden/make_golden.py).
The reference ships no tests; these fixtures are the pin."""
import numpy as np
import pytest
import torch

from cnmnet_amd import synthetic as syn
from oracle import closed_form as cf
from oracle import ref_arrangement as ra
from conftest import torch_state

T = torch.from_numpy


def _weights(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True)))
    return module.eval()


def test_planesweep_torch_restatement_bitclose(golden):
    g = golden("planesweep_32x64.npz")
    vol = ra.plane_sweep_volume(T(g["left"]), T(g["right"]), T(g["left_cam"]), T(g["right_cam"]), 3.0, 64)
    assert g["frac_negative_t2"] > 0.05          # fixture really exercises behind-camera pixels
    np.testing.assert_allclose(vol.numpy(), g["volume"], rtol=0, atol=1e-6)


def test_planesweep_closed_form(golden):
    g = golden("planesweep_32x64.npz")
    vol = cf.plane_sweep_volume(g["left"], g["right"], g["left_cam"], g["right_cam"], 3.0, 64)
    # fp64 closed form vs the reference's fp32 op chain: rounding of u',v' (|u'| up to ~1e3 px
    # for the hard pair) times image gradient
    err = np.abs(vol - g["volume"])
    assert np.median(err) < 2e-5 and np.quantile(err, 0.999) < 2e-3 and err.max() < 5e-2


def test_planesweep_scale2_branch(golden):
    g, g2 = golden("planesweep_32x64.npz"), golden("planesweep_scale2_32x64.npz")
    vol = ra.plane_sweep_volume(T(g["left"][:1]), T(g["right"][:1]), T(g["left_cam"][:1]), T(g["right_cam"][:1]), 2.0, 64)
    np.testing.assert_allclose(vol.numpy(), g2["volume"], rtol=0, atol=1e-6)


def test_closed_form_plane_count_is_free(golden):
    """D=64 closed form == D=127 closed form on the shared planes (every second one)."""
    g = golden("planesweep_32x64.npz")
    a = cf.plane_sweep_volume(g["left"][:1], g["right"][:1], g["left_cam"][:1], g["right_cam"][:1], 3.0, 64)
    b = cf.plane_sweep_volume(g["left"][:1], g["right"][:1], g["left_cam"][:1], g["right_cam"][:1], 3.0, 127)
    np.testing.assert_allclose(a, b[:, ::2], atol=1e-9)


def test_depthnet_and_refine_match_golden(golden):
    g, gr = golden("depthnet_64x96.npz"), golden("refine_64x96.npz")
    img, cams = syn.frames(2, 2, 64, 96, seed=int(g["seed"]))
    assert abs(float(np.abs(img).sum()) - float(g["input_checksum"])) < 1e-3 * float(g["input_checksum"])
    np.testing.assert_array_equal(cams, g["cams"])
    net = _weights(ra.DepthNetCPU(3.0, 64), int(g["weight_seed"]))
    with torch.no_grad():
        outs, feat = net(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]))
        outs_b, feat_b = net(T(img[:, 0]), T(img[:, 2]), T(cams[:, 0]), T(cams[:, 2]))
    for i in range(4):
        np.testing.assert_allclose(outs[i].numpy(), g["disp%d" % (i + 1)], atol=2e-5)
    ch = list(g["iconv1_channels"])
    np.testing.assert_allclose(feat[:, ch].numpy(), g["iconv1"], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(feat_b[:, ch].numpy(), g["iconv1_b"], atol=2e-4, rtol=1e-5)
    ref = _weights(ra.DepthRefineNetCPU(32, 3.0), int(gr["weight_seed"]))
    with torch.no_grad():
        disp, prob, vf = ref(idepth01=outs[0], idepth02=outs_b[0], iconv01=feat, iconv02=feat_b, ReturnVolume=True)
    np.testing.assert_allclose(disp.numpy(), gr["disp_refined"], atol=2e-5)
    np.testing.assert_allclose(prob.numpy(), gr["prob_map"], atol=2e-5)
    np.testing.assert_allclose(vf[:, ch].numpy(), gr["iconv1_depth"], atol=5e-4, rtol=1e-5)


def test_state_dict_layout_matches_reference_counts():
    # SURVEY.md section 5 (checkpoint row): 128 entries for depthNet, 112 for the refine net
    d, r = ra.DepthNetCPU(3.0).state_dict(), ra.DepthRefineNetCPU().state_dict()
    assert len(d) == 128 and len(r) == 112
    assert tuple(d["conv1.0.weight"].shape) == (128, 67, 7, 7)
    assert tuple(d["iconv3.0.weight"].shape) == (256, 513, 3, 3)
    assert "disp1.0.bias" in d and "upconv5.1.weight" in d and "conv1.4.running_var" in d
    assert sum(v.numel() for k, v in d.items() if "running" not in k and "num_batches" not in k) == 33898500
    assert sum(v.numel() for k, v in r.items() if "running" not in k and "num_batches" not in k) == 10776066


@pytest.mark.parametrize("k", [9, 5])
def test_depth2normal(golden, k):
    g = golden("depth2normal_48x64.npz")
    n, p = ra.depth_to_normal(T(g["depth"]), T(g["K_inv"]), k)
    np.testing.assert_allclose(p.numpy(), g["points_k%d" % k], atol=1e-6)
    # the normal equations are ill-conditioned in fp32: thread count / bmm
    # blocking already moves the reference itself by ~1e-4
    np.testing.assert_allclose(n.numpy(), g["normal_k%d" % k], atol=2e-3)
    n64, p64, bad = cf.depth_to_normal(g["depth"], g["K_inv"], k)
    err = np.abs(n64 - g["normal_k%d" % k]).max(1)[~bad]
    assert np.quantile(err, 0.99) < 2e-3 and err.max() < 5e-2
    np.testing.assert_allclose(p64, g["points_k%d" % k], atol=1e-5)


def test_depth2normal_planar_known_answer():
    """A 3-D plane n.P = 1 has inverse depth linear in the pixel: 1/z = a + b x, and then
    the least-squares solution of depth_util.py:183-200 is exactly g = K^T (b,0,a)."""
    H, W, a, b = 48, 64, 0.5, -0.001
    xs = np.tile(np.arange(W, dtype=np.float64), (H, 1))
    depth = (1.0 / (a + b * xs))[None]
    K = syn.intrinsics(H, W)[:3, :3]
    n, _, bad = cf.depth_to_normal(depth, np.linalg.inv(K)[None], 9)
    g = K.T @ np.array([b, 0.0, a])
    want = g / (np.linalg.norm(g) + 1e-5)
    assert not bad.any()
    np.testing.assert_allclose(n[0], np.broadcast_to(want[:, None, None], n[0].shape), atol=1e-7)
    nt, _ = ra.depth_to_normal(T(depth.astype(np.float32)), T(np.linalg.inv(K)[None].astype(np.float32)), 9)
    np.testing.assert_allclose(nt[0].numpy(), np.broadcast_to(want[:, None, None], n[0].shape), atol=5e-3)


def test_inverse_warp(golden):
    g = golden("inverse_warp_32x64.npz")
    args = (T(g["depth"]), T(g["pose"]), T(g["K"]), T(g["K_inv"]))
    np.testing.assert_allclose(ra.inverse_warp(T(g["feat"]), *args).numpy(), g["warped_c3"], atol=1e-6)
    np.testing.assert_allclose(ra.inverse_warp(T(g["feat"][:, :1]), *args).numpy(), g["warped_c1"], atol=1e-6)
    w = cf.inverse_warp(g["feat"], g["depth"], g["pose"], g["K"], g["K_inv"])
    err = np.abs(w - g["warped_c3"])
    assert np.quantile(err, 0.999) < 1e-3 and np.median(err) < 1e-5


def test_inverse_warp_identity_pose_is_not_identity():
    """inverse_warp.py:69-70 normalises with (W-1) but samples with align_corners=False:
    reproduce as-is."""
    img, cams = syn.frames(1, 1, 16, 24, seed=1, smooth=False)
    K = cams[:, 0, 1, :3, :3]; Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    pose = np.eye(4, dtype=np.float32)[None, :3]
    w = cf.inverse_warp(img[:, 0], np.full((1, 16, 24), 2.0, np.float32), pose, K, Kinv)
    assert np.abs(w - img[:, 0]).max() > 0.5


def test_upsample_check_vector(golden):
    np.testing.assert_allclose(cf.upsample2x_bilinear(np.array([[0.0, 4.0, 8.0]]))[0], [0, 1, 3, 5, 7, 8])
    g = golden("upsample2x_5x7.npz")
    np.testing.assert_allclose(cf.upsample2x_bilinear(g["x"]), g["y"], atol=1e-6)


def test_plane_normals_oracle_vs_reference_golden(golden):
    """Plane branch of Depth2normal / get_normal_by_planes (depth_util.py:205-278), overlapping instances included."""
    import torch
    from oracle import ref_arrangement as ra
    g = golden("planes_24x32.npz")
    n, pts = ra.depth_to_normal(torch.from_numpy(g["depth"]), torch.from_numpy(g["K_inv"]), 9)
    assert np.abs(n.numpy() - g["normal_plain"]).max() < 1e-5
    reg, loss = ra.plane_normals(n, torch.from_numpy(g["seg"]), g["planes_num"])
    assert np.abs(reg.numpy() - g["normal_reg"]).max() < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    byp, none = ra.plane_normals(torch.from_numpy(g["normal_plain"]), torch.from_numpy(g["seg"]), g["planes_num"], with_loss=False)
    assert none is None and np.abs(byp.numpy() - g["normal_by_planes"]).max() < 1e-6
