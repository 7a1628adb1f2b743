"""CPU, authoring container only: oracle/ref_arrangement.py against the LIVE imported
reference on fresh seeds/shapes (skipped where /root/reference is absent, e.g. the GPU box)."""
import numpy as np
import pytest
import torch

from cnmnet_amd import synthetic as syn
from oracle import import_reference as ir
from oracle import ref_arrangement as ra
from conftest import torch_state

pytestmark = pytest.mark.skipif(not ir.available(), reason="reference checkout not present")
T = torch.from_numpy


def _pair(ref_mod, my_mod, seed):
    shapes = {k: tuple(v.shape) for k, v in ref_mod.state_dict().items()}
    assert list(shapes) and set(shapes) == set(my_mod.state_dict())
    w = torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True))
    ref_mod.load_state_dict(w); my_mod.load_state_dict(w)
    return ref_mod.eval(), my_mod.eval()


@torch.no_grad()
def test_full_frame_matches_live_reference():
    ns = ir.load()
    img, cams = syn.frames(1, 2, 32, 64, seed=77)
    rd, md = _pair(ns.depthNet(3.0), ra.DepthNetCPU(3.0), 5)
    rr, mr = _pair(ns.DepthRefineNet(32, 3.0), ra.DepthRefineNetCPU(32, 3.0), 6)
    L, lc = T(img[:, 0]), T(cams[:, 0])
    o1, f1 = rd(L, T(img[:, 1]), lc, T(cams[:, 1])); o2, f2 = rd(L, T(img[:, 2]), lc, T(cams[:, 2]))
    disp, prob = rr(idepth01=o1[0], idepth02=o2[0], iconv01=f1, iconv02=f2)
    out = ra.frame_forward(md, mr, L, T(img[:, 1]), T(img[:, 2]), lc, T(cams[:, 1]), T(cams[:, 2]), k_size=9)
    np.testing.assert_allclose(out["disp"].numpy(), disp.numpy(), atol=1e-6)
    np.testing.assert_allclose(out["prob"].numpy(), prob.numpy(), atol=1e-6)
    n, p = ns.Depth2normal(9)(1.0 / disp.squeeze(1), lc[:, 1, :3, :3].inverse())
    np.testing.assert_allclose(out["points"].numpy(), p.numpy(), atol=1e-6)
    assert np.quantile(np.abs(out["normal"].numpy() - n.numpy()), 0.999) < 2e-3


@torch.no_grad()
def test_train_mode_forward_backward_matches_live_reference():
    ns = ir.load()
    img, cams = syn.frames(2, 1, 32, 32, seed=78)
    rd, md = _pair(ns.depthNet(3.0), ra.DepthNetCPU(3.0), 8)
    rd.train(); md.train()
    grads = []
    with torch.enable_grad():
        for net in (rd, md):
            outs, feat = net(T(img[:, 0]), T(img[:, 1]), T(cams[:, 0]), T(cams[:, 1]))
            (outs[0].mean() + feat.mean()).backward()
            grads.append((net.conv1[0].weight.grad.clone(), net.disp1[0].bias.grad.clone()))
    np.testing.assert_allclose(grads[0][0].numpy(), grads[1][0].numpy(), atol=1e-7)
    np.testing.assert_allclose(grads[0][1].numpy(), grads[1][1].numpy(), atol=1e-7)
