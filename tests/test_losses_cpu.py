"""CPU: cnmnet_amd.depthnet.losses (a-9) against the imported reference's losses on tensors with
NaN / zero / negative holes (skipped where /root/reference is absent) and closed-form values."""
import numpy as np
import pytest
import torch

from cnmnet_amd.depthnet import losses as L
from oracle import import_reference as ir


def _data(seed=0):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(2, 1, 16, 24, generator=g) * 2 + 0.1
    pred = gt + 0.1 * torch.randn(2, 1, 16, 24, generator=g)
    gt[0, 0, 2:5, 3:9] = 0.0; gt[1, 0, 7, 7] = float("nan"); pred[0, 0, 10, 10] = float("inf"); pred[1, 0, 0, :4] = -0.2
    prob = torch.rand(2, 1, 16, 24, generator=g)
    preds = [pred, F_half(pred, 2), F_half(pred, 4), F_half(pred, 8)]
    return pred, gt, prob, preds


def F_half(x, k):
    return torch.nn.functional.avg_pool2d(torch.nan_to_num(x, posinf=1.0), k)


def test_closed_form_values():
    pred = torch.tensor([[[[1.0, 2.0], [3.0, -1.0]]]]); gt = torch.tensor([[[[1.5, 0.0], [2.0, 1.0]]]])
    assert abs(float(L.IdepthLoss()(pred, gt)) - 0.75) < 1e-7                 # elements (0,0) and (1,0): |-.5|, |1|
    prob = torch.tensor([[[[0.5, 1.0], [0.25, 1.0]]]])
    assert abs(float(L.IdepthwithProbLoss()(pred, gt, prob)) - (0.5 * 0.5 + 0.25 * 1.0) / 2) < 1e-7
    n = torch.tensor([0.0, 0.0, 1.0]).view(1, 3, 1, 1).expand(1, 3, 2, 2).clone()
    m = torch.tensor([0.0, 1.0, 0.0]).view(1, 3, 1, 1).expand(1, 3, 2, 2).clone()
    loss, ang = L.surface_normal_loss(n, m, torch.ones(1, 1, 2, 2, dtype=torch.bool))
    assert abs(float(loss) - 1.0) < 1e-6 and abs(float(ang) - 90.0) < 1e-4
    assert torch.isnan(L.IdepthLoss()(pred, torch.zeros_like(gt)))            # empty mask -> NaN, as the reference


@pytest.mark.skipif(not ir.available(), reason="reference checkout not present")
def test_against_live_reference_with_gradients():
    R = ir.load().losses
    pred, gt, prob, preds = _data()
    for log in (False, True):
        a = L.IdepthLoss()(pred, gt, log); b = R.IdepthLoss()(pred, gt, log)
        assert torch.allclose(a, b, atol=1e-7)
        a = L.IdepthwithProbLoss()(pred, gt, prob, log); b = R.IdepthwithProbLoss()(pred, gt, prob, log)
        assert torch.allclose(a, b, atol=1e-7)
    assert torch.allclose(L.IdepthLoss_234()(preds, gt.nan_to_num(1.0)), R.IdepthLoss_234()(preds, gt.nan_to_num(1.0)), atol=1e-7)
    g = torch.Generator().manual_seed(3)
    n1 = torch.nn.functional.normalize(torch.randn(1, 3, 16, 24, generator=g), dim=1).requires_grad_(True)
    n2 = torch.nn.functional.normalize(torch.randn(1, 3, 16, 24, generator=g), dim=1); n2[0, :, 3, 4] = float("nan")
    valid = torch.rand(1, 1, 16, 24, generator=g) > 0.3
    for pm in (None, prob[:1]):
        la, aa = L.surface_normal_loss(n1, n2, valid, pm); lb, ab = R.surface_normal_loss(n1, n2, valid, pm)
        assert torch.allclose(la, lb, atol=1e-6) and torch.allclose(aa, ab, atol=1e-4)
        ga, = torch.autograd.grad(la, n1, retain_graph=True); gb, = torch.autograd.grad(lb, n1)
        assert torch.allclose(ga, gb, atol=1e-7)
