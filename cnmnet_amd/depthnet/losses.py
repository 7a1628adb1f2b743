"""Training losses with the reference's names and semantics (reference depthnet/losses.py:7-122).

SURVEY.md section 8 (row a-9) keeps these in PyTorch: they are masked means -- no arithmetic worth a kernel -- and
they need autograd.  The reference gathers `x[mask]` (dynamic shapes: one device-to-host synchronisation per term); here
the masked-out elements are zeroed and sums are divided by the mask count -- same values, same gradients, static shapes.  They are device-agnostic torch code
(the engine's no-CPU rule concerns the HIP operators, not these reductions).
Semantics spelled out:
  mask = gt > 0 & isfinite(gt) & isfinite(pred) & pred > 0; L1 = mean over masked elements (NaN if empty);
  multi-scale ground truth = nearest F.interpolate, NO mask, weight 0.1/3;
  normal loss = mean (or prob-weighted mean) of 1 - cos over pixels where valid_region holds and both
  normal maps are finite, plus the mean angular error in degrees.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _valid(pred, gt):
    return (gt > 0.0) & torch.isfinite(gt) & torch.isfinite(pred) & (pred > 0.0)       # losses.py:39-40, :61


class IdepthLoss_234(nn.Module):
    """0.1/3 * sum of unmasked mean-L1 of disp2..4 against nearest-resized ground truth (losses.py:7-27)."""

    def forward(self, idepth_preds, idepth_ground_truth):
        total = 0.0
        for disp in idepth_preds[1:4]:
            gt = F.interpolate(idepth_ground_truth, size=disp.shape[2:4])                 # nearest (:18-20)
            total = total + (disp - gt).abs().mean()
        return 0.1 * total / 3.0


def _masked(m, *tensors):
    """Masked-out elements replaced by a neutral value BEFORE any arithmetic: a non-finite ground truth never reaches the
    loss or its gradient, and every shape stays static (no `x[mask]` gather, which costs a device-to-host synchronisation)."""
    return [torch.where(m, t, torch.zeros((), dtype=t.dtype, device=t.device)) for t in tensors]


class IdepthLoss(nn.Module):
    """Masked mean L1 (optionally on log10) at full resolution (losses.py:30-48): sum over the mask / mask count
    (NaN for an empty mask, like the reference's mean of an empty selection)."""

    def forward(self, idepth_pred, idepth_groud_truth, log=False):
        m = _valid(idepth_pred, idepth_groud_truth)
        n = m.sum().to(idepth_pred.dtype)
        if log:
            one = torch.ones((), dtype=idepth_pred.dtype, device=idepth_pred.device)
            p, g = torch.where(m, idepth_pred, one), torch.where(m, idepth_groud_truth, one)
            return (torch.log10(p) - torch.log10(g)).abs().sum() / n
        p, g = _masked(m, idepth_pred, idepth_groud_truth)
        return (p - g).abs().sum() / n


class IdepthwithProbLoss(nn.Module):
    """Probability-weighted masked L1, plain mean over the masked elements (losses.py:51-73)."""

    def forward(self, idepth_pred, idepth_gt, prob_map, log=False):
        m = _valid(idepth_pred, idepth_gt)
        n = m.sum().to(idepth_pred.dtype)
        (w,) = _masked(m, prob_map)
        if log:
            one = torch.ones((), dtype=idepth_pred.dtype, device=idepth_pred.device)
            diff = 10 * (torch.log10(torch.where(m, idepth_pred, one)) - torch.log10(torch.where(m, idepth_gt, one))).abs()
        else:
            p, g = _masked(m, idepth_pred, idepth_gt)
            diff = (p - g).abs()
        return (w * diff).sum() / n


def surface_normal_loss(prediction, surface_normal, valid_region, probability_map=None):
    """[B,3,h,w] normals, valid_region [B,1,h,w] bool -> (loss, mean angular error in degrees)
    (losses.py:76-122), in static shapes: pixels outside the mask are zero vectors (cosine 0) with zero weight."""
    finite = torch.isfinite(surface_normal.sum(1, keepdim=True)) & torch.isfinite(prediction.sum(1, keepdim=True))
    keep = finite & valid_region                                                          # [B,1,h,w]
    p, g = _masked(keep, prediction, surface_normal)
    sim = F.cosine_similarity(p, g, dim=1)                                                # [B,h,w]
    k = keep.squeeze(1).to(sim.dtype)
    n = k.sum()
    if probability_map is None:
        loss = ((1 - sim) * k).sum() / n
    else:
        w = _masked(keep, probability_map)[0].squeeze(1)
        loss = ((1 - sim) * w).sum() / w.sum()
    angle = (torch.acos(sim.clamp(-1, 1)) * k).sum() / n
    return loss, angle / math.pi * 180
