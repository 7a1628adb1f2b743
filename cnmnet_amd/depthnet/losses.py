"""Training losses with the reference's names and semantics (reference depthnet/losses.py:7-122).

SURVEY.md section 8 (row a-9) keeps these in PyTorch: they are boolean-mask gathers followed by
means -- no arithmetic worth a kernel -- and they need autograd.  They are device-agnostic torch code
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


class IdepthLoss(nn.Module):
    """Masked mean L1 (optionally on log10) at full resolution (losses.py:30-48)."""

    def forward(self, idepth_pred, idepth_groud_truth, log=False):
        m = _valid(idepth_pred, idepth_groud_truth)
        p, g = idepth_pred[m], idepth_groud_truth[m]
        return F.l1_loss(torch.log10(p), torch.log10(g)) if log else F.l1_loss(p, g)


class IdepthwithProbLoss(nn.Module):
    """Probability-weighted masked L1, plain mean over the masked elements (losses.py:51-73)."""

    def forward(self, idepth_pred, idepth_gt, prob_map, log=False):
        m = _valid(idepth_pred, idepth_gt)
        p, g, w = idepth_pred[m], idepth_gt[m], prob_map[m]
        diff = 10 * (torch.log10(p) - torch.log10(g)).abs() if log else (p - g).abs()
        return (w * diff).mean()


def surface_normal_loss(prediction, surface_normal, valid_region, probability_map=None):
    """[B,3,h,w] normals, valid_region [B,1,h,w] bool -> (loss, mean angular error in degrees)
    (losses.py:76-122)."""
    finite = torch.isfinite(surface_normal.sum(1, keepdim=True)) & torch.isfinite(prediction.sum(1, keepdim=True))
    keep = (finite & valid_region).squeeze(1)                                             # [B,h,w]
    p = prediction.permute(0, 2, 3, 1)[keep]                                              # [n,3]
    g = surface_normal.permute(0, 2, 3, 1)[keep]
    sim = F.cosine_similarity(p, g, dim=1)
    if probability_map is None:
        loss = (1 - sim).mean()
    else:
        w = probability_map.squeeze(1)[keep]
        loss = ((1 - sim) * w).sum() / w.sum()
    angle = torch.acos(sim.clamp(-1, 1)).mean()
    return loss, angle / math.pi * 180
