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


def total(x):
    """Sum of all elements (integer for a bool mask) as rows of 256, then the row sums, until at most 1024 values are left.
    Why not x.sum(): torch splits a reduction to few outputs over several workgroups that meet through a scratch block cleared
    by cudaMemsetAsync; replayed inside a HIP graph (TrainStep(graph=True)) such reductions returned stale values after a
    device-wide synchronise between replays (tools/graph_sync_probe.py).  One workgroup per output needs
    no scratch.  A size that is not a multiple of 256 is zero-padded up to one first."""
    x = x.reshape(-1)
    if x.dtype == torch.bool:
        x = x.to(torch.int32)
    while x.numel() > 1024:
        if x.numel() % 256:
            x = F.pad(x, (0, -x.numel() % 256))
        x = x.view(-1, 256).sum(1)
    return x.sum()


def row_totals(x):
    """[B, ...] -> [B]: per-sample sums, staged as `total` is."""
    x = x.flatten(1)
    while x.shape[1] > 1024:
        if x.shape[1] % 256:
            x = F.pad(x, (0, -x.shape[1] % 256))
        x = x.view(x.shape[0], -1, 256).sum(2)
    return x.sum(1)


def mean_all(x):
    return total(x) / x.numel()


def _valid(pred, gt):
    return (gt > 0.0) & torch.isfinite(gt) & torch.isfinite(pred) & (pred > 0.0)       # losses.py:39-40, :61


class IdepthLoss_234(nn.Module):
    """0.1/3 * sum of unmasked mean-L1 of disp2..4 against nearest-resized ground truth (losses.py:7-27)."""

    def forward(self, idepth_preds, idepth_ground_truth):
        acc = 0.0
        for disp in idepth_preds[1:4]:
            gt = F.interpolate(idepth_ground_truth, size=disp.shape[2:4])                 # nearest (:18-20)
            acc = acc + mean_all((disp - gt).abs())
        return 0.1 * acc / 3.0


def _masked(m, *tensors):
    """Masked-out elements replaced by a neutral value BEFORE any arithmetic: a non-finite ground truth never reaches the
    loss or its gradient, and every shape stays static (no `x[mask]` gather, which costs a device-to-host synchronisation)."""
    return [torch.where(m, t, torch.zeros((), dtype=t.dtype, device=t.device)) for t in tensors]


class IdepthLoss(nn.Module):
    """Masked mean L1 (optionally on log10) at full resolution (losses.py:30-48): sum over the mask / mask count
    (NaN for an empty mask, like the reference's mean of an empty selection)."""

    def forward(self, idepth_pred, idepth_groud_truth, log=False):
        m = _valid(idepth_pred, idepth_groud_truth)
        n = total(m).to(idepth_pred.dtype)
        if log:
            one = torch.ones((), dtype=idepth_pred.dtype, device=idepth_pred.device)
            p, g = torch.where(m, idepth_pred, one), torch.where(m, idepth_groud_truth, one)
            return total((torch.log10(p) - torch.log10(g)).abs()) / n
        p, g = _masked(m, idepth_pred, idepth_groud_truth)
        return total((p - g).abs()) / n


class IdepthwithProbLoss(nn.Module):
    """Probability-weighted masked L1, plain mean over the masked elements (losses.py:51-73)."""

    def forward(self, idepth_pred, idepth_gt, prob_map, log=False):
        m = _valid(idepth_pred, idepth_gt)
        n = total(m).to(idepth_pred.dtype)
        (w,) = _masked(m, prob_map)
        if log:
            one = torch.ones((), dtype=idepth_pred.dtype, device=idepth_pred.device)
            diff = 10 * (torch.log10(torch.where(m, idepth_pred, one)) - torch.log10(torch.where(m, idepth_gt, one))).abs()
        else:
            p, g = _masked(m, idepth_pred, idepth_gt)
            diff = (p - g).abs()
        return total(w * diff) / n


def surface_normal_loss(prediction, surface_normal, valid_region, probability_map=None):
    """[B,3,h,w] normals, valid_region [B,1,h,w] bool -> (loss, mean angular error in degrees)
    (losses.py:76-122), in static shapes: pixels outside the mask are zero vectors (cosine 0) with zero weight."""
    finite = torch.isfinite(surface_normal.sum(1, keepdim=True)) & torch.isfinite(prediction.sum(1, keepdim=True))
    keep = finite & valid_region                                                          # [B,1,h,w]
    p, g = _masked(keep, prediction, surface_normal)
    sim = F.cosine_similarity(p, g, dim=1)                                                # [B,h,w]
    k = keep.squeeze(1).to(sim.dtype)
    n = total(k)
    if probability_map is None:
        loss = total((1 - sim) * k) / n
    else:
        w = _masked(keep, probability_map)[0].squeeze(1)
        loss = total((1 - sim) * w) / total(w)
    angle = total(torch.acos(sim.clamp(-1, 1)) * k) / n
    return loss, angle / math.pi * 180
