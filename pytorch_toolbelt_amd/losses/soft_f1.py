"""Soft (differentiable) F1 losses with the reference's names (``pytorch_toolbelt/losses/soft_f1.py``).

The soft counts TP = sum p t, FP = sum p (1 - t), FN = sum (1 - p) t over the samples are region statistics: TP = I,
FP = P - I, FN = T - I, so ``F1 = 2 I / (P + T + eps)``.  On GPU tensors they come from ONE fused HIP pass each:

* ``BinarySoftF1Loss`` (any shape, e.g. segmentation maps): sigmoid + clamp + ignore mask + the three sums in
  ``ptb_pointwise_loss_fwd`` kind 5 (csrc/ptb_pointwise.hip); backward = one read + one write;
* ``soft_micro_f1`` / ``SoftF1Loss`` ([N, C] probabilities / logits): the per-class region-statistics kernels of the Dice /
  Jaccard losses on the class-major view.

CPU tensors (classification-sized unit tests) take the same formulas as torch algebra."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import _kernels as K
from . import _pointwise as P

__all__ = ["soft_micro_f1", "BinarySoftF1Loss", "SoftF1Loss"]


def _f1_from_counts(inter, pred_mass, true_mass, eps):
    return (1 - 2 * inter / (pred_mass + true_mass + eps)).mean()      # 2 TP / (2 TP + FN + FP + eps), soft_f1.py:25-27


def soft_micro_f1(preds: Tensor, targets: Tensor, eps=1e-6) -> Tensor:
    """Mean over classes of ``1 - F1`` with soft counts taken over dim 0 of ``[N, C]`` probabilities / targets:
    ``F1_c = 2 TP_c / (2 TP_c + FN_c + FP_c + eps)``, ``TP = sum p t``, ``FP = sum p (1 - t)``, ``FN = sum (1 - p) t``."""
    if preds.is_cuda and preds.dim() == 2 and targets.shape == preds.shape and preds.numel():
        n, c = preds.shape
        x = K._f32c(preds, "soft_micro_f1")
        t = K._f32c(targets.to(preds.device), "soft_micro_f1")
        if c == 1:
            sums, _ = P.PointwiseSums.apply(x.reshape(-1), t.reshape(-1), None, None, P.SOFT_F1, P.F_SMOOTH, 0.0, 0.0, 0.0, 0.0, 1, 1, False)
            return _f1_from_counts(sums[0:1], sums[1:2], sums[2:3], eps).to(preds.dtype)
        # class-major view [1, C, N]: the statistics kernels stream N contiguously per class
        stats = K.RegionStats.apply(x.t().contiguous().unsqueeze(0), None, t.t().contiguous().unsqueeze(0), K.PROB_IDENTITY, False, 0, 0.0)
        return _f1_from_counts(stats[0], stats[1], stats[2], eps).to(preds.dtype)
    hits = (preds * targets).sum(0)
    false_alarms = (preds * (1 - targets)).sum(0)
    misses = ((1 - preds) * targets).sum(0)
    f1 = 2 * hits / (2 * hits + misses + false_alarms + eps)
    return (1 - f1).mean()


def _drop_ignored(preds: Tensor, targets: Tensor, ignore_index):
    """Entries whose target equals ``ignore_index`` are removed; returns None when nothing is left."""
    if ignore_index is None:
        return preds, targets
    keep = targets != ignore_index
    preds, targets = preds[keep], targets[keep]
    return None if targets.numel() == 0 else (preds, targets)


class BinarySoftF1Loss(nn.Module):
    """Soft F1 of ``sigmoid(logits)`` (clamped to ``[eps, 1 - eps]``) against flat 0/1 targets."""

    def __init__(self, ignore_index: Optional[int] = None, eps=1e-6):
        super().__init__()
        self.ignore_index = ignore_index
        self.eps = eps

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        if preds.is_cuda and preds.numel():
            x = K._f32c(preds, "BinarySoftF1Loss").reshape(-1)
            t = K._f32c(targets.to(preds.device), "BinarySoftF1Loss").reshape(-1)
            if t.numel() != x.numel():
                raise RuntimeError(f"target shape {tuple(targets.shape)} does not match prediction shape {tuple(preds.shape)}")
            ign = self.ignore_index is not None
            sums, _ = P.PointwiseSums.apply(x, t, None, None, P.SOFT_F1, P.F_IGNORE if ign else 0, float(self.eps), 0.0, 0.0,
                                            float(self.ignore_index) if ign else 0.0, 1, 1, False)
            loss = _f1_from_counts(sums[0:1], sums[1:2], sums[2:3], 1e-6)   # (soft_micro_f1's own default eps, soft_f1.py:78)
            if ign:   # everything ignored -> 0 (soft_f1.py:73-74), decided on the device: no host synchronisation
                loss = torch.where(sums[3] > 0, loss, torch.zeros_like(loss))
            return loss.to(preds.dtype)
        kept = _drop_ignored(preds.view(-1), targets.view(-1), self.ignore_index)
        if kept is None:
            return torch.tensor(0, dtype=preds.dtype, device=preds.device)
        logits, flat_t = kept
        probs = logits.sigmoid().clamp(self.eps, 1 - self.eps)
        return soft_micro_f1(probs.view(-1, 1), flat_t.view(-1, 1))


class SoftF1Loss(nn.Module):
    """Soft F1 of ``softmax(logits, 1)`` (clamped) against one-hot encoded integer targets."""

    def __init__(self, ignore_index: Optional[int] = None, eps=1e-6):
        super().__init__()
        self.ignore_index = ignore_index
        self.eps = eps

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        if preds.is_cuda and preds.dim() == 2 and targets.dim() == 1 and self.ignore_index is None and preds.numel() and self.eps <= 1e-6:
            # softmax + one-hot + the three per-class sums in one pass over the class-major view [1, C, N]; the clamp to
            # [eps, 1 - eps] (1e-6 by default) moves a probability by at most eps -- far inside the 1e-5 parity tolerance; a caller's
            # larger eps takes the op chain below, which clamps for real
            x = K._f32c(preds, "SoftF1Loss").t().contiguous().unsqueeze(0)
            labels = targets.to(device=preds.device, dtype=torch.int64).reshape(1, -1).contiguous()
            if labels.shape[1] != x.shape[2]:
                raise RuntimeError(f"target shape {tuple(targets.shape)} does not match prediction shape {tuple(preds.shape)}")
            stats = K.RegionStats.apply(x, labels, None, K.PROB_SOFTMAX, False, 0, 0.0)
            return _f1_from_counts(stats[0], stats[1], stats[2], 1e-6).to(preds.dtype)
        probs = preds.softmax(dim=1).clamp(self.eps, 1 - self.eps)
        onehot = torch.nn.functional.one_hot(targets, probs.size(1))
        kept = _drop_ignored(probs, onehot, self.ignore_index)
        if kept is None:
            return torch.tensor(0, dtype=probs.dtype, device=probs.device)
        return soft_micro_f1(*kept)
