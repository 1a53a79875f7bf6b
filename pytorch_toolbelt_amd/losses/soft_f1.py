"""Soft (differentiable) F1 losses with the reference's names (``pytorch_toolbelt/losses/soft_f1.py``).
Classification-shaped ``[samples, classes]`` inputs: plain torch tensor algebra (runs on the MI355X through ATen); not
part of the tiled-inference hot path."""
from typing import Optional

import torch
from torch import Tensor, nn

__all__ = ["soft_micro_f1", "BinarySoftF1Loss", "SoftF1Loss"]


def soft_micro_f1(preds: Tensor, targets: Tensor, eps=1e-6) -> Tensor:
    """Mean over classes of ``1 - F1`` with soft counts taken over dim 0 of ``[N, C]`` probabilities / targets:
    ``F1_c = 2 TP_c / (2 TP_c + FN_c + FP_c + eps)``, ``TP = sum p t``, ``FP = sum p (1 - t)``, ``FN = sum (1 - p) t``."""
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
        probs = preds.softmax(dim=1).clamp(self.eps, 1 - self.eps)
        onehot = torch.nn.functional.one_hot(targets, probs.size(1))
        kept = _drop_ignored(probs, onehot, self.ignore_index)
        if kept is None:
            return torch.tensor(0, dtype=probs.dtype, device=probs.device)
        return soft_micro_f1(*kept)
