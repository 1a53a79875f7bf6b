"""Soft (differentiable) F1 losses (reference losses/soft_f1.py).  Classification-shaped ``[samples, classes]`` inputs:
plain torch tensor algebra (runs on the MI355X through ATen); not part of the tiled-inference hot path."""
from typing import Optional

import torch
from torch import Tensor, nn

__all__ = ["soft_micro_f1", "BinarySoftF1Loss", "SoftF1Loss"]


def soft_micro_f1(preds: Tensor, targets: Tensor, eps=1e-6) -> Tensor:
    """``mean_c (1 - 2 TP_c / (2 TP_c + FN_c + FP_c + eps))`` with soft counts summed over dim 0 of ``[N, C]``
    probabilities / targets."""
    tp = (preds * targets).sum(dim=0)
    fp = (preds * (1 - targets)).sum(dim=0)
    fn = ((1 - preds) * targets).sum(dim=0)
    return (1 - 2 * tp / (2 * tp + fn + fp + eps)).mean()


class BinarySoftF1Loss(nn.Module):
    def __init__(self, ignore_index: Optional[int] = None, eps=1e-6):
        super().__init__()
        self.ignore_index = ignore_index
        self.eps = eps

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        targets, preds = targets.view(-1), preds.view(-1)
        if self.ignore_index is not None:
            keep = targets != self.ignore_index
            preds, targets = preds[keep], targets[keep]
            if targets.numel() == 0:
                return torch.tensor(0, dtype=preds.dtype, device=preds.device)
        preds = preds.sigmoid().clamp(self.eps, 1 - self.eps)
        return soft_micro_f1(preds.view(-1, 1), targets.view(-1, 1))


class SoftF1Loss(nn.Module):
    def __init__(self, ignore_index: Optional[int] = None, eps=1e-6):
        super().__init__()
        self.ignore_index = ignore_index
        self.eps = eps

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        preds = preds.softmax(dim=1).clamp(self.eps, 1 - self.eps)
        targets = torch.nn.functional.one_hot(targets, preds.size(1))
        if self.ignore_index is not None:
            keep = targets != self.ignore_index
            preds, targets = preds[keep], targets[keep]
            if targets.numel() == 0:
                return torch.tensor(0, dtype=preds.dtype, device=preds.device)
        return soft_micro_f1(preds, targets)
