"""Class-balanced BCE (reference losses/balanced_bce.py; https://arxiv.org/pdf/1504.06375.pdf, formula 2).

The class counts and both log-sigmoid sums come out of ONE fused HIP pass; the balance weights are then scalar algebra
on the device (no host synchronisation)."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import _pointwise as P

__all__ = ["BalancedBCEWithLogitsLoss", "balanced_binary_cross_entropy_with_logits"]


def balanced_binary_cross_entropy_with_logits(logits: Tensor, targets: Tensor, gamma: float = 1.0, ignore_index: Optional[int] = None,
                                               reduction: str = "mean") -> Tensor:
    """``-(w_pos * t * logsigmoid(x) + w_neg * (1 - t) * logsigmoid(-x))`` with ``w_pos = (n_neg / n)^gamma`` raised to
    ``gamma`` once more and ``w_neg = (1 - (n_neg / n)^gamma)^gamma`` (the reference applies the power twice,
    balanced_bce.py:30-34), ``n_pos / n_neg`` = number of targets equal to 1 / 0.  Targets are expected to be hard 0/1;
    elements equal to ``ignore_index`` contribute 0.  "mean" | "sum" | otherwise unreduced."""
    if not logits.is_cuda:
        from . import _host as H

        return H.balanced_bce(logits, targets, gamma, ignore_index, reduction)
    x = P.as_f32(logits, "balanced_binary_cross_entropy_with_logits")
    t = P.as_f32(targets.detach(), "balanced_binary_cross_entropy_with_logits")
    if x.shape != t.shape:
        t = t.expand_as(x).contiguous()
    flags = P.F_IGNORE if ignore_index is not None else 0
    ign = float(ignore_index if ignore_index is not None else 0)
    sums, _ = P.PointwiseSums.apply(x, t, None, None, P.BALANCED_BCE, flags, 0.0, 0.0, 0.0, ign, 1, 1, False)
    n_pos, n_neg = sums[2].detach(), sums[3].detach()
    pos_weight = torch.pow((n_neg / (n_pos + n_neg + 1e-7)).float(), gamma)       # :30-31 (float32 like the reference)
    neg_weight = 1.0 - pos_weight
    w_pos, w_neg = pos_weight.pow(gamma), neg_weight.pow(gamma)                     # :33-34
    out_dtype = logits.dtype if logits.dtype.is_floating_point else torch.float32
    if reduction in ("mean", "sum"):
        total = -(w_pos.double() * sums[0] + w_neg.double() * sums[1])
        if reduction == "mean":
            total = total / max(x.numel(), 1)
        return total.to(out_dtype)
    loss = P.BalancedElementwise.apply(x, t, torch.stack([w_pos, w_neg]), flags, ign)
    return loss.view(logits.shape).to(out_dtype)


class BalancedBCEWithLogitsLoss(nn.Module):
    """Module form of :func:`balanced_binary_cross_entropy_with_logits`."""

    __constants__ = ["gamma", "reduction", "ignore_index"]

    def __init__(self, gamma: float = 1.0, reduction="mean", ignore_index: Optional[int] = None):
        super().__init__()
        self.gamma = gamma
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, output: Tensor, target: Tensor) -> Tensor:
        return balanced_binary_cross_entropy_with_logits(output, target, gamma=self.gamma, ignore_index=self.ignore_index,
                                                         reduction=self.reduction)
