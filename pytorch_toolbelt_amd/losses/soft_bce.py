"""``SoftBCEWithLogitsLoss`` (reference losses/soft_bce.py): BCE-with-logits with label smoothing and ``ignore_index``,
evaluated as one fused HIP pass (forward) / one pass (backward)."""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import _pointwise as P

__all__ = ["SoftBCEWithLogitsLoss"]


def _per_channel(w: Optional[Tensor], shape) -> Optional[Tensor]:
    """``w`` as a [C] vector if it broadcasts against ``shape`` along dim 1 only (or is a scalar); else None."""
    if w is None:
        return None
    C = int(shape[1]) if len(shape) > 1 else 1
    if w.numel() == 1:
        return w.reshape(1).expand(C)
    if w.dim() > len(shape):
        return None
    padded = (1,) * (len(shape) - w.dim()) + tuple(w.shape)
    if len(shape) > 1 and padded[1] == C and all(s == 1 for i, s in enumerate(padded) if i != 1):
        return w.reshape(C)
    return None


class SoftBCEWithLogitsLoss(nn.Module):
    """Drop-in for ``nn.BCEWithLogitsLoss`` plus ``ignore_index`` (elements whose target equals it contribute 0) and
    ``smooth_factor`` (targets become ``(1 - t) * s + t * (1 - s)``).  reduction: "mean" (over all elements) | "sum" |
    anything else -> unreduced."""

    __constants__ = ["weight", "pos_weight", "reduction", "ignore_index", "smooth_factor"]

    def __init__(self, weight=None, ignore_index: Optional[int] = -100, reduction="mean", smooth_factor=None, pos_weight=None):
        super().__init__()
        self.ignore_index = ignore_index
        self.reduction = reduction
        self.smooth_factor = smooth_factor
        self.register_buffer("weight", weight)
        self.register_buffer("pos_weight", pos_weight)

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        cw = _per_channel(self.weight, input.shape)
        cpw = _per_channel(self.pos_weight, input.shape)
        native = (input.is_cuda and input.shape == target.shape and not target.requires_grad
                  and (self.weight is None or cw is not None) and (self.pos_weight is None or cpw is not None))
        if not native:
            return self._composite(input, target)
        x = P.as_f32(input, "SoftBCEWithLogitsLoss")
        t = P.as_f32(target.detach(), "SoftBCEWithLogitsLoss")
        C = int(input.shape[1]) if input.dim() > 1 else 1
        HW = 1
        for s in input.shape[2:]:
            HW *= int(s)
        flags = (P.F_IGNORE if self.ignore_index is not None else 0) | (P.F_SMOOTH if self.smooth_factor is not None else 0)
        cw = cw.to(device=x.device, dtype=torch.float32).contiguous() if cw is not None else None
        cpw = cpw.to(device=x.device, dtype=torch.float32).contiguous() if cpw is not None else None
        reduce = self.reduction in ("mean", "sum")
        sums, elem = P.PointwiseSums.apply(x, t, cw, cpw, P.SOFT_BCE, flags, float(self.smooth_factor or 0.0), 0.0, 0.0,
                                           float(self.ignore_index if self.ignore_index is not None else 0), C, HW, not reduce)
        if self.reduction == "mean":
            return (sums[0] / max(x.numel(), 1)).to(input.dtype if input.dtype.is_floating_point else torch.float32)
        if self.reduction == "sum":
            return sums[0].to(input.dtype if input.dtype.is_floating_point else torch.float32)
        return elem.view(input.shape).to(input.dtype)

    def _composite(self, input, target):
        # general broadcast weights / differentiable targets: torch's own BCE (rare configurations, not a tuned path)
        if self.smooth_factor is not None:
            soft = ((1 - target) * self.smooth_factor + target * (1 - self.smooth_factor)).type_as(input)
        else:
            soft = target.type_as(input)
        loss = F.binary_cross_entropy_with_logits(input, soft, self.weight, pos_weight=self.pos_weight, reduction="none")
        if self.ignore_index is not None:
            loss = loss * (target != self.ignore_index).type_as(loss)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss
