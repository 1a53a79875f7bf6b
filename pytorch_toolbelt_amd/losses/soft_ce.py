"""``SoftCrossEntropyLoss`` (reference losses/soft_ce.py): cross entropy with label smoothing, one fused HIP pass over
the logits (log-softmax statistics, the target gather and the smoothing term are computed per pixel in registers)."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import _pointwise as P

__all__ = ["SoftCrossEntropyLoss"]


class SoftCrossEntropyLoss(nn.Module):
    """``label_smoothed_nll_loss(log_softmax(input, dim), target, smooth_factor, ignore_index, reduction, dim)``
    (soft_ce.py:24-33, functional.py:280-323): ``(1 - eps) * nll + eps / C * smooth`` with ``smooth = -sum_c log p_c``;
    ignored pixels contribute 0 and "mean" divides by ALL pixels.  Unreduced output keeps the class dim as size 1 when
    ``ignore_index`` is not None and drops it otherwise, like the reference."""

    __constants__ = ["reduction", "ignore_index", "smooth_factor"]

    def __init__(self, reduction: str = "mean", smooth_factor: float = 0.0, ignore_index: Optional[int] = -100, dim=1):
        super().__init__()
        self.smooth_factor = smooth_factor
        self.ignore_index = ignore_index
        self.reduction = reduction
        self.dim = dim

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if not input.is_cuda:      # host tensors: the reference's own composition (soft_ce.py:24-33)
            from .functional import label_smoothed_nll_loss

            return label_smoothed_nll_loss(torch.log_softmax(input, dim=self.dim), target, epsilon=self.smooth_factor, ignore_index=self.ignore_index,
                                           reduction=self.reduction, dim=self.dim)
        dim = self.dim % input.dim()
        x = input if dim == 1 else input.movedim(dim, 1)
        if target.dim() == input.dim():
            target = target.squeeze(dim)
        lead = x.shape[0]
        C = x.shape[1]
        rest = tuple(x.shape[2:])
        x3 = P.as_f32(x, "SoftCrossEntropyLoss").reshape(lead, C, -1)
        labels = target.to(device=x3.device, dtype=torch.int64).reshape(lead, -1).contiguous()
        if labels.shape[1] != x3.shape[2]:
            raise ValueError(f"target of shape {tuple(target.shape)} does not match input of shape {tuple(input.shape)}")
        reduce = self.reduction in ("mean", "sum")
        has_ignore = self.ignore_index is not None
        total, pix = P.SoftCESums.apply(x3, labels, float(self.smooth_factor), has_ignore, int(self.ignore_index) if has_ignore else 0,
                                        not reduce)
        out_dtype = input.dtype if input.dtype.is_floating_point else torch.float32
        if self.reduction == "mean":
            return (total / max(labels.numel(), 1)).to(out_dtype)
        if self.reduction == "sum":
            return total.to(out_dtype)
        out = pix.view((lead,) + rest)
        if has_ignore:   # functional.py:299-307: the gathered dim is kept in this branch
            out = out.unsqueeze(1)
            if dim != 1:
                out = out.movedim(1, dim)
        return out.to(out_dtype)
