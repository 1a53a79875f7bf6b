"""Quality Focal Loss (https://arxiv.org/abs/2006.04388; reference losses/quality_focal_loss.py) as one fused HIP pass."""
import torch  # noqa: F401  (module attribute of the reference's file)
from torch import Tensor, nn

from . import _pointwise as P

__all__ = ["QualityFocalLoss"]


class QualityFocalLoss(nn.Module):
    """``|sigmoid(x) - t|^beta * BCE(x, t)`` evaluated in float32.  reduction: "mean" | "sum" | "normalized" (sum of
    the losses divided by the sum of the focal terms) | anything else -> unreduced."""

    __constants__ = ["beta", "reduction"]

    def __init__(self, beta: float = 2, reduction="mean"):
        super().__init__()
        self.beta = beta
        self.reduction = reduction

    def forward(self, predictions: Tensor, targets: Tensor) -> Tensor:
        if not predictions.is_cuda:
            from . import _host as H

            return H.quality_focal(predictions, targets, self.beta, self.reduction)
        x = P.as_f32(predictions, "QualityFocalLoss")
        t = P.as_f32(targets.detach(), "QualityFocalLoss")
        reduce = self.reduction in ("mean", "sum", "normalized")
        sums, elem = P.PointwiseSums.apply(x, t, None, None, P.QFL, 0, float(self.beta), 0.0, 0.0, 0.0, 1, 1, not reduce)
        if self.reduction == "mean":
            return (sums[0] / max(x.numel(), 1)).float()
        if self.reduction == "sum":
            return sums[0].float()
        if self.reduction == "normalized":
            return (sums[0] / sums[1]).float()
        return elem.view(predictions.shape)
