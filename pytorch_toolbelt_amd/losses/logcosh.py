"""Module form of :func:`pytorch_toolbelt_amd.losses.functional.log_cosh_loss` (mean log-cosh of the residual, one
fused HIP pass)."""
from torch import Tensor, nn

from . import functional as LF

__all__ = ["LogCoshLoss"]


class LogCoshLoss(nn.Module):
    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        return LF.log_cosh_loss(y_pred, y_true)
