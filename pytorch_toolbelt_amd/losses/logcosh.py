"""``LogCoshLoss`` module (reference losses/logcosh.py) over :func:`functional.log_cosh_loss`."""
import torch
from torch import nn

from .functional import log_cosh_loss

__all__ = ["LogCoshLoss"]


class LogCoshLoss(nn.Module):
    def forward(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        return log_cosh_loss(y_pred, y_true)
