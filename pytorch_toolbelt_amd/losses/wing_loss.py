"""Module form of the wing loss (landmark regression, https://arxiv.org/abs/1711.06753); the math is the fused HIP pass of
:func:`pytorch_toolbelt_amd.losses.functional.wing_loss`."""
from torch.nn.modules.loss import _Loss

from .functional import wing_loss

__all__ = ["WingLoss"]


class WingLoss(_Loss):
    """``width * log(1 + |d| / curvature)`` below ``width``, linear beyond; ``reduction``: "mean" | "sum" | anything else
    for the unreduced map."""

    def __init__(self, width=5, curvature=0.5, reduction="mean"):
        super().__init__(reduction=reduction)
        self.width, self.curvature = width, curvature

    def forward(self, prediction, target):
        return wing_loss(prediction, target, width=self.width, curvature=self.curvature, reduction=self.reduction)
