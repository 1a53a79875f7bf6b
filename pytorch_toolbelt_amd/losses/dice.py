"""Dice loss (drop-in for ``pytorch_toolbelt.losses.dice``) on the fused region-statistics kernel."""
from typing import List

from torch import Tensor
from torch.nn.modules.loss import _Loss

from . import _region as R

__all__ = ["DiceLoss"]

BINARY_MODE = R.BINARY_MODE
MULTICLASS_MODE = R.MULTICLASS_MODE
MULTILABEL_MODE = R.MULTILABEL_MODE


class DiceLoss(_Loss):
    """Soft Dice loss for binary, multiclass (label targets) and multilabel segmentation.

    score_c = (2 I_c + smooth) / max(P_c + T_c + smooth, eps); loss = mean_c [T_c > 0] * (1 - score_c) (or -log score).
    ``ignore_index`` masks both prediction and target.  ``classes`` restricts the mean to a subset of channels.
    """

    def __init__(self, mode: str, classes: List[int] = None, log_loss=False, from_logits=True, smooth: float = 0.0,
                 ignore_index=None, eps=1e-7):
        assert mode in {BINARY_MODE, MULTILABEL_MODE, MULTICLASS_MODE}
        super().__init__()
        self.mode = mode
        self.classes = R.prepare_classes(mode, classes)
        self.from_logits = from_logits
        self.smooth = smooth
        self.eps = eps
        self.ignore_index = ignore_index
        self.log_loss = log_loss

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        loss = R.fused_region_loss(y_pred, y_true, self.mode, self.from_logits, self.ignore_index, 1.0, 0.0, self.smooth, self.eps,
                                   self.log_loss, self.classes)
        if loss is not None:
            return loss
        inter, pred_mass, true_mass = R.region_statistics(y_pred, y_true, self.mode, self.from_logits, self.ignore_index)
        scores = (2.0 * inter + self.smooth) / (pred_mass + true_mass + self.smooth).clamp_min(self.eps)
        return R.finish(scores, true_mass, self.log_loss, self.eps, self.classes)
