"""Jaccard (IoU) loss (drop-in for ``pytorch_toolbelt.losses.jaccard``) on the fused region-statistics kernel."""
from typing import List

from torch import Tensor
from torch.nn.modules.loss import _Loss

from . import _region as R

__all__ = ["JaccardLoss", "BINARY_MODE", "MULTICLASS_MODE", "MULTILABEL_MODE"]

BINARY_MODE = R.BINARY_MODE
MULTICLASS_MODE = R.MULTICLASS_MODE
MULTILABEL_MODE = R.MULTILABEL_MODE


class JaccardLoss(_Loss):
    """Soft Jaccard loss: score_c = (I_c + smooth) / max(P_c + T_c - I_c + smooth, eps).  No ``ignore_index`` (as in the
    reference); otherwise the same conventions as :class:`DiceLoss`."""

    def __init__(self, mode: str, classes: List[int] = None, log_loss=False, from_logits=True, smooth=0, eps=1e-7):
        assert mode in {BINARY_MODE, MULTILABEL_MODE, MULTICLASS_MODE}
        super().__init__()
        self.mode = mode
        self.classes = R.prepare_classes(mode, classes)
        self.from_logits = from_logits
        self.smooth = smooth
        self.eps = eps
        self.log_loss = log_loss

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        loss = R.fused_region_loss(y_pred, y_true, self.mode, self.from_logits, None, 0.0, 1.0, self.smooth, self.eps, self.log_loss,
                                   self.classes)
        if loss is not None:
            return loss
        inter, pred_mass, true_mass = R.region_statistics(y_pred, y_true, self.mode, self.from_logits, None)
        union = pred_mass + true_mass - inter
        scores = (inter + self.smooth) / (union + self.smooth).clamp_min(self.eps)
        return R.finish(scores, true_mass, self.log_loss, self.eps, self.classes)
