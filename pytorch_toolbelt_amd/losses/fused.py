"""Fused focal + Dice + Jaccard loss (extension, BASELINE.json configs[3]): one HIP pass over logits and labels feeds
all three losses.  Values are identical to ``w_f * BinaryFocalLoss(...) + w_d * DiceLoss(...) + w_j * JaccardLoss(...)``
of this package (and therefore, within 1e-5, to the reference's three modules)."""
from typing import Optional

import torch
from torch import Tensor, nn

from . import _kernels as K
from . import _region as R

__all__ = ["FocalDiceJaccardLoss"]


class FocalDiceJaccardLoss(nn.Module):
    """``focal_weight * BinaryFocalLoss + dice_weight * DiceLoss(mode) + jaccard_weight * JaccardLoss(mode)``.

    mode "multiclass": ``y_true`` are class indices [B, *] (Dice/Jaccard use softmax probabilities, focal the sigmoid
    of the same logits with an on-the-fly one-hot); "multilabel"/"binary": ``y_true`` is a dense 0/1 map.
    """

    def __init__(self, mode: str = R.MULTICLASS_MODE, focal_weight: float = 1.0, dice_weight: float = 1.0, jaccard_weight: float = 1.0,
                 alpha: Optional[float] = None, gamma: float = 2.0, smooth: float = 0.0, eps: float = 1e-7, log_loss: bool = False,
                 ignore_index: Optional[int] = None):
        super().__init__()
        assert mode in {R.BINARY_MODE, R.MULTILABEL_MODE, R.MULTICLASS_MODE}
        self.mode, self.alpha, self.gamma, self.smooth, self.eps = mode, alpha, gamma, smooth, eps
        self.log_loss, self.ignore_index = log_loss, ignore_index
        self.weights = (focal_weight, dice_weight, jaccard_weight)

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        wf, wd, wj = self.weights
        loss = R.fused_region_loss(y_pred, y_true, self.mode, True, self.ignore_index, wd, wj, self.smooth, self.eps, self.log_loss, None,
                                   focal=dict(weight=wf, gamma=self.gamma, alpha=self.alpha))
        if loss is not None:
            return loss
        if not y_pred.is_cuda:      # host tensors: the three losses composed from their host evaluations
            from .functional import _sigmoid_focal

            labels, dense = (y_true, None) if self.mode == R.MULTICLASS_MODE else (None, y_true.reshape(y_pred.shape))
            focal_loss = _sigmoid_focal(y_pred, labels, dense, self.gamma, self.alpha, "mean", False, None, 1e-6, self.ignore_index, None)
            inter, pred_mass, true_mass = R.region_statistics(y_pred, y_true, self.mode, True, self.ignore_index)
            dice = (2.0 * inter + self.smooth) / (pred_mass + true_mass + self.smooth).clamp_min(self.eps)
            jacc = (inter + self.smooth) / (pred_mass + true_mass - inter + self.smooth).clamp_min(self.eps)
            return (wf * focal_loss + wd * R.finish(dice, true_mass, self.log_loss, self.eps, None)
                    + wj * R.finish(jacc, true_mass, self.log_loss, self.eps, None))
        bs = y_pred.size(0)
        x = K._f32c(y_pred, "fused loss")
        flags = (K.SEG_HAS_ALPHA if self.alpha is not None else 0) | (K.SEG_HAS_IGNORE if self.ignore_index is not None else 0) | K.SEG_NO_TERM
        ign = self.ignore_index
        if self.mode == R.MULTICLASS_MODE:
            x = x.reshape(bs, x.size(1), -1)
            labels = y_true.to(device=x.device, dtype=torch.int64).reshape(bs, -1).contiguous()
            focal, stats = K.FusedSegSums.apply(x, labels, None, None, flags, K.PROB_SOFTMAX, float(self.gamma), float(self.alpha or 0.0), 0.0,
                                                int(ign) if ign is not None else 0, 0.0)
        else:
            C = 1 if self.mode == R.BINARY_MODE else x.size(1)
            x = x.reshape(bs, C, -1)
            dense = K._f32c(y_true.to(device=x.device), "fused loss").reshape(bs, C, -1)
            focal, stats = K.FusedSegSums.apply(x, None, dense, None, flags, K.PROB_SIGMOID, float(self.gamma), float(self.alpha or 0.0), 0.0,
                                                0, float(ign) if ign is not None else 0.0)
        from ..parallel import sync_region_statistics

        inter, pred_mass, true_mass = sync_region_statistics.apply((stats[0].float(), stats[1].float(), stats[2].float()))
        wf, wd, wj = self.weights
        focal_loss = (focal[0] / x.numel()).float()
        dice = (2.0 * inter + self.smooth) / (pred_mass + true_mass + self.smooth).clamp_min(self.eps)
        jacc = (inter + self.smooth) / (pred_mass + true_mass - inter + self.smooth).clamp_min(self.eps)
        return wf * focal_loss + wd * R.finish(dice, true_mass, self.log_loss, self.eps, None) + wj * R.finish(jacc, true_mass, self.log_loss, self.eps, None)
