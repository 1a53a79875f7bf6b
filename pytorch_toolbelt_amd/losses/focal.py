"""Focal losses (drop-in for ``pytorch_toolbelt.losses.focal``) on the fused HIP loss kernel."""
from typing import Optional

import torch
from torch import Tensor, nn

from ..utils.support import pytorch_toolbelt_deprecated
from .functional import _sigmoid_focal, _softmax_act_focal, softmax_focal_loss_with_logits

__all__ = ["CrossEntropyFocalLoss", "BinaryFocalLoss", "FocalLoss"]


class BinaryFocalLoss(nn.Module):
    """Sigmoid focal loss for binary / multi-label problems.

    ``inputs`` are logits ``[B, C, *]``; ``targets`` are either a same-shaped 0/1 (or soft) map, or a label map
    ``[B, *]`` of class indices, which the reference expands with an int64 ``one_hot`` (1 GiB at [32,16,512,512],
    reference losses/focal.py:88-105).  The HIP kernel forms the one-hot on the fly from the labels instead.
    Pixels whose label equals ``ignore_index`` are ignored in every channel.
    NB: the module default is ``alpha=None`` while the functional's default is 0.25 (as in the reference).
    """

    __constants__ = ["alpha", "gamma", "reduction", "ignore_index", "normalized", "reduced_threshold", "activation"]

    def __init__(self, alpha: Optional[float] = None, gamma: float = 2.0, ignore_index: Optional[int] = None, reduction: str = "mean",
                 normalized: bool = False, reduced_threshold: Optional[float] = None, activation: str = "sigmoid",
                 softmax_dim: Optional[int] = None, class_weights: Optional[Tensor] = None):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma
        self.ignore_index = ignore_index
        self.reduction = reduction
        self.normalized = normalized
        self.reduced_threshold = reduced_threshold
        self.activation = activation
        self.softmax_dim = softmax_dim
        if class_weights is not None and not torch.is_tensor(class_weights):
            class_weights = torch.tensor(list(class_weights), dtype=torch.float32)
        self.register_buffer("class_weights", class_weights, persistent=False)

    def __repr__(self):
        cw = None if self.class_weights is None else self.class_weights.tolist()  # (the reference crashes on None here)
        return (
            f"{self.__class__.__name__}(alpha={self.alpha}, gamma={self.gamma}, ignore_index={self.ignore_index}, "
            f"reduction={self.reduction}, normalized={self.normalized}, reduced_threshold={self.reduced_threshold}, "
            f"activation={self.activation}, softmax_dim={self.softmax_dim},class_weights={cw}, )"
        )

    def forward(self, inputs: Tensor, targets: Tensor) -> Tensor:
        labels, dense = (targets, None) if targets.dim() + 1 == inputs.dim() else (None, targets)
        if self.activation != "sigmoid":   # "softmax" (functional.py:61-64)
            return _softmax_act_focal(inputs, labels, dense, self.softmax_dim, self.gamma, self.alpha, self.reduction, self.normalized,
                                      self.reduced_threshold, 1e-6, self.ignore_index, self.class_weights)
        return _sigmoid_focal(inputs, labels, dense, self.gamma, self.alpha, self.reduction, self.normalized,
                              self.reduced_threshold, 1e-6, self.ignore_index, self.class_weights)


class CrossEntropyFocalLoss(nn.Module):
    """Multi-class focal loss whose focal term comes from the softmax probabilities; targets are class indices
    ``[B, *]`` like ``nn.CrossEntropyLoss``."""

    def __init__(self, gamma: float = 2.0, reduction: str = "mean", normalized: bool = False, reduced_threshold: Optional[float] = None,
                 ignore_index: int = -100, class_weights: Optional[Tensor] = None):
        super().__init__()
        self.gamma = gamma
        self.reduction = reduction
        self.reduced_threshold = reduced_threshold
        self.normalized = normalized
        self.ignore_index = ignore_index
        self.register_buffer("class_weights", class_weights, persistent=False)

    def forward(self, inputs: Tensor, targets: Tensor) -> Tensor:
        return softmax_focal_loss_with_logits(
            inputs, targets, gamma=self.gamma, reduction=self.reduction, normalized=self.normalized,
            reduced_threshold=self.reduced_threshold, ignore_index=self.ignore_index, class_weights=self.class_weights)


@pytorch_toolbelt_deprecated("Class FocalLoss is deprecated. Please use CrossEntropyFocalLoss instead.")
def FocalLoss(*input, **kwargs):
    return CrossEntropyFocalLoss(*input, **kwargs)
