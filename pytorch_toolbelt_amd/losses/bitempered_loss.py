"""Bi-tempered logistic loss (Amid et al., https://arxiv.org/abs/1906.03361; reference losses/bitempered_loss.py).

The general form is classification-shaped (rows of ``num_classes`` activations with an iterative per-row normalisation): on the
GPU one fused HIP pass per direction with a wave per row (``ptb_bitempered_rows``); the torch tensor algebra below is what the
kernels restate (used for CPU tensors, soft labels that require a gradient and the degenerate t1 = 2).  The binary form on GPU
segmentation maps has its own per-pixel kernel (``BinaryBiTemperedLogisticLoss._native``).  The tempered
logarithm / exponential are ``log_t(u) = (u^(1-t) - 1) / (1 - t)`` and ``exp_t(u) = [1 + (1-t) u]_+^(1/(1-t))``.
"""
from typing import Optional

import torch
from torch import Tensor, nn

__all__ = ["BiTemperedLogisticLoss", "BinaryBiTemperedLogisticLoss"]


def log_t(u: Tensor, t: float) -> Tensor:
    return u.log() if t == 1.0 else (u.pow(1.0 - t) - 1.0) / (1.0 - t)


def exp_t(u: Tensor, t: float) -> Tensor:
    return u.exp() if t == 1 else (1.0 + (1.0 - t) * u).relu().pow(1.0 / (1.0 - t))


def _normalization_heavy_tail(a: Tensor, t: float, iters: int) -> Tensor:
    """t > 1: fixed-point iteration on the activations shifted by their row maximum."""
    top = a.max(dim=-1, keepdim=True).values
    shifted = a - top
    cur = shifted
    for _ in range(iters):
        z = exp_t(cur, t).sum(dim=-1, keepdim=True)
        cur = shifted * z.pow(1.0 - t)
    z = exp_t(cur, t).sum(dim=-1, keepdim=True)
    return top - log_t(1.0 / z, t)


def _normalization_finite_support(a: Tensor, t: float, iters: int) -> Tensor:
    """t < 1: bisection on the log_t-partition between 0 and -log_t(1 / effective_dim)."""
    top = a.max(dim=-1, keepdim=True).values
    shifted = a - top
    support = (shifted > -1.0 / (1.0 - t)).to(torch.int32).sum(dim=-1, keepdim=True).to(a.dtype)
    lo = torch.zeros_like(top)
    hi = -log_t(1.0 / support, t) * torch.ones_like(lo)
    for _ in range(iters):
        mid = (hi + lo) / 2.0
        mass = exp_t(shifted - mid, t).sum(dim=-1, keepdim=True)
        grow = (mass < 1.0).to(a.dtype)     # too little mass: the partition is smaller than mid
        lo = lo * grow + (1.0 - grow) * mid
        hi = hi * (1.0 - grow) + grow * mid
    return (hi + lo) / 2.0 + top


class _Normalization(torch.autograd.Function):
    """Row-wise normalisation constant with the closed-form backward (the escort distribution)."""

    @staticmethod
    def forward(ctx, activations, t, iters):
        fn = _normalization_finite_support if t < 1.0 else _normalization_heavy_tail
        const = fn(activations, t, iters)
        ctx.save_for_backward(activations, const)
        ctx.t = t
        return const

    @staticmethod
    def backward(ctx, g):
        activations, const = ctx.saved_tensors
        escort = exp_t(activations - const, ctx.t).pow(ctx.t)
        escort = escort / escort.sum(dim=-1, keepdim=True)
        return escort * g, None, None


def compute_normalization(activations: Tensor, t: float, num_iters: int = 5) -> Tensor:
    return _Normalization.apply(activations, t, num_iters)


def tempered_softmax(activations: Tensor, t: float, num_iters: int = 5) -> Tensor:
    if t == 1.0:
        return activations.softmax(dim=-1)
    return exp_t(activations - compute_normalization(activations, t, num_iters), t)


def bi_tempered_logistic_loss(activations, labels, t1, t2, label_smoothing=0.0, num_iters=5, reduction="mean"):
    """activations ``[..., num_classes]``; labels one-hot of the same shape, or int64 of one dim less.  t1 < 1 bounds the
    loss, t2 > 1 makes the softmax heavy-tailed (t2 < 1: finite support).  reduction 'none' | 'sum' | 'mean'."""
    if labels.dim() < activations.dim():
        onehot = torch.zeros_like(activations)
        onehot.scatter_(1, labels[..., None], 1)
    else:
        onehot = labels
    if (activations.is_cuda and activations.dim() >= 1 and activations.is_floating_point() and not onehot.requires_grad and t1 != 2.0
            and onehot.shape == activations.shape and activations.numel() > 0 and (label_smoothing <= 0 or activations.shape[-1] > 1)):
        # GPU: one fused HIP pass per direction, a wave per row of classes (csrc/ptb_pointwise.hip: bitempered_rows_kernel)
        from . import _pointwise as P

        kc = activations.shape[-1]
        act = P.as_f32(activations, "bi_tempered_logistic_loss").reshape(-1, kc)
        hot = P.as_f32(onehot.detach(), "bi_tempered_logistic_loss").reshape(-1, kc)
        loss = P.BiTemperedRows.apply(act, hot, float(t1), float(t2), float(max(label_smoothing, 0.0)), int(num_iters)).reshape(activations.shape[:-1])
        loss = loss.to(activations.dtype)
        if reduction == "none":
            return loss
        if reduction == "sum":
            return loss.sum()
        if reduction == "mean":
            return loss.mean()
        return None
    if label_smoothing > 0:
        k = onehot.shape[-1]
        onehot = (1 - label_smoothing * k / (k - 1)) * onehot + label_smoothing / (k - 1)
    probs = tempered_softmax(activations, t2, num_iters)
    per_class = (onehot * log_t(onehot + 1e-10, t1) - onehot * log_t(probs, t1)
                 - onehot.pow(2.0 - t1) / (2.0 - t1) + probs.pow(2.0 - t1) / (2.0 - t1))
    loss = per_class.sum(dim=-1)
    if reduction == "none":
        return loss
    if reduction == "sum":
        return loss.sum()
    if reduction == "mean":
        return loss.mean()


class BiTemperedLogisticLoss(nn.Module):
    def __init__(self, t1: float, t2: float, smoothing=0.0, ignore_index=None, reduction: str = "mean"):
        super().__init__()
        self.t1 = t1
        self.t2 = t2
        self.smoothing = smoothing
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, predictions: Tensor, targets: Tensor) -> Tensor:
        loss = bi_tempered_logistic_loss(predictions, targets, t1=self.t1, t2=self.t2, label_smoothing=self.smoothing, reduction="none")
        if self.ignore_index is not None:
            loss = loss * ~targets.eq(self.ignore_index)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


class BinaryBiTemperedLogisticLoss(nn.Module):
    """Two-class form with the signature of ``nn.BCEWithLogitsLoss``: predictions and targets are ``[B, 1, ...]``."""

    def __init__(self, t1: float, t2: float, smoothing: float = 0.0, ignore_index: Optional[int] = None, reduction: str = "mean"):
        super().__init__()
        self.t1 = t1
        self.t2 = t2
        self.smoothing = smoothing
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, predictions: Tensor, targets: Tensor) -> Tensor:
        if predictions.size(1) != 1 or targets.size(1) != 1:
            raise ValueError("Channel dimension for predictions and targets must be equal to 1")
        if predictions.is_cuda and predictions.shape == targets.shape and not targets.requires_grad and self.t1 != 2.0:
            return self._native(predictions, targets)
        two = torch.cat([-predictions, predictions], dim=1).moveaxis(1, -1)
        hot = torch.cat([1 - targets, targets], dim=1).moveaxis(1, -1)
        loss = bi_tempered_logistic_loss(two, hot, t1=self.t1, t2=self.t2, label_smoothing=self.smoothing, reduction="none").unsqueeze(dim=1)
        if self.ignore_index is not None:
            loss = torch.masked_fill(loss, targets.eq(self.ignore_index), 0)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss

    def _native(self, predictions: Tensor, targets: Tensor) -> Tensor:
        """Segmentation maps on the GPU: one fused HIP pass per direction (csrc/ptb_pointwise.hip) -- the two activations
        (-x, x), the iterative normalisation (5 iterations, like the reference's default) and the loss terms per pixel in
        registers, instead of ~60 full-tensor torch ops on a [B, ..., 2] expansion."""
        from . import _pointwise as P

        x = P.as_f32(predictions, "BinaryBiTemperedLogisticLoss")
        t = P.as_f32(targets.detach(), "BinaryBiTemperedLogisticLoss")
        reduce = self.reduction in ("mean", "sum")
        has_ignore = self.ignore_index is not None
        total, elem = P.BiTemperedBinarySums.apply(x.view(-1), t.view(-1), float(self.t1), float(self.t2), float(self.smoothing), 5,
                                                   has_ignore, float(self.ignore_index) if has_ignore else 0.0, not reduce)
        out_dtype = predictions.dtype if predictions.dtype.is_floating_point else torch.float32
        if self.reduction == "mean":
            return (total / max(x.numel(), 1)).to(out_dtype)
        if self.reduction == "sum":
            return total.to(out_dtype)
        return elem.view(predictions.shape).to(out_dtype)
