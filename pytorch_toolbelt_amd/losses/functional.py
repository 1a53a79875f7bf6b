"""Loss functionals (drop-in for ``pytorch_toolbelt.losses.functional``).

``focal_loss_with_logits``, ``softmax_focal_loss_with_logits``, ``soft_dice_score`` and ``soft_jaccard_score`` run as
single-pass HIP reductions (csrc/ptb_losses.hip), ``wing_loss`` and ``log_cosh_loss`` as fused elementwise + reduce
passes (csrc/ptb_pointwise.hip); the scalar epilogues are torch ops on tiny tensors.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F  # noqa: F401  (module attribute of the reference's functional.py)

from ..utils.support import pytorch_toolbelt_deprecated
from . import _host as H
from . import _kernels as K

__all__ = [
    "focal_loss_with_logits",
    "softmax_focal_loss_with_logits",
    "sigmoid_focal_loss",
    "soft_jaccard_score",
    "soft_dice_score",
    "wing_loss",
    "log_cosh_loss",
    "label_smoothed_nll_loss",
]


def focal_loss_with_logits(
    output: torch.Tensor,
    target: torch.Tensor,
    gamma: float = 2.0,
    alpha: Optional[float] = 0.25,
    reduction: str = "mean",
    normalized: bool = False,
    reduced_threshold: Optional[float] = None,
    eps: float = 1e-6,
    ignore_index=None,
    activation: str = "sigmoid",
    softmax_dim: Optional[int] = None,
    class_weights: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Binary focal loss between logits and a same-shaped (0/1 or soft) target, always evaluated in float32.

    ``loss_i = (1 - pt_i)^gamma * BCE_i`` (optionally alpha-balanced, class-weighted, reduced-threshold variant,
    normalised by the sum of focal terms); elements whose target equals ``ignore_index`` contribute 0.
    reduction: "mean" (over all elements, ignored ones included) | "sum" | "batchwise_mean" (a sum over dim 0, as in
    the reference) | anything else -> unreduced tensor.
    """
    if activation != "sigmoid":   # anything else means softmax in the reference (functional.py:61-64)
        return _softmax_act_focal(output, None, target, softmax_dim, gamma, alpha, reduction, normalized, reduced_threshold, eps,
                                  ignore_index, class_weights)
    return _sigmoid_focal(output, None, target, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index, class_weights)


def _focal_flags(alpha, reduced_threshold, ignore_index, normalized, reduction):
    flags = 0
    if alpha is not None:
        flags |= K.SEG_HAS_ALPHA
    if reduced_threshold is not None:
        flags |= K.SEG_REDUCED
    if ignore_index is not None:
        flags |= K.SEG_HAS_IGNORE
        if normalized:
            flags |= K.SEG_MASK_FOCAL_TERM
    if not normalized:
        flags |= K.SEG_NO_TERM      # sums[1] is only read for normalized=True
    want_map = reduction not in ("mean", "sum")
    if want_map:
        flags |= K.SEG_ELEMWISE
    return flags, want_map


def _softmax_act_focal(output, labels, dense, softmax_dim, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index,
                       class_weights):
    """``activation="softmax"``: the focal term's probability is ``softmax(output, dim=softmax_dim)`` while the BCE term keeps the
    logits (functional.py:61-78).  The contiguous tensor is handed to the kernel as the [B, C, HW] view whose C is the softmax
    dimension; class_weights keep following dim 1 of the original tensor."""
    if not output.is_cuda:      # host tensors: plain torch algebra (losses/_host.py); CUDA tensors never go there
        return H.softmax_act_focal(output, labels, dense, softmax_dim, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index,
                                   class_weights)
    if softmax_dim is None:
        raise RuntimeError("focal_loss_with_logits(activation='softmax'): softmax_dim must be given (torch.softmax(dim=None) fails too)")
    shape = tuple(output.shape)
    nd = len(shape)
    if not -nd <= softmax_dim < nd:
        raise IndexError(f"Dimension out of range (expected to be in range of [{-nd}, {nd - 1}], but got {softmax_dim})")
    d = softmax_dim % nd
    x = K._f32c(output, "focal loss")
    B, C, HW = math.prod(shape[:d]), shape[d], math.prod(shape[d + 1:])
    x = x.reshape(B, C, HW)
    flags, want_map = _focal_flags(alpha, reduced_threshold, ignore_index, normalized, reduction)
    if dense is not None:
        if tuple(dense.shape) != shape:
            raise RuntimeError(f"target shape {tuple(dense.shape)} does not match output shape {shape}")
        dense = K._f32c(dense, "focal loss").reshape(B, C, HW)
    else:
        if d != 1:   # index labels one-hot along dim 1 (focal.py:88-105); only that layout has the on-the-fly one-hot
            raise NotImplementedError("BinaryFocalLoss with label maps and activation='softmax' needs softmax_dim=1")
        labels = labels.to(device=x.device, dtype=torch.int64).reshape(B, -1).contiguous()
        if labels.shape[1] != HW:
            raise RuntimeError(f"target shape {tuple(labels.shape)} does not match output shape {shape}")
    cw = (None, 0, 0, 1)
    if class_weights is not None:
        if nd < 2:
            raise RuntimeError("class_weights need a tensor with a class dimension (dim 1)")
        w = class_weights.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if w.numel() != shape[1]:
            raise RuntimeError("class_weights must have one entry per channel")
        if d == 1:
            cw = (w, 1, shape[1], 1)
        elif d > 1:
            cw = (w, 2, shape[1], max(1, math.prod(shape[2:d])))
        else:
            cw = (w, 3, shape[1], max(1, math.prod(shape[2:])))
    sums, elem = K.SoftmaxActFocalSums.apply(
        x, labels, dense, cw, flags, float(gamma), float(alpha if alpha is not None else 0.0),
        float(reduced_threshold if reduced_threshold is not None else 0.0),
        int(ignore_index) if (ignore_index is not None and labels is not None) else 0,
        float(ignore_index) if ignore_index is not None else 0.0,
    )
    norm = sums[1].clamp_min(eps) if normalized else None
    if want_map:
        loss = elem.view(shape)
        if normalized:
            loss = loss / norm.float()
        if reduction == "batchwise_mean":
            loss = loss.sum(dim=0)
        return loss
    total = sums[0] / norm if normalized else sums[0]
    if reduction == "mean":
        total = total / x.numel()
    return total.float()


def _sigmoid_focal(output, labels, dense, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index, class_weights):
    if not output.is_cuda:
        return H.sigmoid_focal(output, labels, dense, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index, class_weights)
    shape = output.shape
    x = K.as_bchw(K._f32c(output, "focal loss"))
    flags = 0
    if alpha is not None:
        flags |= K.SEG_HAS_ALPHA
    if reduced_threshold is not None:
        flags |= K.SEG_REDUCED
    if ignore_index is not None:
        flags |= K.SEG_HAS_IGNORE
        if normalized:
            flags |= K.SEG_MASK_FOCAL_TERM
    if not normalized:
        flags |= K.SEG_NO_TERM      # sums[1] is only read for normalized=True
    want_map = reduction not in ("mean", "sum")
    if want_map:
        flags |= K.SEG_ELEMWISE
    if dense is not None:
        if dense.shape != shape:
            raise RuntimeError(f"target shape {tuple(dense.shape)} does not match output shape {tuple(shape)}")
        dense = K.as_bchw(K._f32c(dense, "focal loss"))
    else:
        labels = labels.to(device=x.device, dtype=torch.int64).reshape(x.shape[0], -1).contiguous()
        if labels.shape[1] != x.shape[2]:   # the kernel takes B, C, HW from the logits: a smaller target would be read past its end
            raise RuntimeError(f"target shape {tuple(labels.shape)} does not match output shape {tuple(shape)}")
    cw = None
    if class_weights is not None:
        cw = class_weights.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if cw.numel() != x.shape[1]:
            raise RuntimeError("class_weights must have one entry per channel")
    if (labels is not None and reduction in ("mean", "sum") and not normalized and cw is None and alpha is None and reduced_threshold is None
            and ignore_index is None and float(gamma) == 2.0 and x.numel()):
        # BinaryFocalLoss() as the README uses it: the whole forward is one launch (kernel + in-launch tail)
        one = K.FocalScalar.run(x, labels, flags, 2.0, 1.0 / x.numel() if reduction == "mean" else 1.0)
        if one is not None:
            return one
    sums, elem = K.SigmoidFocalSums.apply(
        x, labels, dense, cw, flags, float(gamma), float(alpha if alpha is not None else 0.0),
        float(reduced_threshold if reduced_threshold is not None else 0.0),
        int(ignore_index) if (ignore_index is not None and labels is not None) else 0,
        float(ignore_index) if ignore_index is not None else 0.0,
    )
    norm = sums[1].clamp_min(eps) if normalized else None
    if want_map:
        loss = elem.view(shape)
        if normalized:
            loss = loss / norm.float()
        if reduction == "batchwise_mean":
            loss = loss.sum(dim=0)
        return loss
    total = sums[0] / norm if normalized else sums[0]
    if reduction == "mean":
        total = total / x.numel()
    return total.float()


def softmax_focal_loss_with_logits(
    output: torch.Tensor,
    target: torch.Tensor,
    class_weights: Optional[torch.Tensor] = None,
    gamma: float = 2.0,
    reduction: str = "mean",
    normalized: bool = False,
    reduced_threshold: Optional[float] = None,
    eps: float = 1e-6,
    ignore_index: int = -100,
) -> torch.Tensor:
    """Softmax flavour of the focal loss for ``output [B, C, *]`` / ``target [B, *]`` (like nn.CrossEntropyLoss):
    per pixel ``sum_c pt_c^gamma * BCE(x_c, onehot_c) * w_c`` with ``pt`` from the softmax probabilities; pixels whose
    label equals ``ignore_index`` give 0 but still count in the "mean" denominator."""
    if not output.is_cuda:
        return H.softmax_focal(output, target, class_weights, gamma, reduction, normalized, reduced_threshold, eps, ignore_index)
    x = K.as_bchw(K._f32c(output, "softmax focal loss"))
    labels = target.to(device=x.device, dtype=torch.int64).reshape(x.shape[0], -1).contiguous()
    if labels.shape[1] != x.shape[2]:
        raise RuntimeError(f"target shape {tuple(target.shape)} does not match output shape {tuple(output.shape)}")
    cw = None
    if class_weights is not None:
        cw = class_weights.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    want_map = reduction not in ("mean", "sum")
    sums, pix = K.SoftmaxFocalSums.apply(x, labels, cw, 1 if reduced_threshold is not None else 0, float(gamma),
                                         float(reduced_threshold if reduced_threshold is not None else 0.0), int(ignore_index), want_map)
    norm = sums[1].clamp_min(eps) if normalized else None
    if want_map:
        loss = pix.view(target.shape)
        if normalized:
            loss = loss / norm.float()
        if reduction == "batchwise_mean":
            loss = loss.sum(0)
        return loss
    total = sums[0] / norm if normalized else sums[0]
    if reduction == "mean":
        total = total / labels.numel()
    return total.float()


@pytorch_toolbelt_deprecated("Function sigmoid_focal_loss is deprecated. Please use focal_loss_with_logits instead.")
def sigmoid_focal_loss(*input, **kwargs):
    return focal_loss_with_logits(*input, **kwargs)


@pytorch_toolbelt_deprecated("Function reduced_focal_loss is deprecated. Please use focal_loss_with_logits instead.")
def reduced_focal_loss(output: torch.Tensor, target: torch.Tensor, threshold=0.5, gamma=2.0, reduction="mean"):
    return focal_loss_with_logits(output, target, alpha=None, gamma=gamma, reduction=reduction, reduced_threshold=threshold)


def _region_sums(output: torch.Tensor, target: torch.Tensor, dims):
    """(intersection, cardinality) = (sum o*t, sum o+t) over ``dims`` via the fused statistics kernel."""
    if not output.is_cuda:
        return H.region_sums(output, target, dims)
    assert output.size() == target.size()
    nd = output.dim()
    if dims is None:
        kept = None
        o3 = output.reshape(1, 1, -1)
    else:
        red = sorted({d % nd for d in (dims if isinstance(dims, (list, tuple)) else [dims])})
        keep = [d for d in range(nd) if d not in red]
        if len(keep) > 1:
            # several kept dimensions (e.g. dims=(2, 3): one score per sample and class): they become ONE class axis of length
            # prod(kept), the reduced ones one pixel axis -- a view when the kept dimensions lead (the usual case), else one copy
            shape = [int(output.shape[d]) for d in keep]
            perm = keep + red
            o3 = output.permute(perm).reshape(1, math.prod(shape), -1)
            t3 = target.permute(perm).reshape(o3.shape)
            stats = K.RegionStats.apply(K._f32c(o3, "soft score"), None, K._f32c(t3, "soft score"), K.PROB_IDENTITY, False, 0, 0.0)
            return stats[0].float().reshape(shape), (stats[1] + stats[2]).float().reshape(shape)
        if not keep:
            kept = None
            o3 = output.reshape(1, 1, -1)
        else:
            kept = keep[0]
            lead = 1
            for s in output.shape[:kept]:
                lead *= int(s)
            o3 = output.reshape(lead, output.shape[kept], -1)
    x = K._f32c(o3, "soft score")
    t = K._f32c(target.reshape(o3.shape), "soft score")
    stats = K.RegionStats.apply(x, None, t, K.PROB_IDENTITY, False, 0, 0.0)
    inter, card = stats[0], stats[0 + 1] + stats[2]
    if kept is None:
        inter, card = inter[0], card[0]
    return inter.float(), card.float()


def soft_jaccard_score(output: torch.Tensor, target: torch.Tensor, smooth: float = 0.0, eps: float = 1e-7, dims=None) -> torch.Tensor:
    """(I + smooth) / max(U + smooth, eps) with I = sum(output*target), U = sum(output+target) - I over ``dims``."""
    inter, card = _region_sums(output, target, dims)
    union = card - inter
    return (inter + smooth) / (union + smooth).clamp_min(eps)


def soft_dice_score(output: torch.Tensor, target: torch.Tensor, smooth: float = 0.0, eps: float = 1e-7, dims=None) -> torch.Tensor:
    """(2 I + smooth) / max(S + smooth, eps) with I = sum(output*target), S = sum(output+target) over ``dims``."""
    inter, card = _region_sums(output, target, dims)
    return (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)


def wing_loss(output: torch.Tensor, target: torch.Tensor, width=5, curvature=0.5, reduction="mean"):
    """Wing loss for landmark regression (https://arxiv.org/pdf/1711.06753.pdf): ``width * log(1 + d / curvature)`` for
    ``d = |target - output| < width``, ``d - C`` beyond (C makes it continuous); "sum" | "mean" | unreduced.  One fused
    HIP pass (reference losses/functional.py:250-277)."""
    if not output.is_cuda:
        return H.wing(output, target, width, curvature, reduction)
    from . import _pointwise as P

    x = P.as_f32(output, "wing_loss")
    t = P.as_f32(target.detach(), "wing_loss")
    if x.shape != t.shape:
        x, t = (v.contiguous() for v in torch.broadcast_tensors(x, t))
    c = width - width * math.log(1 + width / curvature)
    reduce = reduction in ("mean", "sum")
    sums, elem = P.PointwiseSums.apply(x, t, None, None, P.WING, 0, float(width), float(curvature), float(c), 0.0, 1, 1, not reduce)
    if reduction == "sum":
        return sums[0].to(output.dtype)
    if reduction == "mean":
        return (sums[0] / max(x.numel(), 1)).to(output.dtype)
    return elem.view(x.shape).to(output.dtype)


def log_cosh_loss(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """``mean(log(cosh(y_pred - y_true)))`` evaluated as ``d + softplus(-2 d) - log 2`` (reference
    losses/functional.py:326-342), one fused HIP pass."""
    if not y_pred.is_cuda:
        return H.log_cosh(y_pred, y_true)
    from . import _pointwise as P

    x = P.as_f32(y_pred, "log_cosh_loss")
    t = P.as_f32(y_true.detach(), "log_cosh_loss")
    if x.shape != t.shape:
        x, t = (v.contiguous() for v in torch.broadcast_tensors(x, t))
    sums, _ = P.PointwiseSums.apply(x, t, None, None, P.LOGCOSH, 0, 0.0, 0.0, 0.0, 0.0, 1, 1, False)
    return (sums[0] / max(x.numel(), 1)).to(y_pred.dtype)


def label_smoothed_nll_loss(lprobs: torch.Tensor, target: torch.Tensor, epsilon: float, ignore_index=None, reduction="mean", dim=-1) -> torch.Tensor:
    """Label-smoothed negative log-likelihood of LOG-PROBABILITIES (reference losses/functional.py:280-323):
    ``(1 - eps) * nll + eps / C * smooth`` with ``nll = -lprobs[target]`` and ``smooth = -sum_c lprobs``; positions whose
    target equals ``ignore_index`` contribute 0 (and keep the gathered dim in the unreduced output).  Plain torch on
    purpose: it consumes log-probabilities somebody already computed; ``SoftCrossEntropyLoss`` is the fused path."""
    if target.dim() == lprobs.dim() - 1:
        target = target.unsqueeze(dim)
    if ignore_index is not None:
        pad = target.eq(ignore_index)
        nll = -lprobs.gather(dim=dim, index=target.masked_fill(pad, 0))
        smooth = -lprobs.sum(dim=dim, keepdim=True)
        nll, smooth = nll.masked_fill(pad, 0.0), smooth.masked_fill(pad, 0.0)
    else:
        nll = -lprobs.gather(dim=dim, index=target).squeeze(dim)
        smooth = -lprobs.sum(dim=dim, keepdim=True).squeeze(dim)
    if reduction == "sum":
        nll, smooth = nll.sum(), smooth.sum()
    if reduction == "mean":
        nll, smooth = nll.mean(), smooth.mean()
    return (1.0 - epsilon) * nll + (epsilon / lprobs.size(dim)) * smooth
