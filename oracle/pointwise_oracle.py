"""Oracle (test infrastructure only) for the reference's remaining elementwise + reduce losses.

float64 numpy restatements; citations are paths under ``pytorch_toolbelt/losses/``."""
import math

import numpy as np


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def _log1pexp_neg_abs(x):
    return np.log1p(np.exp(-np.abs(x)))


def _logsigmoid(x):
    return -(np.maximum(-x, 0.0) + _log1pexp_neg_abs(x))


def _reduce(loss, reduction):
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def soft_bce(x, t, weight=None, pos_weight=None, ignore_index=-100, reduction="mean", smooth_factor=None):
    """SoftBCEWithLogitsLoss.forward, soft_bce.py:29-46.  BCE-with-logits (torch formula):
    ``(1 - s) * x + (1 + (pw - 1) * s) * (log1p(exp(-|x|)) + max(-x, 0))``, times ``weight``; ignored elements -> 0."""
    x, t = _f64(x), _f64(t)
    s = (1 - t) * smooth_factor + t * (1 - smooth_factor) if smooth_factor is not None else t      # :30-33
    pw = _f64(pos_weight) if pos_weight is not None else 1.0
    loss = (1 - s) * x + (1 + (pw - 1) * s) * (_log1pexp_neg_abs(x) + np.maximum(-x, 0.0))         # :35-37
    if weight is not None:
        loss = loss * _f64(weight)
    if ignore_index is not None:
        loss = loss * (t != ignore_index)                                                          # :39-41
    return _reduce(loss, reduction)


def balanced_bce(x, t, gamma=1.0, ignore_index=None, reduction="mean"):
    """balanced_binary_cross_entropy_with_logits, balanced_bce.py:27-49 (the power is applied twice, :31 and :33-34)."""
    x, t = _f64(x), _f64(t)
    n_pos, n_neg = float((t == 1).sum()), float((t == 0).sum())                                    # :27-28
    pos_weight = np.float32(np.float32(n_neg) / np.float32(n_pos + n_neg + 1e-7)) ** np.float32(gamma)   # :30-31 in float32
    neg_weight = np.float32(1.0) - pos_weight                                                      # :32
    pos_term = float(np.float32(pos_weight) ** np.float32(gamma)) * t * _logsigmoid(x)             # :33
    neg_term = float(np.float32(neg_weight) ** np.float32(gamma)) * (1 - t) * _logsigmoid(-x)      # :34
    loss = -(pos_term + neg_term)                                                                  # :36
    if ignore_index is not None:
        loss = np.where(t == ignore_index, 0.0, loss)                                              # :38-39
    return _reduce(loss, reduction)


def quality_focal(x, t, beta=2.0, reduction="mean"):
    """QualityFocalLoss.forward, quality_focal_loss.py:30-45."""
    x, t = _f64(x), _f64(t)
    bce = np.maximum(x, 0.0) - x * t + _log1pexp_neg_abs(x)                                        # :33
    p = 1.0 / (1.0 + np.exp(-x))
    focal = np.abs(p - t) ** beta                                                                  # :34
    loss = focal * bce                                                                             # :35
    if reduction == "normalized":
        return loss.sum() / focal.sum()                                                            # :41-42
    return _reduce(loss, reduction)


def wing(output, target, width=5, curvature=0.5, reduction="mean"):
    """wing_loss, functional.py:250-277."""
    d = np.abs(_f64(target) - _f64(output))                                                        # :260
    c = width - width * math.log(1 + width / curvature)                                            # :268
    loss = np.where(d < width, width * np.log(1 + d / curvature), d - c)                           # :266-269
    return _reduce(loss, reduction)


def log_cosh(y_pred, y_true):
    """log_cosh_loss, functional.py:326-342: mean(z + softplus(-2 z) - log 2)."""
    z = _f64(y_pred) - _f64(y_true)
    y = -2.0 * z
    return (z + np.maximum(y, 0.0) + _log1pexp_neg_abs(y) - math.log(2.0)).mean()


def soft_ce(x, labels, smooth_factor=0.0, ignore_index=-100, reduction="mean", dim=1):
    """SoftCrossEntropyLoss.forward = label_smoothed_nll_loss(log_softmax(x, dim), ...), soft_ce.py:24-33 and
    functional.py:280-323 (the gathered dim is kept in the ignore branch, squeezed otherwise)."""
    x = _f64(x)
    z = x - x.max(axis=dim, keepdims=True)
    lprobs = z - np.log(np.exp(z).sum(axis=dim, keepdims=True))                                    # soft_ce.py:25
    tgt = np.expand_dims(np.asarray(labels, dtype=np.int64), dim)                                  # functional.py:292-293
    if ignore_index is not None:
        pad = tgt == ignore_index                                                                  # :296
        nll = -np.take_along_axis(lprobs, np.where(pad, 0, tgt), axis=dim)                         # :297-298
        smooth = -lprobs.sum(axis=dim, keepdims=True)                                              # :299
        nll, smooth = np.where(pad, 0.0, nll), np.where(pad, 0.0, smooth)                          # :303-304
    else:
        nll = -np.take_along_axis(lprobs, tgt, axis=dim).squeeze(dim)                              # :306, :309
        smooth = -lprobs.sum(axis=dim, keepdims=True).squeeze(dim)                                 # :307, :310
    if reduction == "sum":
        nll, smooth = nll.sum(), smooth.sum()
    if reduction == "mean":
        nll, smooth = nll.mean(), smooth.mean()
    return (1.0 - smooth_factor) * nll + (smooth_factor / x.shape[dim]) * smooth                   # :319-321
