"""Host-tensor evaluation of the loss surface: what runs when the prediction is a CPU tensor.

The reference's losses are ordinary torch modules and work on any device (its own tests run them on CPU tensors); the HIP kernels of
this package serve CUDA tensors only.  A CPU prediction therefore takes this module: plain differentiable torch algebra with the
reference's semantics (file:line cited per function), written against ``torch.nn.functional`` -- no kernels, no ``oracle/``.
The dispatch rule is the prediction's DEVICE and nothing else; a CUDA tensor never comes here and fails loudly when the extension
is missing.
"""
import math

import torch
import torch.nn.functional as F


def _reduce(loss, reduction):
    """"mean" | "sum" | "batchwise_mean" (a SUM over dim 0, as in the reference: functional.py:104-105) | anything else: unreduced."""
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "batchwise_mean":
        return loss.sum(dim=0)
    return loss


def _class_shaped(weights, like):
    return weights.to(device=like.device).reshape(1, -1, *([1] * (like.dim() - 2)))


def onehot_from_labels(labels, num_classes, ignore_index):
    """Label map [B, *] -> int64 one-hot [B, C, *]; pixels equal to ``ignore_index`` carry ``ignore_index`` in EVERY channel
    (losses/focal.py:92-105)."""
    labels = labels.long()
    if ignore_index is None:
        return F.one_hot(labels, num_classes).movedim(-1, 1)
    void = labels == ignore_index
    hot = F.one_hot(labels.masked_fill(void, 0), num_classes).movedim(-1, 1)
    return hot.masked_fill(void.unsqueeze(1), ignore_index)


def binary_focal(output, target, prob, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index, class_weights):
    """Focal-modulated BCE on logits (losses/functional.py:58-107): ``prob`` is the probability the focal term is built from
    (sigmoid of the logits, or their softmax along some dim); the BCE term always comes from the logits."""
    x, t = output.float(), target.float()
    p_true = prob * t + (1 - prob) * (1 - t)
    base = 1.0 - p_true
    if ignore_index is not None:
        # Deviation (tests/golden/losses6.npz): an ignored entry carries t = ignore_index, its base 1 - pt is negative
        # for about half of the logits, and a non-integer power of it is NaN -- the reference's masked_fill hides that in the VALUE,
        # but `0 * NaN` puts NaN into the GRADIENT of exactly the ignored entries (losses/functional.py:70, 90-94).  Here, as in the HIP
        # kernels, an ignored entry contributes a gradient of 0: its base is replaced before the power (the value is masked below
        # either way, so every finite number of the reference is unchanged).
        base = base.masked_fill(t == ignore_index, 1.0)
    if reduced_threshold is None:
        modulator = base ** gamma
    else:
        modulator = (base / (1 - reduced_threshold)) ** gamma
        modulator = torch.where(p_true < reduced_threshold, torch.ones_like(modulator), modulator)
    loss = modulator * F.binary_cross_entropy_with_logits(x, t, reduction="none")
    if alpha is not None:
        loss = loss * (alpha * t + (1 - alpha) * (1 - t))
    if class_weights is not None:
        loss = loss * _class_shaped(class_weights, loss)
    if ignore_index is not None:
        void = t == ignore_index
        loss = loss.masked_fill(void, 0)
        if normalized:
            modulator = modulator.masked_fill(void, 0)
    if normalized:
        loss = loss / modulator.sum(dtype=torch.float32).clamp_min(eps)
    return _reduce(loss, reduction)


def sigmoid_focal(output, labels, dense, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index, class_weights):
    target = dense if dense is not None else onehot_from_labels(labels, output.size(1), ignore_index)
    return binary_focal(output, target, torch.sigmoid(output.float()), gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index,
                        class_weights)


def softmax_act_focal(output, labels, dense, softmax_dim, gamma, alpha, reduction, normalized, reduced_threshold, eps, ignore_index,
                      class_weights):
    if softmax_dim is None:
        raise RuntimeError("focal_loss_with_logits(activation='softmax'): softmax_dim must be given (torch.softmax(dim=None) fails too)")
    target = dense if dense is not None else onehot_from_labels(labels, output.size(1), ignore_index)
    return binary_focal(output, target, torch.softmax(output.float(), dim=softmax_dim), gamma, alpha, reduction, normalized, reduced_threshold,
                        eps, ignore_index, class_weights)


def softmax_focal(output, target, class_weights, gamma, reduction, normalized, reduced_threshold, eps, ignore_index):
    """losses/functional.py:110-173: per pixel ``sum_c pt_c^gamma * BCE(x_c, onehot_c) * w_c`` with ``pt`` from the softmax
    probabilities; void pixels give 0 but still count in the "mean" denominator; ``normalized`` divides by the sum of ALL focal terms."""
    target = target.long()
    void = target == ignore_index
    hot = F.one_hot(target.masked_fill(void, 0), output.size(1)).movedim(-1, 1).float()
    prob = F.softmax(output, dim=1)
    p_wrong = (1 - hot) * prob + hot * (1 - prob)
    if reduced_threshold is None:
        modulator = p_wrong ** gamma
    else:
        modulator = torch.where(p_wrong < reduced_threshold, torch.ones_like(p_wrong), (p_wrong / reduced_threshold) ** gamma)
    loss = modulator * F.binary_cross_entropy_with_logits(output, hot, reduction="none")
    if class_weights is not None:
        loss = loss * _class_shaped(class_weights, loss)
    loss = loss.sum(dim=1) * (~void)
    if normalized:
        loss = loss / modulator.sum().clamp_min(eps)
    if reduction == "batchwise_mean":
        return loss.sum(0)
    return _reduce(loss, reduction)


# ------------------------------------------------------------------------------------------------ region losses
def region_sums(output, target, dims):
    """(sum o * t, sum o + t) over ``dims`` (losses/functional.py:188-247)."""
    assert output.size() == target.size()
    if dims is None:
        return torch.sum(output * target), torch.sum(output + target)
    dims = tuple(dims) if isinstance(dims, (list, tuple)) else (dims,)
    return torch.sum(output * target, dim=dims), torch.sum(output + target, dim=dims)


def region_statistics(y_pred, y_true, mode, from_logits, ignore_index):
    """Per-class (intersection, predicted mass, target mass) over batch and pixels (losses/dice.py:59-113, losses/jaccard.py:48-90):
    probabilities via log-softmax / log-sigmoid + exp when ``from_logits``; multiclass labels one-hot; the ignore mask multiplies both."""
    assert y_true.size(0) == y_pred.size(0)
    bs = y_true.size(0)
    y_true = y_true.to(y_pred.device)
    if mode == "multiclass":
        prob = y_pred.log_softmax(dim=1).exp() if from_logits else y_pred
        C = prob.size(1)
        prob = prob.reshape(bs, C, -1)
        labels = y_true.reshape(bs, -1)
        if labels.size(1) != prob.size(2):
            raise RuntimeError(f"target shape {tuple(y_true.shape)} does not match prediction shape {tuple(y_pred.shape)}")
        if ignore_index is not None:
            keep = labels != ignore_index
            prob = prob * keep.unsqueeze(1)
            hot = F.one_hot((labels * keep).long(), C).permute(0, 2, 1) * keep.unsqueeze(1)
        else:
            hot = F.one_hot(labels.long(), C).permute(0, 2, 1)
        true = hot.type_as(prob)
    else:
        prob = F.logsigmoid(y_pred).exp() if from_logits else y_pred
        C = 1 if mode == "binary" else prob.size(1)
        if y_true.numel() != prob.numel():
            raise RuntimeError(f"target shape {tuple(y_true.shape)} does not match prediction shape {tuple(y_pred.shape)}")
        prob = prob.reshape(bs, C, -1)
        true = y_true.reshape(bs, C, -1)
        if ignore_index is not None:
            keep = true != ignore_index
            prob, true = prob * keep, true * keep
        true = true.type_as(prob)
    return (prob * true).sum(dim=(0, 2)), prob.sum(dim=(0, 2)), true.sum(dim=(0, 2))


# ------------------------------------------------------------------------------------------------ Lovasz
def _lovasz_columns(errors, fg, hinge):
    """Every column of ``errors`` / ``fg`` [P, S] is one segment: sort its errors in decreasing order, weight them by the discrete
    gradient of the Jaccard index along that order (losses/lovasz.py:23-34), sum.  Returns [S].  (Ties in the errors may come out in
    another order than the reference's sort: the dot product does not depend on it.)"""
    sorted_err, order = torch.sort(errors, dim=0, descending=True)
    fg_sorted = fg.gather(0, order)
    positives = fg_sorted.sum(dim=0, keepdim=True)
    inter = positives - fg_sorted.cumsum(0)
    union = positives + (1 - fg_sorted).cumsum(0)
    jaccard = 1.0 - inter / union
    grad = torch.cat([jaccard[:1], jaccard[1:] - jaccard[:-1]], dim=0)
    if hinge:
        sorted_err = F.relu(sorted_err)
    return (sorted_err * grad).sum(dim=0)


def lovasz_hinge(logits, labels, per_image, ignore_index):
    """losses/lovasz.py:37-86: errors ``1 - x * (2 y - 1)``; void pixels dropped; only void pixels -> 0 with a zero gradient."""
    def one(x, y):
        x, y = x.reshape(-1), y.reshape(-1)
        if ignore_index is not None:
            keep = y != ignore_index
            x, y = x[keep], y[keep]
        if y.numel() == 0:
            return x.sum() * 0.0
        y = y.float()
        return _lovasz_columns((1.0 - x * (2.0 * y - 1.0)).unsqueeze(1), y.unsqueeze(1), hinge=True)[0]

    if not per_image:
        return one(logits, labels)
    parts = [one(x, y) for x, y in zip(logits, labels)]
    return sum(parts[1:], parts[0]) / len(parts) if parts else logits.sum() * 0.0


def lovasz_softmax(probas, labels, classes, per_image, ignore_index):
    """losses/lovasz.py:92-160: per class ``|fg - p_c|`` sorted, dotted with the Jaccard gradient; mean over the selected classes
    ("present": those that occur among the valid pixels)."""
    if probas.dim() == 3:
        probas = probas.unsqueeze(1)
    C = probas.size(1)
    if C == 1 and len(classes) > 1:
        raise ValueError("Sigmoid output possible only with 1 class")

    def one(p, y):
        p = p.movedim(1, -1).reshape(-1, C)
        y = y.reshape(-1)
        if ignore_index is not None:
            keep = y != ignore_index
            p, y = p[keep], y[keep]
        if p.numel() == 0:
            return p.sum() * 0.0
        wanted = list(range(C)) if classes in ("all", "present") else list(classes)
        fg = (y.unsqueeze(1) == torch.as_tensor(wanted, device=y.device).unsqueeze(0)).type_as(p)      # [P, S]
        per_class = _lovasz_columns((fg - p[:, wanted]).abs(), fg, hinge=False)
        if classes == "present":
            occurs = fg.sum(dim=0) > 0
            if not bool(occurs.any()):
                return p.sum() * 0.0
            per_class = per_class[occurs]
        return per_class.mean()

    if not per_image:
        return one(probas, labels)
    parts = [one(p.unsqueeze(0), y.unsqueeze(0)) for p, y in zip(probas, labels)]
    return sum(parts[1:], parts[0]) / len(parts) if parts else probas.sum() * 0.0


# ------------------------------------------------------------------------------------------------ elementwise losses
def wing(output, target, width, curvature, reduction):
    """losses/functional.py:250-277: ``width * log(1 + d / curvature)`` for ``d < width``, ``d - C`` beyond."""
    d = (target - output).abs()
    c = width - width * math.log(1 + width / curvature)
    loss = torch.where(d < width, width * torch.log(1 + d / curvature), d - c)
    return _reduce(loss, reduction) if reduction in ("mean", "sum") else loss


def log_cosh(y_pred, y_true):
    d = y_pred - y_true
    return torch.mean(d + F.softplus(-2.0 * d) - math.log(2.0))


def balanced_bce(logits, targets, gamma, ignore_index, reduction):
    """losses/balanced_bce.py:27-48 (the class-balance weights are raised to ``gamma`` twice there, kept)."""
    n_pos, n_neg = targets.eq(1).sum(), targets.eq(0).sum()
    w_pos = torch.pow(n_neg / (n_pos + n_neg + 1e-7), gamma)
    w_neg = 1.0 - w_pos
    loss = -(w_pos.pow(gamma) * targets * F.logsigmoid(logits) + w_neg.pow(gamma) * (1 - targets) * F.logsigmoid(-logits))
    if ignore_index is not None:
        loss = loss.masked_fill(targets.eq(ignore_index), 0)
    return _reduce(loss, reduction) if reduction in ("mean", "sum") else loss


def quality_focal(predictions, targets, beta, reduction):
    """losses/quality_focal_loss.py:30-48: ``|sigmoid(x) - t|^beta * BCE(x, t)`` in float32."""
    x, t = predictions.float(), targets.float()
    modulator = (x.sigmoid() - t).abs() ** beta
    loss = modulator * F.binary_cross_entropy_with_logits(x, t, reduction="none")
    if reduction == "normalized":
        return loss.sum() / modulator.sum()
    return _reduce(loss, reduction) if reduction in ("mean", "sum") else loss
